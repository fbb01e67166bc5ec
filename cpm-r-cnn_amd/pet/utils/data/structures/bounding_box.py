"""BoxList: a set of boxes of one image plus per-box fields (counterpart of
pet/utils/data/structures/bounding_box.py:9-335, the subset the detection path touches).
Conventions kept: size = (image_width, image_height); xyxy boxes are pixel-inclusive, so widths and
areas carry the "+1" (TO_REMOVE) of the reference (bounding_box.py:306-316)."""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2:
            raise ValueError("bbox should have 2 dimensions, got {}".format(bbox.ndimension()))
        if bbox.size(-1) != 4:
            raise ValueError("last dimension of bbox should have a size of 4, got {}".format(bbox.size(-1)))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox, self.size, self.mode = bbox, image_size, mode
        self.extra_fields = {}

    # ---- fields -------------------------------------------------------------------------------
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, other):
        self.extra_fields.update(other.extra_fields)

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        for f in fields if isinstance(fields, (list, tuple)) else [fields]:
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return out

    # ---- geometry -----------------------------------------------------------------------------
    def _xyxy(self):
        if self.mode == "xyxy":
            return self.bbox.split(1, dim=-1)
        x, y, w, h = self.bbox.split(1, dim=-1)
        return x, y, x + (w - 1).clamp(min=0), y + (h - 1).clamp(min=0)

    def convert(self, mode):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        x1, y1, x2, y2 = self._xyxy()
        if mode == "xyxy":
            box = torch.cat((x1, y1, x2, y2), dim=-1)
        else:
            box = torch.cat((x1, y1, x2 - x1 + 1, y2 - y1 + 1), dim=-1)
        out = BoxList(box, self.size, mode=mode)
        out._copy_extra_fields(self)
        return out

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    def clip_to_image(self, remove_empty=True):
        w, h = self.size
        self.bbox[:, 0].clamp_(min=0, max=w - 1)
        self.bbox[:, 1].clamp_(min=0, max=h - 1)
        self.bbox[:, 2].clamp_(min=0, max=w - 1)
        self.bbox[:, 3].clamp_(min=0, max=h - 1)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def resize(self, size):
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        x1, y1, x2, y2 = self._xyxy()
        out = BoxList(torch.cat((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=-1), size, mode="xyxy")
        out._copy_extra_fields(self)
        return out.convert(self.mode)

    def transpose(self, method):
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        w, h = self.size
        x1, y1, x2, y2 = self._xyxy()
        if method == FLIP_LEFT_RIGHT:
            box = torch.cat((w - x2 - 1, y1, w - x1 - 1, y2), dim=-1)
        else:
            box = torch.cat((x1, h - y2, x2, h - y1), dim=-1)
        out = BoxList(box, self.size, mode="xyxy")
        out._copy_extra_fields(self)
        return out.convert(self.mode)

    # ---- tensor-like --------------------------------------------------------------------------
    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        if isinstance(item, torch.Tensor) and item.dtype in (torch.bool, torch.uint8) and self.extra_fields:
            item = item.nonzero().squeeze(1)       # one device->host size query instead of one per field
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)
