"""Detection glue (counterpart of pet/rcnn/utils/misc.py:6-94)."""
import torch

from pet.utils.data.structures.bounding_box import BoxList


def cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def permute_and_flatten(layer, N, A, C, H, W):
    """[N, A*C, H, W] -> [N, H*W*A, C]; free when `layer` is NHWC in memory."""
    return layer.view(N, -1, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)


def concat_box_prediction_layers(box_cls, box_regression):
    cls_flat, reg_flat = [], []
    C = 1
    for c, r in zip(box_cls, box_regression):
        N, AxC, H, W = c.shape
        A = r.shape[1] // 4
        C = AxC // A
        cls_flat.append(permute_and_flatten(c, N, A, C, H, W))
        reg_flat.append(permute_and_flatten(r, N, A, 4, H, W))
    return cat(cls_flat, dim=1).reshape(-1, C), cat(reg_flat, dim=1).reshape(-1, 4)


def keep_only_positive_boxes(boxes, roi_batch_size=-1, across_sample=False):
    """labels > 0 only, at most roi_batch_size per image (random subset), misc.py:54-94."""
    assert isinstance(boxes, (list, tuple)) and isinstance(boxes[0], BoxList) and boxes[0].has_field("labels")
    out = []
    if (not across_sample) or len(boxes) < 2:
        for b in boxes:
            hl = getattr(b, "host_labels", None)
            if hl is not None and len(hl) == len(b):
                # labels already on the host (fused sampler): choose on the host, one asynchronous index upload
                import numpy as np
                inds_h = np.flatnonzero(hl > 0)
                if 0 < roi_batch_size < len(inds_h):
                    inds_h = inds_h[torch.randperm(len(inds_h))[:roi_batch_size].numpy()]
                inds = torch.from_numpy(inds_h).pin_memory().to(b.bbox.device, non_blocking=True)
                out.append(b[inds])
                continue
            inds = (b.get_field("labels") > 0).nonzero().squeeze(1)
            if 0 < roi_batch_size < inds.shape[0]:
                inds = inds[torch.randperm(inds.shape[0], device=inds.device)[:roi_batch_size]]
            out.append(b[inds])
        return out
    assert len(boxes) == 2, "only support 2 images on one gpu, but get {}".format(boxes)
    per = [(b.get_field("labels") > 0).nonzero().squeeze(1) for b in boxes]
    allpos = torch.cat(per)
    split = per[0].shape[0]
    if allpos.shape[0] > roi_batch_size:
        ind = torch.sort(torch.randperm(allpos.shape[0], device=allpos.device)[:roi_batch_size])[0]
        per = [allpos[ind[ind < split]], allpos[ind[ind >= split]]]
    return [b[i] for b, i in zip(boxes, per)]
