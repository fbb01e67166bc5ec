"""Device image preparation: resize (Pillow-exact antialiased bilinear on uint8) + flip + channel order +
normalisation + zero padding in two launches (cpm_image_prep, csrc/image_prep.hip).

Replaces the reference's host chain Resize -> RandomHorizontalFlip -> ToTensor -> Normalize
(pet/utils/data/transforms/transforms.py:29-115) and to_image_list's padding (structures/image_list.py:56-66);
the decoded uint8 image is what crosses PCIe.  Bit-identical to that chain.
"""
import ctypes
import functools
import math

import numpy as np
import torch

from . import _hip as H

PRECISION_BITS = 32 - 8 - 2


@functools.lru_cache(maxsize=4096)
def resample_tables(in_size, out_size):
    """Pillow's precompute_coeffs (bilinear filter, box = the whole axis) + normalize_coeffs_8bpc, in the same
    double-precision operation order.  Returns (bounds int32 [out,2], coeffs int32 [out,ksize], ksize)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    t = np.abs(((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss)
    w = np.where((t < 1.0) & (x < xmax[:, None]), 1.0 - t, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):                                   # sequential sum, as the C loop
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(0.5 + w * float(1 << PRECISION_BITS)).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk, ksize


def value_table(mean, std, to_bgr255=True):
    """[3,256] fp32: what ToTensor (uint8 -> float / 255) followed by Normalize (x255 when to_bgr255, - mean, / std)
    makes of byte v in OUTPUT channel o -- evaluated with the same fp32 tensor operations (transforms.py:99-115)."""
    v = torch.arange(256, dtype=torch.float32).div(255)
    if to_bgr255:
        v = v * 255
    mean = torch.as_tensor(np.asarray(mean, dtype=np.float64).reshape(-1), dtype=torch.float32)
    std = torch.as_tensor(np.asarray(std, dtype=np.float64).reshape(-1), dtype=torch.float32)
    return v[None, :].repeat(3, 1).sub_(mean[:, None]).div_(std[:, None]).contiguous()


_dev_tables = {}


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, device.index)
    t = _dev_tables.get(key)
    if t is None:
        b, k, ks = resample_tables(in_size, out_size)
        if len(_dev_tables) > 8192:
            _dev_tables.clear()
        t = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
        _dev_tables[key] = t
    return t


def image_prep(src, out_size, flip, lut, swap_rb, dst):
    """src: uint8 [H,W,3] on the device (decoded RGB).  out_size (oh, ow).  lut: [3,256] fp32 on the device
    (value_table).  dst: one image slot of the batch -- a [3,dstH,dstW] fp32 view that is either contiguous (NCHW)
    or channels-last (strides of an NHWC batch); every element is written."""
    if not (src.is_cuda and dst.is_cuda and lut.is_cuda):
        raise RuntimeError("image_prep runs on MI355X only (no CPU fallback)")
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3 or not src.is_contiguous():
        raise RuntimeError("image_prep: src must be a contiguous uint8 [H,W,3] tensor")
    if dst.dtype != torch.float32 or dst.dim() != 3 or dst.shape[0] != 3:
        raise RuntimeError("image_prep: dst must be a fp32 [3,H,W] slot")
    hgt, wid = int(src.shape[0]), int(src.shape[1])
    oh, ow = int(out_size[0]), int(out_size[1])
    dh, dw = int(dst.shape[1]), int(dst.shape[2])
    if dst.stride() == (1, 3 * dw, 3):
        layout = 1
    elif dst.is_contiguous():
        layout = 0
    else:
        raise RuntimeError("image_prep: dst slot must be NCHW-contiguous or channels-last")
    null = ctypes.c_void_p(0)
    hb = hk = vb = vk = null
    hks = vks = 0
    tmp = null
    keep = []
    if ow != wid:
        b, k, hks = _device_tables(wid, ow, src.device)
        ws = H.workspace(hgt * ow * 3, src.device)
        hb, hk, tmp = H.ptr(b), H.ptr(k), H.ptr(ws)
        keep += [b, k, ws]
    if oh != hgt:
        b, k, vks = _device_tables(hgt, oh, src.device)
        vb, vk = H.ptr(b), H.ptr(k)
        keep += [b, k]
    with H.guard(src.device):
        rc = H.lib().cpm_image_prep(H.ptr(src), hgt, wid, hb, hk, hks, vb, vk, vks, oh, ow, int(bool(flip)),
                                    H.ptr(lut), int(bool(swap_rb)), tmp, H.ptr(dst), dh, dw, layout, H.stream())
    H.check(rc, "image_prep")
    return dst


def cv_round(x):
    """cvRound: nearest integer, ties to even (OpenCV's dsize = cvRound(src * f))."""
    return int(np.rint(x))


def resize_linear(src, scale, flip=False, swap_rb=True):
    """Test-time image blob (pet/rcnn/core/test.py:340-358): src uint8 [H,W,3] on the device -> fp32 [3, oh, ow] where
    (ow, oh) = cvRound(W * scale), cvRound(H * scale); cv2.resize(..., fx=scale, fy=scale, INTER_LINEAR) on float data,
    optional mirror of the source, RGB -> BGR plane order."""
    if not src.is_cuda:
        raise RuntimeError("resize_linear runs on MI355X only (no CPU fallback)")
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3 or not src.is_contiguous():
        raise RuntimeError("resize_linear: src must be a contiguous uint8 [H,W,3] tensor")
    hgt, wid = int(src.shape[0]), int(src.shape[1])
    oh, ow = cv_round(hgt * float(scale)), cv_round(wid * float(scale))
    dst = torch.empty((3, oh, ow), dtype=torch.float32, device=src.device)
    inv = 1.0 / float(scale)
    with H.guard(src.device):
        rc = H.lib().cpm_image_resize_linear(H.ptr(src), hgt, wid, oh, ow, H.f(inv), H.f(inv), int(bool(flip)),
                                             int(bool(swap_rb)), H.ptr(dst), H.stream())
    H.check(rc, "image_resize_linear")
    return dst
