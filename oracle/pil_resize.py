"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image transform in front of the hot path:

    transforms.Compose([ToPILImage(), Resize(256), CenterCrop(224), ToTensor(), Normalize(mean, std)])   (util/data_utils.py:48-54)

The resize is Pillow's (pinned version in this image: 12.2.0) antialiased bilinear resample for 8-bit images, restated from its
published algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
Vertical_8bpc): per output pixel a window of `support = max(scale, 1)` input pixels on either side of the centre, triangle
weights normalised to 1, converted to 22-bit fixed point, accumulated from 1 << 21 and shifted back -- horizontally first into an
8-bit intermediate, then vertically.  Pinned bit for bit against Pillow itself: tests/golden/resize_pil.npz
(oracle/gen_resize_golden.py), tests/test_oracle_golden.py::test_pil_resize_matches_pillow.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_tables(in_size, out_size):
    """(bounds [out, 2] int32 = (first input index, tap count), coefficients [out, ksize] int32 fixed point)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, np.float64)
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, out_size, axis):
    """one 8-bit resample pass along `axis` (0 = vertical, 1 = horizontal) of an (H, W, C) uint8 image"""
    bounds, kk = bilinear_tables(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for i in range(out_size):
        lo, n = bounds[i]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[i, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """Pillow's Image.resize((out_w, out_h), Image.BILINEAR) on an (H, W, 3) uint8 array"""
    if img.shape[1] != out_w:
        img = _pass(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _pass(img, out_h, 0)
    return img


def resized_hw(h, w, size=256):
    """torchvision.transforms.Resize(int): the shorter side becomes `size`, the longer int(size * long / short)"""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def crop_origin(h, w, ch=224, cw=224):
    """torchvision.transforms.CenterCrop: int(round((h - ch) / 2.0)) with Python's round (half to even)"""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def reference_transform(frame, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), size=256, crop=224):
    """(H, W, 3) uint8 frame -> (3, crop, crop) float32, as the reference's Compose does"""
    hr, wr = resized_hw(frame.shape[0], frame.shape[1], size)
    img = resize_bilinear_u8(frame, hr, wr)
    top, left = crop_origin(hr, wr, crop, crop)
    img = img[top:top + crop, left:left + crop].astype(np.float32) / np.float32(255.0)
    img = (img - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(img.transpose(2, 0, 1))
