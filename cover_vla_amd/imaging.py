"""Image pre-processing either side of the hot path, restated from the published
algorithms of the un-vendored libraries the reference calls, in two forms that give the same bytes:

  * host (numpy / PIL)  -- what the drop-in classes use when handed host images, the reference's own flow;
  * device (libcover_hip `cover_resample_axis`, `cover_u8_hwc_to_f32_chw_norm`, `cover_resize_bilinear_pad_f32`) -- the raw
    camera frame is uploaded once (0.9 MB) and resampled in HBM, no host round trip in front of the verifier towers.

  process_raw_image_to_jpg(frame)      CoVer_VLA/inference/experiments/robot/simpler/eval_utils.py:228-286:
        tf.image.resize(BILINEAR, antialias=True) to 256 x 256, then tf.cast(uint8). TensorFlow is not in this image and
        not in /root/reference: restated from TF's ScaleAndTranslate op (kernels/image/scale_and_translate_op.cc:
        triangle kernel of radius 1 stretched by max(1/scale, 1), spans clamped to the image, weights normalised per
        span, rows gathered first, then columns, fp32 accumulation in span order, truncating cast). PARITY UNPINNED at TF
        (no TF here to generate vectors); cross-checked within 1 grey level against Pillow's antialiased BILINEAR.
  siglip_preprocess(image, size=384)   open_clip's transform for the SigLIP2 checkpoints (efficient_ensemble_merged.py:69,
        338): Resize((size, size), BICUBIC) on the PIL image ("squash"), ToTensor, Normalize(0.5, 0.5). torchvision on a
        PIL image calls Image.resize, so the host form IS that call; the device form restates Pillow's Resample.c 8-bit
        path (22-bit fixed-point coefficients) and is bit-exact against Pillow (tests/test_imaging_*.py).
  resize_with_pad(img, w, h, pad)      lerobot_custom/lerobot/common/policies/pi0/modeling_pi0.py:131-150 on the device.
  cv2_resize_lanczos4(img, (w, h))     the policy-side adapter's cv2.resize(..., interpolation=cv2.INTER_LANCZOS4)
        (INT-ACT/src/experiments/env_adapters/simpler.py:48-52) followed by process_images (src/utils/pipeline.py:55-67:
        x * (1 / 255), then (x - 0.5) / 0.5). OpenCV is not in this image and not in /root/reference: restated from OpenCV's
        published imgproc/src/resize.cpp -- 8 taps per axis at sx - 3 .. sx + 4 with sx = floor((dx + 0.5) * scale - 0.5), NO
        antialiasing when shrinking, interpolateLanczos4's closed-form weights normalised to sum 1, stored as shorts with 11
        fractional bits (cvRound), taps beyond the border replicated, exact int32 row pass, then saturate((sum + 2^21) >> 22).
        PARITY UNPINNED at OpenCV (no cv2 here to generate vectors); the device form is bit-exact against this host form, and both
        are cross-checked against Pillow's LANCZOS (a = 3, antialiased) on smooth images within a stated bound.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2     # Pillow Resample.c


# ------------------------------------------------------------------------------------------------ filter banks (host, exact)
def _bicubic(x: float) -> float:
    a = -0.5                      # Pillow's bicubic_filter
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _triangle(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def pillow_coeffs(in_size: int, out_size: int, filt: str = "bicubic") -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow precompute_coeffs + normalize_coeffs_8bpc (Resample.c) -> (bounds int32 [out,2] = (xmin, count),
    coefficients int32 [out, ksize] with PRECISION_BITS fractional bits, ksize). Doubles, as Pillow."""
    f, support = {"bicubic": (_bicubic, 2.0), "bilinear": (_triangle, 1.0)}[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
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
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def tf_spans(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """TF ComputeSpansCore for the triangle kernel with antialias=True, scale = out/in (fp32), translate 0 ->
    (bounds int32 [out,2] = (start, count), weights fp32 [out, span_size], span_size). Every intermediate is fp32, as in TF."""
    f32 = np.float32
    scale = f32(out_size) / f32(in_size)
    inv_scale = f32(1.0) / scale
    kernel_scale = max(inv_scale, f32(1.0))
    radius = f32(1.0)
    span_size = min(2 * int(math.ceil(float(radius * kernel_scale))) + 1, in_size)
    one_over = f32(1.0) / kernel_scale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    weights = np.zeros((out_size, span_size), dtype=np.float32)
    for x in range(out_size):
        sample_f = f32(f32(x) + f32(0.5)) * inv_scale
        if sample_f < 0 or sample_f > in_size:
            continue
        span_start = int(math.ceil(float(f32(f32(sample_f - f32(radius * kernel_scale)) - f32(0.5)))))
        span_end = int(math.floor(float(f32(f32(sample_f + f32(radius * kernel_scale)) - f32(0.5)))))
        span_start = min(max(span_start, 0), in_size - 1)
        span_end = min(max(span_end, 0), in_size - 1) + 1
        tmp = []
        total = f32(0.0)
        for src in range(span_start, span_end):
            kernel_pos = f32(f32(f32(src) + f32(0.5)) - sample_f)
            w = f32(_triangle(float(abs(f32(kernel_pos * one_over)))))
            total = f32(total + w)
            tmp.append(w)
        if abs(total) >= 1000.0 * np.finfo(np.float32).tiny:
            inv = f32(1.0) / total
            for j, w in enumerate(tmp):
                weights[x, j] = f32(w * inv)
        bounds[x] = (span_start, span_end - span_start)
    return bounds, weights, span_size


def cv2_lanczos4_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """OpenCV resize() tables for INTER_LANCZOS4 on 8-bit images along one axis -> (bounds int32 [out,2] = (start, count),
    coefficients int32 [out, 8] holding the 11-bit fixed-point shorts, 8). Taps that fall outside the image are REPLICATED border
    pixels in OpenCV (HResizeLanczos4 / the clip() of the vertical pass); integer sums are linear, so their weights are folded
    into the border pixel here and the span stays contiguous -- the same sum, exactly."""
    scale = 1.0 / (float(out_size) / float(in_size))          # double inv_scale_x = dsize / ssize; scale_x = 1. / inv_scale_x
    s45 = 0.70710678118654752440084436210485
    cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, 8), dtype=np.int32)
    for dx in range(out_size):
        fx = np.float32((dx + 0.5) * scale - 0.5)             # fx = (float)((dx + 0.5) * scale_x - 0.5)
        sx = int(math.floor(float(fx)))                       # cvFloor
        fx = np.float32(fx - np.float32(sx))
        co = np.zeros(8, dtype=np.float32)
        if fx < np.finfo(np.float32).eps:                     # interpolateLanczos4: x < FLT_EPSILON
            co[3] = 1.0
        else:
            x = float(fx)
            y0 = -(x + 3) * math.pi * 0.25
            s0, c0 = math.sin(y0), math.cos(y0)
            total = np.float32(0.0)
            for i in range(8):
                y = -(x + 3 - i) * math.pi * 0.25
                co[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
                total = np.float32(total + co[i])
            inv = np.float32(1.0) / total
            co = (co * inv).astype(np.float32)
        q = np.rint(co.astype(np.float32) * np.float32(2048.0)).astype(np.int64)      # saturate_cast<short>(cbuf * INTER_RESIZE_COEF_SCALE): cvRound
        q = np.clip(q, -32768, 32767)
        taps = [min(max(sx - 3 + j, 0), in_size - 1) for j in range(8)]                # replicated border
        lo, hi = taps[0], taps[-1]
        w = np.zeros(8, dtype=np.int64)
        for t, v in zip(taps, q):
            w[t - lo] += v
        bounds[dx] = (lo, hi - lo + 1)
        kk[dx] = w.astype(np.int32)
    return bounds, kk, 8


# ------------------------------------------------------------------------------------------------ host forms
def cv2_resize_lanczos4(image: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(image, (width, height), interpolation=cv2.INTER_LANCZOS4) for uint8 HWC images, host form (numpy int64 sums)."""
    a = np.asarray(image)
    if a.dtype != np.uint8 or a.ndim != 3:
        raise ValueError("uint8 HWC image expected")
    W, H = size
    bw, kw, _ = cv2_lanczos4_coeffs(a.shape[1], W)
    bh, kh, _ = cv2_lanczos4_coeffs(a.shape[0], H)
    x = a.astype(np.int64)
    rows = np.zeros((a.shape[0], W, a.shape[2]), dtype=np.int64)
    for o in range(W):
        s, n = int(bw[o, 0]), int(bw[o, 1])
        rows[:, o] = np.tensordot(x[:, s:s + n], kw[o, :n].astype(np.int64), axes=([1], [0]))
    out = np.zeros((H, W, a.shape[2]), dtype=np.int64)
    for o in range(H):
        s, n = int(bh[o, 0]), int(bh[o, 1])
        out[o] = np.tensordot(kh[o, :n].astype(np.int64), rows[s:s + n], axes=([0], [0]))
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def simpler_preprocess_image(frame: np.ndarray, image_size=(224, 224)):
    """BridgeSimplerAdapter.preprocess, image half (simpler.py:48-65): LANCZOS4 resize -> uint8 [1,3,H,W] -> x * (1/255) -> (x - 0.5) / 0.5,
    fp32 in [-1, 1]. Host form (torch CPU)."""
    import torch
    img = cv2_resize_lanczos4(frame, image_size)
    t = torch.as_tensor(img, dtype=torch.uint8).permute(2, 0, 1)[None]
    t = t * (1 / 255.0)                                       # rescale(): uint8 tensor * python float -> fp32
    return (t - torch.tensor([0.5, 0.5, 0.5])[None, :, None, None]) / torch.tensor([0.5, 0.5, 0.5])[None, :, None, None]


def _to_rgb_u8(image) -> np.ndarray:
    """The shape handling of process_raw_image_to_jpg (eval_utils.py:253-268): grey -> 3 channels, RGBA -> RGB."""
    a = np.asarray(image)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, axis=-1)
    elif a.ndim != 3:
        raise ValueError(f"Expected 2D or 3D image, got shape: {a.shape}")
    if a.shape[-1] == 1:
        a = np.repeat(a, 3, axis=-1)
    elif a.shape[-1] == 4:
        a = a[..., :3]
    elif a.shape[-1] != 3:
        raise ValueError(f"Expected 1, 3, or 4 channels, got: {a.shape[-1]}")
    return a


def _gather_f32(x: np.ndarray, bounds: np.ndarray, weights: np.ndarray, axis: int) -> np.ndarray:
    """out[o] = sum_j fp32(w[o,j] * x[start+j]) accumulated sequentially in fp32 along `axis` (0 rows, 1 columns)."""
    x = np.moveaxis(x.astype(np.float32), axis, 0)
    out = np.zeros((bounds.shape[0],) + x.shape[1:], dtype=np.float32)
    for o in range(bounds.shape[0]):
        s, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.zeros(x.shape[1:], dtype=np.float32)
        for j in range(n):
            acc = (acc + (x[s + j] * weights[o, j]).astype(np.float32)).astype(np.float32)
        out[o] = acc
    return np.moveaxis(out, 0, axis)


def process_raw_image_to_jpg(image, max_res: int = 256) -> np.ndarray:
    """eval_utils.py:228-286 for array inputs -> uint8 [max_res, max_res, 3]."""
    a = _to_rgb_u8(image)
    bh, wh, _ = tf_spans(a.shape[0], max_res)
    bw, ww, _ = tf_spans(a.shape[1], max_res)
    rows = _gather_f32(a, bh, wh, 0)                 # [max_res, W, 3]   (GatherRows)
    out = _gather_f32(rows, bw, ww, 1)               # [max_res, max_res, 3] (GatherColumns)
    return np.clip(out, 0, 255).astype(np.uint8)     # tf.cast(float -> uint8) truncates


def siglip_preprocess(image, size: int = 384, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """open_clip's SigLIP / SigLIP2 eval transform on a PIL image or an HWC uint8 array -> torch fp32 [3, size, size]."""
    import torch
    from PIL import Image
    if isinstance(image, np.ndarray):
        image = Image.fromarray(image.astype("uint8"))       # efficient_ensemble_merged.py:334-337
    image = image.convert("RGB").resize((size, size), Image.BICUBIC)
    t = torch.from_numpy(np.asarray(image).copy()).permute(2, 0, 1).to(torch.float32).div(255)      # ToTensor
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return (t - m) / s                                        # Normalize


# ------------------------------------------------------------------------------------------------ device forms
class DeviceImagePipeline:
    """raw camera frame (uint8 HWC, host or device) -> [verifier image fp32 [1,3,S,S] on the device] without leaving HBM:
    TF antialias-bilinear to 256^2 (process_raw_image_to_jpg) -> Pillow bicubic to S^2 -> ToTensor/Normalize.
    Span tables are built once per input geometry on the host and kept on the device."""

    def __init__(self, device="cuda:0", mid: int = 256, size: int = 384, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        import torch
        self.dev, self.mid, self.size = torch.device(device), mid, size
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self._tf = {}
        b, k, ks = pillow_coeffs(mid, size, "bicubic")
        self._pil = (torch.from_numpy(b).to(self.dev), torch.from_numpy(k).to(self.dev), ks)

    def _tables(self, H, W):
        import torch
        key = (H, W)
        if key not in self._tf:
            bh, wh, sh = tf_spans(H, self.mid)
            bw, ww, sw = tf_spans(W, self.mid)
            self._tf[key] = tuple(torch.from_numpy(x).to(self.dev) for x in (bh, wh, bw, ww)) + (sh, sw)
        return self._tf[key]

    def raw_to_jpg(self, frame_u8):
        """uint8 [H,W,3] (device) -> uint8 [mid,mid,3] (device): process_raw_image_to_jpg."""
        import torch
        from . import _lib as L
        h = L.lib()
        H, W, Cc = frame_u8.shape
        bh, wh, bw, ww, sh, sw = self._tables(H, W)
        st = torch.cuda.current_stream().cuda_stream
        rows = torch.empty(self.mid, W, Cc, dtype=torch.float32, device=self.dev)
        L.check(h.cover_resample_axis(frame_u8.data_ptr(), 0, rows.data_ptr(), 1, H, W, Cc, self.mid, W, 0, bh.data_ptr(), wh.data_ptr(),
                                      sh, 0, st), "resample rows")
        out = torch.empty(self.mid, self.mid, Cc, dtype=torch.uint8, device=self.dev)
        L.check(h.cover_resample_axis(rows.data_ptr(), 1, out.data_ptr(), 0, self.mid, W, Cc, self.mid, self.mid, 1, bw.data_ptr(),
                                      ww.data_ptr(), sw, 0, st), "resample columns")
        return out

    def siglip(self, img_u8):
        """uint8 [mid,mid,3] (device) -> fp32 [1,3,size,size] (device): Pillow bicubic (horizontal pass, then vertical)."""
        import torch
        from . import _lib as L
        h = L.lib()
        b, k, ks = self._pil
        M, S = self.mid, self.size
        assert tuple(img_u8.shape) == (M, M, 3)
        st = torch.cuda.current_stream().cuda_stream
        hor = torch.empty(M, S, 3, dtype=torch.uint8, device=self.dev)
        L.check(h.cover_resample_axis(img_u8.data_ptr(), 0, hor.data_ptr(), 0, M, M, 3, M, S, 1, b.data_ptr(), k.data_ptr(), ks, 1, st),
                "bicubic horizontal")
        ver = torch.empty(S, S, 3, dtype=torch.uint8, device=self.dev)
        L.check(h.cover_resample_axis(hor.data_ptr(), 0, ver.data_ptr(), 0, M, S, 3, S, S, 0, b.data_ptr(), k.data_ptr(), ks, 1, st),
                "bicubic vertical")
        out = torch.empty(1, 3, S, S, dtype=torch.float32, device=self.dev)
        L.check(h.cover_u8_hwc_to_f32_chw_norm(ver.data_ptr(), out.data_ptr(), S, S, self.mean, self.std, st), "normalise")
        return out

    def __call__(self, frame_u8):
        import torch
        if isinstance(frame_u8, np.ndarray):
            frame_u8 = torch.from_numpy(np.ascontiguousarray(_to_rgb_u8(frame_u8))).to(self.dev)
        return self.siglip(self.raw_to_jpg(frame_u8.contiguous()))

    # ---- policy side (second half): cv2 LANCZOS4 to the policy's input size + process_images, on the device
    def _cv_tables(self, H, W, size):
        import torch
        key = ("cv", H, W, size)
        if key not in self._tf:
            bw, kw, _ = cv2_lanczos4_coeffs(W, size[0])
            bh, kh, _ = cv2_lanczos4_coeffs(H, size[1])
            self._tf[key] = tuple(torch.from_numpy(x).to(self.dev) for x in (bw, kw, bh, kh))
        return self._tf[key]

    def policy_resize(self, frame_u8, size=(224, 224)):
        """uint8 [H,W,3] (device) -> uint8 [size[1], size[0], 3] (device): cv2.resize(frame, size, INTER_LANCZOS4), simpler.py:48-52."""
        import torch
        from . import _lib as L
        h = L.lib()
        H, W, Cc = frame_u8.shape
        bw, kw, bh, kh = self._cv_tables(H, W, tuple(size))
        st = torch.cuda.current_stream().cuda_stream
        rows = torch.empty(H, size[0], Cc, dtype=torch.int32, device=self.dev)
        L.check(h.cover_resample_axis(frame_u8.data_ptr(), 0, rows.data_ptr(), 2, H, W, Cc, H, size[0], 1, bw.data_ptr(), kw.data_ptr(), 8, 2, st),
                "lanczos4 horizontal")
        out = torch.empty(size[1], size[0], Cc, dtype=torch.uint8, device=self.dev)
        L.check(h.cover_resample_axis(rows.data_ptr(), 2, out.data_ptr(), 0, H, size[0], Cc, size[1], size[0], 0, bh.data_ptr(), kh.data_ptr(), 8, 2, st),
                "lanczos4 vertical")
        return out

    def policy_image(self, frame_u8, size=(224, 224)):
        """raw camera frame -> the policy's fp32 [1,3,H,W] input in [-1, 1] on the device (simpler.py:48-65): LANCZOS4 resize, then
        process_images (x * (1/255), (x - 0.5) / 0.5)."""
        import torch
        from . import _lib as L
        if isinstance(frame_u8, np.ndarray):
            frame_u8 = torch.from_numpy(np.ascontiguousarray(_to_rgb_u8(frame_u8))).to(self.dev)
        img = self.policy_resize(frame_u8.contiguous(), size)
        out = torch.empty(1, 3, size[1], size[0], dtype=torch.float32, device=self.dev)
        half = (C.c_float * 3)(0.5, 0.5, 0.5)
        L.check(L.lib().cover_u8_hwc_to_f32_chw_scale_norm(img.data_ptr(), out.data_ptr(), size[1], size[0], C.c_float(np.float32(1 / 255.0)), half, half,
                                                           torch.cuda.current_stream().cuda_stream), "scale + normalise")
        return out


def resize_with_pad(img, width: int, height: int, pad_value: float = -1.0):
    """modeling_pi0.py:131-150 on the device: fp32 [b,c,h,w] -> [b,c,height,width], aspect-preserving bilinear resize,
    padded on the LEFT and TOP. A no-op when the size already fits (same tensor returned)."""
    import torch
    from . import _lib as L
    if img.ndim != 4:
        raise ValueError(f"(b,c,h,w) expected, but {img.shape}")
    cur_height, cur_width = img.shape[2:]
    if cur_height == height and cur_width == width:
        return img
    if not img.is_cuda:
        raise L.CoverError("resize_with_pad: device tensor expected (no CPU path)")
    ratio = max(cur_width / width, cur_height / height)
    rh, rw = int(cur_height / ratio), int(cur_width / ratio)
    pad_h, pad_w = max(0, int(height - rh)), max(0, int(width - rw))
    b, c = img.shape[:2]
    x = img.to(torch.float32).contiguous()
    out = torch.empty(b, c, rh + pad_h, rw + pad_w, dtype=torch.float32, device=img.device)
    L.check(L.lib().cover_resize_bilinear_pad_f32(x.data_ptr(), out.data_ptr(), b * c, cur_height, cur_width, rh, rw, rh + pad_h, rw + pad_w,
                                                  pad_h, pad_w, float(pad_value), torch.cuda.current_stream().cuda_stream),
            "resize_bilinear_pad")
    return out
