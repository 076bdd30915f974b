"""Host forms of the image pre-processing restatements (cover_vla_amd/imaging.py): the Pillow coefficient tables reproduce
Image.resize bit for bit; the TensorFlow antialias-bilinear restatement (parity unpinned at TF: not installed, not vendored)
agrees with Pillow's antialiased BILINEAR within one grey level; ToTensor/Normalize arithmetic."""
import numpy as np
import pytest
import torch
from PIL import Image

from cover_vla_amd import imaging as IM


def _images():
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    smooth = np.stack([(yy * 255 // 479), (xx * 255 // 639), ((yy + xx) * 255 // 1118)], -1).astype(np.uint8)
    blocks = (((yy // 37) + (xx // 53)) % 2 * 255).astype(np.uint8)[:, :, None].repeat(3, -1)
    return {"noise": noise, "smooth": smooth, "blocks": blocks}


def _apply_fixed(img, bounds, kk, axis):
    """numpy form of Pillow's 8-bit pass (what cover_resample_axis does with fixed_point = 1)."""
    x = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.zeros((bounds.shape[0],) + x.shape[1:], dtype=np.int64)
    for o in range(bounds.shape[0]):
        s, n = bounds[o]
        acc = np.full(x.shape[1:], 1 << (IM.PRECISION_BITS - 1), dtype=np.int64)
        for j in range(n):
            acc = acc + x[s + j] * int(kk[o, j])
        out[o] = np.clip(acc >> IM.PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


@pytest.mark.parametrize("name", ["noise", "smooth", "blocks"])
@pytest.mark.parametrize("in_hw,out,filt", [((256, 256), 384, "bicubic"), ((480, 640), 256, "bilinear"), ((480, 640), 224, "bicubic")])
def test_pillow_tables_reproduce_pillow(name, in_hw, out, filt):
    img = _images()[name][: in_hw[0], : in_hw[1]]
    bw, kw, _ = IM.pillow_coeffs(in_hw[1], out, filt)
    bh, kh, _ = IM.pillow_coeffs(in_hw[0], out, filt)
    mine = _apply_fixed(_apply_fixed(img, bw, kw, 1), bh, kh, 0)          # horizontal pass, then vertical (ImagingResample)
    ref = np.asarray(Image.fromarray(img).resize((out, out), Image.BICUBIC if filt == "bicubic" else Image.BILINEAR))
    assert np.array_equal(mine, ref)


@pytest.mark.parametrize("name", ["noise", "smooth", "blocks"])
def test_process_raw_image_to_jpg_vs_pillow_antialias(name):
    img = _images()[name]
    out = IM.process_raw_image_to_jpg(img)
    assert out.shape == (256, 256, 3) and out.dtype == np.uint8
    ref = np.asarray(Image.fromarray(img).resize((256, 256), Image.BILINEAR)).astype(np.int32)
    d = np.abs(out.astype(np.int32) - ref)
    # same triangle filter stretched by the scale factor; TF accumulates in fp32 and TRUNCATES, Pillow rounds in fixed point
    assert d.max() <= 1, d.max()
    # truncation vs rounding: TF's value is never above Pillow's
    assert (out.astype(np.int32) - ref).max() <= 0


def test_tf_spans_properties():
    b, w, span = IM.tf_spans(480, 256)
    assert span == 2 * int(np.ceil(480 / 256)) + 1 == 5
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)                          # normalised per span
    assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= 480).all() and (b[:, 1] <= span).all()
    # identity geometry: a 256 -> 256 resize is the identity (single unit weight on the pixel itself)
    b1, w1, _ = IM.tf_spans(256, 256)
    img = _images()["noise"][:256, :256]
    assert np.array_equal(IM.process_raw_image_to_jpg(img), img)
    # grey and RGBA inputs (eval_utils.py:253-268)
    assert IM.process_raw_image_to_jpg(img[:, :, 0]).shape == (256, 256, 3)
    assert IM.process_raw_image_to_jpg(np.concatenate([img, img[:, :, :1]], -1)).shape == (256, 256, 3)


def test_siglip_preprocess_matches_manual_transform():
    img = _images()["noise"][:256, :256]
    t = IM.siglip_preprocess(img, 384)
    assert t.shape == (3, 384, 384) and t.dtype == torch.float32
    pil = Image.fromarray(img).resize((384, 384), Image.BICUBIC)
    ref = (torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float().div(255) - 0.5) / 0.5
    assert torch.equal(t, ref)
    assert torch.equal(IM.siglip_preprocess(Image.fromarray(img), 384), t)   # PIL input == ndarray input (:334-337)


def test_cv2_lanczos4_restatement_properties():
    """The policy-side resize (cv2.resize(..., INTER_LANCZOS4), INT-ACT/src/experiments/env_adapters/simpler.py:48-52). PARITY UNPINNED
    at OpenCV (cv2 is neither installed nor vendored): the restatement of resize.cpp's 8-bit fixed-point path is checked on what the
    algorithm guarantees -- identity geometry is the identity (fx = 0 -> weight 1 on the pixel itself), constants are preserved, the
    11-bit coefficient rows sum to 2048 +- 2, border taps replicate -- and cross-checked against Pillow's LANCZOS (a = 3, antialiased:
    another kernel) on smooth images: within 1 grey level, mean |difference| < 0.5."""
    imgs = _images()
    noise, smooth = imgs["noise"], imgs["smooth"]
    assert np.array_equal(IM.cv2_resize_lanczos4(noise, (noise.shape[1], noise.shape[0])), noise)
    const = np.full((37, 53, 3), 201, dtype=np.uint8)
    assert (IM.cv2_resize_lanczos4(const, (224, 224)) == 201).all()
    b, k, ks = IM.cv2_lanczos4_coeffs(640, 224)
    assert ks == 8 and (np.abs(k.sum(1) - 2048) <= 2).all()
    assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= 640).all() and (b[:, 1] <= 8).all()
    # interior spans are the eight taps sx - 3 .. sx + 4 around floor((dx + 0.5) * 640 / 224 - 0.5)
    dx = 100
    sx = int(np.floor((dx + 0.5) * (640 / 224) - 0.5))
    assert tuple(b[dx]) == (sx - 3, 8)
    out = IM.cv2_resize_lanczos4(smooth, (224, 224))
    assert out.shape == (224, 224, 3) and out.dtype == np.uint8
    pil = np.asarray(Image.fromarray(smooth).resize((224, 224), Image.LANCZOS)).astype(np.int32)
    d = np.abs(out.astype(np.int32) - pil)
    assert d.max() <= 2 and d.mean() < 0.5, (d.max(), d.mean())
    up = IM.cv2_resize_lanczos4(smooth[:120, :160], (320, 240))
    pu = np.asarray(Image.fromarray(smooth[:120, :160]).resize((320, 240), Image.LANCZOS)).astype(np.int32)
    assert np.abs(up.astype(np.int32) - pu).max() <= 2
    # the adapter's image half: uint8 [1,3,H,W] * (1/255), (x - 0.5) / 0.5
    t = IM.simpler_preprocess_image(smooth, (224, 224))
    ref = (torch.from_numpy(out).permute(2, 0, 1)[None] * (1 / 255.0) - 0.5) / 0.5
    assert t.dtype == torch.float32 and torch.equal(t, ref) and float(t.min()) >= -1.0 and float(t.max()) <= 1.0
