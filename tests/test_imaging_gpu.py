"""Device forms of the image pre-processing (cover_resample_axis / cover_u8_hwc_to_f32_chw_norm through the C ABI) against the
host forms: bit-exact with Pillow for open_clip's SigLIP2 transform, bit-exact with the host TF restatement for
process_raw_image_to_jpg, and the verifier scoring the same features through either path."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from cover_vla_amd import imaging as IM  # noqa: E402
from tests.test_imaging_cpu import _images  # noqa: E402


@pytest.mark.parametrize("name", ["noise", "smooth", "blocks"])
def test_device_pipeline_equals_host_forms(dev, name):
    raw = _images()[name]                                                    # 480 x 640 x 3 uint8 (the simulator's frame size)
    pipe = IM.DeviceImagePipeline(device="cuda:0")
    jpg = pipe.raw_to_jpg(torch.from_numpy(raw).to(dev))
    host_jpg = IM.process_raw_image_to_jpg(raw)
    assert jpg.dtype == torch.uint8 and np.array_equal(jpg.cpu().numpy(), host_jpg)        # same fp32 operations in the same order
    x = pipe.siglip(jpg)
    ref = IM.siglip_preprocess(host_jpg, 384)                                              # PIL bicubic + ToTensor + Normalize
    assert x.shape == (1, 3, 384, 384) and torch.equal(x[0].cpu(), ref)
    # the 8-bit intermediate is Pillow's, bit for bit
    pil = np.asarray(Image.fromarray(host_jpg).resize((384, 384), Image.BICUBIC))
    back = ((x[0].cpu() * 0.5 + 0.5) * 255).round().permute(1, 2, 0).numpy().astype(np.uint8)
    assert np.array_equal(back, pil)
    assert torch.equal(pipe(raw), x)                                                       # ndarray entry point


def test_device_pipeline_other_geometry(dev):
    g = np.random.default_rng(3).integers(0, 256, size=(224, 224, 3), dtype=np.uint8)      # an upscale through the TF stage
    pipe = IM.DeviceImagePipeline(device="cuda:0")
    assert np.array_equal(pipe.raw_to_jpg(torch.from_numpy(g).to(dev)).cpu().numpy(), IM.process_raw_image_to_jpg(g))


@pytest.mark.parametrize("name,size", [("noise", (224, 224)), ("smooth", (224, 224)), ("blocks", (256, 192)), ("noise", (700, 500))])
def test_device_lanczos4_policy_resize_equals_host_form(dev, name, size):
    """SURVEY 8 f2, policy side: cv2.resize(frame, size, INTER_LANCZOS4) + process_images (simpler.py:48-65) on the device (the OpenCV
    8-bit fixed-point mode of cover_resample_axis + cover_u8_hwc_to_f32_chw_scale_norm) against the host restatement: same integers,
    same fp32 operations -> bit-exact. (Parity with OpenCV itself is unpinned: cv2 is not in the image -- tests/test_imaging_cpu.py.)"""
    raw = _images()[name]
    pipe = IM.DeviceImagePipeline(device="cuda:0")
    out = pipe.policy_resize(torch.from_numpy(raw).to(dev), size)
    host = IM.cv2_resize_lanczos4(raw, size)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (size[1], size[0], 3)
    assert np.array_equal(out.cpu().numpy(), host)
    x = pipe.policy_image(raw, size)
    assert torch.equal(x.cpu(), IM.simpler_preprocess_image(raw, size))
