"""Writes tests/golden/jpeg_*.jpg and jpeg_expected.npz: small JPEG files encoded by Pillow (libjpeg-turbo) and the RGB bytes Pillow
decodes them to — what VTexture2D::LoadJPEG (csrc/host/JpegDecoder.cpp) must give byte for byte.  Run from the repository root:
python tests/golden/make_jpeg_fixtures.py (needs Pillow; the test that reads the fixtures does not)."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)


def picture(w, h):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 9.0 + y / 17.0), 127 + 120 * np.cos(x / 5.0), 127 + 100 * np.sin(y / 7.0) * np.cos(x / 11.0)], -1)
    return np.clip(img + rng.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)


CASES = {
    "420_odd": dict(size=(37, 23), mode="RGB", quality=75, subsampling=2),
    "422": dict(size=(40, 24), mode="RGB", quality=90, subsampling=1),
    "444_q30": dict(size=(33, 17), mode="RGB", quality=30, subsampling=0),
    "grey": dict(size=(29, 31), mode="L", quality=80),
    "420_restart": dict(size=(64, 40), mode="RGB", quality=85, subsampling=2, restart_marker_blocks=3),
    "420_optimized": dict(size=(50, 50), mode="RGB", quality=60, subsampling=2, optimize=True),
    "420_progressive": dict(size=(45, 35), mode="RGB", quality=80, subsampling=2, progressive=True),
    "grey_progressive_restart": dict(size=(40, 33), mode="L", quality=70, progressive=True, restart_marker_blocks=4),
}
expected = {}
for name, c in CASES.items():
    w, h = c.pop("size")
    mode = c.pop("mode")
    a = picture(w, h)
    path = os.path.join(HERE, f"jpeg_{name}.jpg")
    Image.fromarray(a if mode == "RGB" else a[..., 0], mode).save(path, **c)
    expected[name] = np.asarray(Image.open(path).convert("RGB"))
np.savez_compressed(os.path.join(HERE, "jpeg_expected.npz"), **expected)
print({k: v.shape for k, v in expected.items()})
