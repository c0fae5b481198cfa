"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/resize_pil.npz with Pillow itself (the library the reference's
ToPILImage -> Resize(256) transform runs, util/data_utils.py:48-54): seeded uint8 frames and Image.resize(..., BILINEAR) of them at
the sizes torchvision's Resize(256) would ask for.  Run in the build container only:   python -B oracle/gen_resize_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.pil_resize import resized_hw  # noqa: E402

rec = {}
rng = np.random.RandomState(7)
for i, (h, w) in enumerate([(84, 84), (130, 100), (300, 260)]):
    # smooth structure + noise, so that a wrong tap shows up as more than a rounding flip
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 100 * np.sin(yy / 7.0 + i)[..., None] * np.cos(xx / 5.0)[..., None] * np.array([1.0, 0.6, -0.8])
    frame = np.clip(base + rng.randint(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)
    hr, wr = resized_hw(h, w, 256)
    out = np.asarray(Image.fromarray(frame).resize((wr, hr), Image.BILINEAR))
    rec["in%d" % i], rec["out%d" % i] = frame, out
import PIL
rec["pillow_version"] = np.array(PIL.__version__)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resize_pil.npz"), **rec)
print("written", {k: v.shape for k, v in rec.items()})
