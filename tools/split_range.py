import sys, importlib, numpy as np, torch
from pathlib import Path; R = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
vad = importlib.import_module("video-anomaly-detection_amd")
import hip_helpers as H
rng = np.random.default_rng(0)
x = rng.standard_normal((4, 64, 32, 32)).astype(np.float32)
w = (rng.standard_normal((64, 64, 3, 3)) * 0.05).astype(np.float32)
b = np.zeros(64, np.float32)
for scale in (1.0, 1e-3, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9):
    xs = (x * scale).astype(np.float32)
    H.PRECISION = 0; ref = np.asarray(H.conv3x3(xs, w, b))
    H.PRECISION = 1; got = np.asarray(H.conv3x3(xs, w, b))
    print(scale, "max rel err (vs max)", float(np.abs(got - ref).max() / np.abs(ref).max()))
