"""A/B of the first-layer kernel (variant 0 = one tile per work-group, 1 = persistent) in one process: time and bit-equality."""
import importlib, sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
import hip_helpers as H
rng = np.random.default_rng(0)
for (n, h, w, pool, act) in [(640, 256, 256, 1, 1), (64, 256, 256, 1, 1), (7, 48, 80, 1, 1), (3, 16, 16, 1, 2)]:
    x = vad.scoring.synth_frames_device(5, 0, n, h, w)
    wt = (rng.standard_normal((32, 3, 3, 3)) / 5).astype(np.float32); b = rng.standard_normal(32).astype(np.float32) * 0.1
    wp, bo = H.pack_conv3x3(wt, b)
    outs, times = [], []
    for variant in (0, 1):
        l.vad_debug_set_conv_variant(variant)
        out = torch.full((n, h // 2 if pool else h, w // 2 if pool else w, 32), float("nan"), device="cuda")
        for it in range(6):
            if it == 1: torch.cuda.synchronize(); t0 = time.perf_counter()
            vad.hip.check(l.vad_conv3x3_c3(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), out.data_ptr(), n, h, w, 32, act, pool, H.stream()))
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 5); outs.append(out)
    l.vad_debug_set_conv_variant(1)
    print(f"n={n} {h}x{w} pool={pool} act={act}: old {times[0]*1e6/n:.3f} us/frame, new {times[1]*1e6/n:.3f} us/frame, bit-identical {torch.equal(outs[0], outs[1])}")
