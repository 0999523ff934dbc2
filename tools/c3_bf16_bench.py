"""First-layer forward of the bf16-tensor training step stand-alone (320 frames of 256x256): tools/c3_bf16_bench.py"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
vad = importlib.import_module("video-anomaly-detection_amd")
import hip_helpers as H
l = vad.hip.lib()
n, h, w = 320, 256, 256
x = torch.rand(n, 3, h, w, device="cuda") * 2 - 1
rng = np.random.default_rng(0)
wp, bo = H.pack_conv3x3((rng.standard_normal((32, 3, 3, 3)) * 0.2).astype(np.float32), np.zeros(32, np.float32))
o16 = torch.empty(n, h, w, 32, dtype=torch.bfloat16, device="cuda")
o32 = torch.empty(n, h, w, 32, device="cuda")
s = vad.hip.current_stream()
for name, fn in (("bf16 operands, bf16 out", lambda: l.vad_conv3x3_c3_bf16op(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), o16.data_ptr(), n, h, w, 32, s)),
                 ("fp32 operands, bf16 out", lambda: l.vad_conv3x3_c3_bf16(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), o16.data_ptr(), n, h, w, 32, s)),
                 ("fp32 operands, fp32 out", lambda: l.vad_conv3x3_c3(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), o32.data_ptr(), n, h, w, 32, 0, 0, s))):
    for _ in range(3): vad.hip.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
