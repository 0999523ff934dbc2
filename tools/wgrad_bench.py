"""Stand-alone timing of the weight-gradient GEMMs of the 32-clip training step (tools/wgrad_bench.py [precision]): every layer
shape, every kernel form of the split-fp16 and bf16-tensor modes (vad_debug_set_wgrad_split / vad_debug_set_wgrad_pairs), torch.cuda events around 20 launches."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vad = importlib.import_module("video-anomaly-detection_amd")
l = vad.hip.lib()
N = 320
shapes = [("enc.4", N, 128, 128, 32, 64, 9, 0), ("enc.8", N, 64, 64, 64, 128, 9, 0), ("enc.12", N, 32, 32, 128, 128, 9, 0),
          ("lstm", N, 16, 16, 256, 512, 9, 0), ("convT0", N, 16, 16, 128, 512, 1, 1), ("convT1", N, 32, 32, 128, 256, 1, 1),
          ("convT2", N, 64, 64, 64, 128, 1, 1), ("to3", N, 128, 128, 32, 32, 1, 3)]
s = vad.hip.current_stream()
import os
ONLY, FORMS = os.environ.get("WGB_ONLY", ""), os.environ.get("WGB_FORMS", "")      # e.g. WGB_ONLY=enc.8 WGB_FORMS=bf16t/3,split/3 (profiling)
for name, n, h, w, cin, ncols, taps, layout in shapes:
    if ONLY and name not in ONLY.split(","):
        continue
    a = torch.randn(n, h, w, cin, device="cuda")
    g = torch.randn(n, h, w, ncols, device="cuda") * 0.05
    dw = torch.empty(taps * cin * ncols, device="cuda")
    ws = torch.empty(l.vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols), device="cuda")
    flops = 2.0 * taps * cin * ncols * n * h * w
    line = [f"{name:7s} {flops / 1e9:6.1f} GF"]
    a16, g16 = a.to(torch.bfloat16), g.to(torch.bfloat16)
    for label, prec, split_mode, pairs_mode in (("fp32/0", 0, 3, 3), ("fp32", 0, 3, 3), ("split/1", 1, 1, 3), ("split/2", 1, 2, 3), ("split/3", 1, 3, 3), ("bf16op", 2, 3, 3),
                                                ("bf16t/1", 3, 3, 1), ("bf16t/2", 3, 3, 2), ("bf16t/3", 3, 3, 3)):
        if FORMS and label not in FORMS.split(","):
            continue
        l.vad_debug_set_wgrad_split(split_mode); l.vad_debug_set_wgrad_pairs(pairs_mode); l.vad_debug_set_wgrad_ring_f32(0 if label == "fp32/0" else 1)
        pa, pg = (a16, g16) if prec == 3 else (a, g)
        call = lambda: l.vad_conv_wgrad(pa.data_ptr(), pg.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cin, ncols, taps, layout, prec, s)
        for _ in range(3):
            vad.hip.check(call())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line.append(f"{label} {ms * 1e3:6.1f} us {flops / ms / 1e9:5.0f} TF")
    l.vad_debug_set_wgrad_split(3); l.vad_debug_set_wgrad_pairs(3)
    print(" | ".join(line))
