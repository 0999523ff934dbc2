"""Stand-alone timing of the weight-gradient GEMMs of the 32-clip training step (tools/wgrad_bench.py [precision]): every layer
shape, every kernel form of the split-fp16 mode (vad_debug_set_wgrad_split 0 / 1 / 2), torch.cuda events around 20 launches."""
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
for name, n, h, w, cin, ncols, taps, layout in shapes:
    a = torch.randn(n, h, w, cin, device="cuda")
    g = torch.randn(n, h, w, ncols, device="cuda") * 0.05
    dw = torch.empty(taps * cin * ncols, device="cuda")
    ws = torch.empty(l.vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols), device="cuda")
    flops = 2.0 * taps * cin * ncols * n * h * w
    line = [f"{name:7s} {flops / 1e9:6.1f} GF"]
    for prec, mode in ((0, 2), (1, 1), (1, 2), (2, 2)):
        l.vad_debug_set_wgrad_split(mode)
        for _ in range(3):
            vad.hip.check(l.vad_conv_wgrad(a.data_ptr(), g.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cin, ncols, taps, layout, prec, s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            l.vad_conv_wgrad(a.data_ptr(), g.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cin, ncols, taps, layout, prec, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line.append(f"{['fp32', 'split', 'bf16op'][prec]}{'' if prec != 1 else '/' + str(mode)} {ms * 1e3:7.1f} us {flops / ms / 1e9:6.1f} TF")
    l.vad_debug_set_wgrad_split(2)
    print("  ".join(line))
