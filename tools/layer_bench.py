"""Time one layer kernel of libvad_hip.so in isolation (developer tool; not part of the product path).

    python tools/layer_bench.py conv3x3 --n 32 --h 256 --w 256 --cin 32 --cout 32 --pool 1
"""
import argparse
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
hip = importlib.import_module("video-anomaly-detection_amd.hip")

IMAGE_LAYERS = [  # name, kind, h, cin, cout, pool
    ("enc1", "c3fused", 256, 3, 32, 1), ("enc2.0", "conv3x3", 128, 32, 64, 0),
    ("enc2.3", "conv3x3", 128, 64, 64, 1), ("enc3.0", "conv3x3", 64, 64, 128, 0), ("enc3.3", "conv3x3", 64, 128, 128, 1),
    ("enc4.0", "conv3x3", 32, 128, 256, 0), ("enc4.3", "conv3x3", 32, 256, 256, 1), ("dec1.0", "convt", 16, 256, 128, 0),
    ("dec1.3", "conv3x3", 32, 128, 128, 0), ("dec2.0", "convt", 32, 128, 64, 0), ("dec2.3", "conv3x3", 64, 64, 64, 0),
    ("dec3.0", "convt", 64, 64, 32, 0), ("dec3.3", "conv3x3", 128, 32, 32, 0), ("dec4.0", "convt", 128, 32, 32, 0),
    ("dec4.3", "tail", 256, 32, 3, 0),
]


def run(kind, n, h, w, cin, cout, pool, iters, warm=3, prec=0):
    l = hip.lib()
    g = torch.Generator(device="cuda").manual_seed(1)
    s = hip.current_stream()
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=g)
    if kind == "c3":
        x = rnd(n, 3, h, w); wt = rnd(28 * cout) * 0.2; b = rnd(cout)
        out = torch.empty(n, h // (2 if pool else 1), w // (2 if pool else 1), cout, device="cuda")
        fn = lambda: l.vad_conv3x3_c3(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), n, h, w, cout, 1, pool, s)
        flop = 2.0 * n * h * w * 27 * cout
        byts = 4.0 * (x.numel() + out.numel())
    elif kind == "c3fused":
        x = rnd(n, 3, h, w); w0 = rnd(28 * 32) * 0.2; b0 = rnd(32); w1 = rnd(9 * 32 * 32) * 0.05; b1 = rnd(32)
        out = torch.empty(n, h // 2, w // 2, 32, device="cuda")
        fn = lambda: l.vad_conv3x3_c3_fused(x.data_ptr(), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                            out.data_ptr(), n, h, w, prec, s)
        flop = 2.0 * n * h * w * (27 * 32 + 9 * 32 * 32)
        byts = 4.0 * (x.numel() + out.numel())
    elif kind == "conv3x3":
        x = rnd(n, h, w, cin); wt = rnd(9 * cin * cout) * 0.05; b = rnd(cout)
        out = torch.empty(n, h // (2 if pool else 1), w // (2 if pool else 1), cout, device="cuda")
        fn = lambda: l.vad_conv3x3(x.data_ptr(), 0, wt.data_ptr(), b.data_ptr(), out.data_ptr(), 0, n, h, w, cin, cout, 1, pool, prec, s)
        flop = 2.0 * n * h * w * 9 * cin * cout
        byts = 4.0 * (x.numel() + out.numel())
    elif kind == "wino":                    # Winograd F(2x2,3x3) form of conv3x3 (random, unpacked weights: timing only)
        x = rnd(n, h, w, cin); wt = rnd(16 * cin * cout) * 0.05; b = rnd(cout)
        out = torch.empty(n, h // (2 if pool else 1), w // (2 if pool else 1), cout, device="cuda")
        fn = lambda: l.vad_conv3x3_wino(x.data_ptr(), 0, wt.data_ptr(), b.data_ptr(), out.data_ptr(), 0, n, h, w, cin, cout, 1, pool, s)
        flop = 2.0 * n * h * w * 9 * cin * cout          # ALGORITHMIC (direct-convolution) FLOPs; executed MFMA FLOPs are 16/36 of them
        byts = 4.0 * (x.numel() + out.numel())
    elif kind == "convt":
        x = rnd(n, h, w, cin); wt = rnd(4 * cin * cout) * 0.05; b = rnd(cout)
        out = torch.empty(n, 2 * h, 2 * w, cout, device="cuda")
        fn = lambda: l.vad_convt2x2(x.data_ptr(), 0, wt.data_ptr(), b.data_ptr(), out.data_ptr(), 0, n, h, w, cin, cout, 2, prec, s)
        flop = 2.0 * n * h * w * 4 * cin * cout
        byts = 4.0 * (x.numel() + out.numel())
    elif kind == "lstm":
        hid = cout
        x = rnd(n, h, w, cin); hp = rnd(n, h, w, hid) * 0.3; cp = rnd(n, h, w, hid)
        wt = rnd(9 * (cin + hid) * 4 * hid) * 0.02; b = rnd(4 * hid)
        ho = torch.empty(n, h, w, hid, device="cuda"); co = torch.empty(n, h, w, hid, device="cuda")
        fn = lambda: l.vad_convlstm_step(x.data_ptr(), 0, hp.data_ptr(), 0, cp.data_ptr(), wt.data_ptr(), b.data_ptr(),
                                         ho.data_ptr(), 0, co.data_ptr(), n, h, w, cin, hid, prec, s)
        flop = 2.0 * n * h * w * 9 * (cin + hid) * 4 * hid
        byts = 4.0 * (x.numel() + 4 * ho.numel())
    elif kind == "tail":
        x = rnd(n, h, w, 32); wt = rnd(8 * 108) * 0.05; b = rnd(4); img = rnd(n, 3, h, w)
        nparts = l.vad_score_partials(0, h, w)
        parts = torch.empty(n * nparts, device="cuda")
        fn = lambda: l.vad_conv3x3_to3_score(x.data_ptr(), wt.data_ptr(), b.data_ptr(), img.data_ptr(), parts.data_ptr(),
                                             None, None, n, h, w, 32, s)
        flop = 2.0 * n * h * w * 9 * 32 * 3
        byts = 4.0 * (x.numel() + img.numel())
    elif kind == "tailt":
        x = rnd(n, h, w, 32); wt = rnd(32 * 12) * 0.05; b = rnd(4); img = rnd(n, 3, 2 * h, 2 * w)
        nparts = l.vad_score_partials(1, 2 * h, 2 * w)
        parts = torch.empty(n * nparts, device="cuda")
        fn = lambda: l.vad_convt2x2_to3_score(x.data_ptr(), wt.data_ptr(), b.data_ptr(), img.data_ptr(), parts.data_ptr(),
                                              None, None, n, h, w, 32, 0, 0, s)
        flop = 2.0 * n * h * w * 4 * 32 * 3
        byts = 4.0 * (x.numel() + img.numel())
    elif kind == "dec4":                    # fused dec4.0 + dec4.3 + score; h, w = INPUT map size (random, unpacked weights: timing only)
        x = rnd(n, h, w, 32); wt = rnd(4096) * 0.05; bt = rnd(32); w2 = rnd(1024) * 0.05; b3 = rnd(4); img = rnd(n, 3, 2 * h, 2 * w)
        parts = torch.empty(n * l.vad_dec4_score_partials(2 * h, 2 * w), device="cuda")
        fn = lambda: l.vad_dec4_score(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), w2.data_ptr(), b3.data_ptr(), img.data_ptr(),
                                      parts.data_ptr(), None, None, n, h, w, s)
        flop = 2.0 * n * h * w * (4 * 32 * 32 + 4 * 32 * 32)
        byts = 4.0 * (x.numel() + img.numel())
    else:
        raise SystemExit(f"unknown kind {kind}")
    for _ in range(warm):
        hip.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flop / ms / 1e9, byts / ms / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kind")
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=0)
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--pool", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", type=int, default=0, help="0 exact fp32, 1 split fp16")
    ap.add_argument("--stamps", type=int, default=0, help="diagnostic build only (VAD_LIB=...stamps.so)")
    ap.add_argument("--variant", type=int, default=-1, help="conv kernel variant (0 one tile per WG, 1 persistent)")
    a = ap.parse_args()
    if a.variant >= 0:
        hip.lib().vad_debug_set_conv_variant(a.variant)
    dbg = None
    if a.stamps:
        import ctypes
        hip.lib().vad_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
        dbg = torch.zeros(1024 * 4 * 12, dtype=torch.int64, device="cuda")
        hip.lib().vad_debug_set_stamp_buffer(dbg.data_ptr())
    if a.kind == "winoab":                  # every 3x3 layer of the image path behind enc1: direct vs Winograd
        for name, kind, h, cin, cout, pool in IMAGE_LAYERS:
            if kind != "conv3x3":
                continue
            ms0, tf0, _ = run("conv3x3", a.n, h, h, cin, cout, pool, a.iters)
            ms1, tf1, _ = run("wino", a.n, h, h, cin, cout, pool, a.iters)
            print(f"{name:8s} {cin:3d}->{cout:3d} @{h:3d} pool={pool}: direct {ms0:7.4f} ms {tf0:6.1f} TF | winograd {ms1:7.4f} ms "
                  f"{tf1:6.1f} TF algorithmic ({tf1 * 16 / 36:6.1f} executed) | speed-up {ms0 / ms1:5.2f}x", flush=True)
    elif a.kind == "image":
        tot = 0.0
        for name, kind, h, cin, cout, pool in IMAGE_LAYERS:
            ms, tf, gbs = run(kind, a.n, h, h, cin, cout, pool, a.iters, prec=a.precision)
            tot += ms
            print(f"{name:8s} {kind:8s} {ms * 1e3 / a.n:8.2f} us/frame  {tf:7.2f} TFLOP/s  {gbs:8.1f} GB/s")
        print(f"total {tot * 1e3 / a.n:.2f} us/frame -> {a.n / tot * 1e3:.0f} frames/s")
    else:
        ms, tf, gbs = run(a.kind, a.n, a.h, a.w or a.h, a.cin, a.cout, a.pool, a.iters, prec=a.precision)
        print(f"{a.kind} n={a.n} {a.h}x{a.w or a.h} {a.cin}->{a.cout} pool={a.pool}: {ms:.4f} ms  {tf:.2f} TFLOP/s  {gbs:.1f} GB/s")
        if dbg is not None:
            d = dbg.cpu().numpy().reshape(-1, 12)
            d = d[d[:, 11] > 0]
            per = d[:, :10] / d[:, 11:12]
            names = ["epi->top", "barrier1", "lds write", "barrier2", "issue+first loads", "36 steps", "epilogue",
                     "fused: issue x", "fused: c3 stage", "fused: barrier3"]
            print(f"waves {len(d)}, stages/wave {d[:, 11].mean():.1f}, shader clock {d[:, 10].mean() / 1e6:.3f} GHz, "
                  f"sum of segments {per.sum(axis=1).mean():.0f} cycles/stage")
            for i, nm in enumerate(names):
                print(f"  {nm:20s} mean {per[:, i].mean():9.0f}  p10 {np.percentile(per[:, i], 10):9.0f}  p90 {np.percentile(per[:, i], 90):9.0f} cycles/stage")
