"""Throughput of the native training step (row f-1): VideoTrainer.step on B clips x T frames of HxW, exact fp32.
    python tools/train_bench.py --clips 8 --t 10 --hw 256 --steps 5 --warmup 2
Prints one JSON line (frames/s trained; algorithmic FLOP = 3 x the forward's 3.01 GFLOP/frame at 256x256, T=10:
forward + data gradients + weight gradients).  Not the headline metric (bench.py is); used with
`rocprofv3 --kernel-trace --stats` to rank the training kernels."""
import argparse, importlib, json, sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
vad = importlib.import_module("video-anomaly-detection_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=8); ap.add_argument("--t", type=int, default=10); ap.add_argument("--hw", type=int, default=256)
ap.add_argument("--latent", type=int, default=128); ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--precision", choices=["fp32", "split", "bf16", "bf16_operands", "winograd"], default="fp32",
                help="split / bf16_operands: 3x3 / transposed convs on split-fp16 / bf16 operands; bf16: activation and gradient tensors bf16 in HBM too")
a = ap.parse_args()
torch.cuda.set_device(0)
m = vad.VideoAutoencoder(in_channels=3, latent_dim=a.latent, lstm_hidden_dim=a.latent, lstm_num_layers=a.layers)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, 5).items()}, strict=True)
m = m.cuda(); tr = vad.VideoTrainer(m, precision=a.precision)
x = vad.scoring.synth_frames_device(0xC0FFEE + 4, 0, a.clips * a.t, a.hw, a.hw, 3, torch.device("cuda", 0)).view(a.clips, a.t, 3, a.hw, a.hw)
losses = []
for _ in range(a.warmup): losses.append(float(tr.step(x)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): loss = tr.step(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
losses.append(float(loss))
frames = a.clips * a.t
fwd_flop_per_frame = 3011510272.0 * (a.hw / 256.0) ** 2 if (a.latent, a.layers, a.t) == (128, 2, 10) else None
print(json.dumps({"metric": "training frames/s (native step, %s)" % {"fp32": "exact fp32", "split": "split-fp16 convolutions, rest fp32", "bf16": "bf16 tensors + bf16 MFMA operands, fp32 arithmetic / statistics / master weights", "bf16_operands": "bf16 convolution operands from fp32 tensors, rest fp32", "winograd": "fp32 everywhere, 3x3 forward / data-gradient convolutions as Winograd F(2x2,3x3)"}[a.precision], "value": round(frames / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                  "clips": a.clips, "t": a.t, "hw": a.hw, "latent": a.latent, "layers": a.layers, "steps": a.steps,
                  "workspace_GiB": round(tr._ws.numel() / 2**30, 2), "losses_first_last": [losses[0], losses[-1]],
                  "algorithmic_tflops": round(3 * fwd_flop_per_frame * frames / dt / 1e12, 2) if fwd_flop_per_frame else None}))
