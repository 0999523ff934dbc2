"""SSIM / combined criterion, forward + backward: fused HIP passes (losses.CombinedLoss on GPU tensors) vs the stock torch
composition on the same GPU (the reference's utils/losses.py arithmetic run by PyTorch-ROCm)."""
import importlib, json, sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vad = importlib.import_module("video-anomaly-detection_amd")
crit = vad.CombinedLoss(alpha=0.5)
res = {}
for b in (16, 128):
    t = vad.scoring.synth_frames_device(3, 0, b)
    p = (t + 0.1 * torch.randn_like(t)).clamp(-1, 1)
    def run(fn):
        for it in range(6):
            if it == 1: torch.cuda.synchronize(); t0 = time.perf_counter()
            x = p.clone().requires_grad_(True); fn(x).backward()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5
    def composition(x):          # force the torch path: same module code, target marked as needing a gradient
        return crit(x, t.clone().requires_grad_(True))
    hip_t, torch_t = run(lambda x: crit(x, t)), run(composition)
    res[f"batch{b}"] = {"hip_ms": round(hip_t * 1e3, 3), "torch_ms": round(torch_t * 1e3, 3), "speedup": round(torch_t / hip_t, 1),
                        "hip_GBps_algorithmic": round(b * 3 * 65536 * 4 * 3 / hip_t / 1e9, 1)}
print(json.dumps(res))
