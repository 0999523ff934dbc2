"""PCIe-inclusive scoring rate (reported in DESIGN.md section 6, never bench.py's `value`): frames start in pinned host
memory, as the reference's loop hands them over (`images.to(device)`, evaluate.py:58), fp32 NCHW or raw uint8 NHWC."""
import importlib, json, sys, time
import numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vad = importlib.import_module("video-anomaly-detection_amd")
m = vad.ConvAutoencoder().cuda().eval()
m.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"        # "fp32" (default) | "split" | "winograd"
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, 1).items()})
n, reps = 512, 6
dev = vad.scoring.synth_frames_device(0xC0FFEE + 1, 0, n)
host_f32 = dev.cpu().pin_memory()
host_u8 = ((dev * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().pin_memory()
out = {}
with torch.no_grad():
    def run(host, overlap):
        copy_s, comp_s = torch.cuda.Stream(), torch.cuda.current_stream()
        bufs = [torch.empty_like(host, device="cuda") for _ in range(2)]
        evs = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        for it in range(reps + 1):
            if it == 1: torch.cuda.synchronize(); t0 = time.perf_counter()
            b = it & 1
            if overlap:
                with torch.cuda.stream(copy_s):
                    copy_s.wait_event(done[b])                 # buffer free again
                    bufs[b].copy_(host, non_blocking=True); evs[b].record(copy_s)
                comp_s.wait_event(evs[b])
            else:
                bufs[b].copy_(host, non_blocking=True)
            s = m.get_reconstruction_error(bufs[b]); done[b].record(comp_s)
        torch.cuda.synchronize()
        return n * reps / (time.perf_counter() - t0)
    t0 = time.perf_counter(); torch.cuda.synchronize()
    out["resident_f32"] = None
    for name, host in (("f32", host_f32), ("u8", host_u8)):
        out[f"{name}_copy_then_score"] = round(run(host, False), 1)
        out[f"{name}_overlapped"] = round(run(host, True), 1)
    for _ in range(2): m.get_reconstruction_error(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): m.get_reconstruction_error(dev)
    torch.cuda.synchronize(); out["resident_f32"] = round(n * reps / (time.perf_counter() - t0), 1)
    g = host_f32.numel() * 4 / 1e9
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): dev.copy_(host_f32, non_blocking=True)
    torch.cuda.synchronize(); out["h2d_GBps_pinned"] = round(5 * g / (time.perf_counter() - t0), 1)
out["precision"] = m.precision
print(json.dumps(out))
