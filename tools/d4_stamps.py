"""Where one wave of the fused dec4 kernel spends its cycles (diagnostic build: dec4_fused.hip with -DVAD_D4_STAMPS linked
into libvad_hip_stamps.so by tools/build_d4_stamps.sh).  s_memtime stamps around the phases of one output row:
0 loop top -> fused block, 1 the fused block (64 MFMAs of this row + the combine of the previous one), 2 LDS writes + stores,
3 barrier."""
import ctypes, importlib, os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
os.environ["VAD_LIB"] = str(ROOT / "video-anomaly-detection_amd" / "libvad_hip_stamps.so")
sys.path.insert(0, str(ROOT))
hip = importlib.import_module("video-anomaly-detection_amd.hip")
l = hip.lib()
n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 128, 128
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = rnd(n, h, w, 32); wt = rnd(4096) * 0.05; bt = rnd(32); w2 = rnd(1024) * 0.05; b3 = rnd(4); img = rnd(n, 3, 2 * h, 2 * w)
parts = torch.empty(n * l.vad_dec4_score_partials(2 * h, 2 * w), device="cuda")
dbg = torch.zeros(1024 * 4 * 12, dtype=torch.int64, device="cuda")
l.vad_debug_set_dec4_stamps.argtypes = [ctypes.c_void_p]
fn = lambda: l.vad_dec4_score(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), w2.data_ptr(), b3.data_ptr(), img.data_ptr(),
                              parts.data_ptr(), None, None, n, h, w, hip.current_stream())
for _ in range(3):
    hip.check(fn())
torch.cuda.synchronize()
l.vad_debug_set_dec4_stamps(dbg.data_ptr())
hip.check(fn())
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 12)
d = d[d[:, 11] > 0]
per = d[:, :4] / d[:, 11:12]
names = ["loop top", "fused block (64 mfma)", "acc drain + lds write + stores", "barrier"]
print(f"waves {len(d)}  phases/wave {d[:, 11].mean():.0f}  clock {d[:, 10].mean() / 1e6:.2f} GHz")
for i, nm in enumerate(names):
    print(f"  {nm:24s} mean {per[:, i].mean():8.0f}  min {per[:, i].min():8.0f}  max {per[:, i].max():8.0f} cycles/phase")
print(f"  total {per.sum(axis=1).mean():.0f} cycles/phase")
# which blocks share a CU
raw = dbg.cpu().numpy().reshape(-1, 12)
nb = 512
hw = raw[0:nb * 4:4, 8]; xcc = raw[0:nb * 4:4, 9] & 0xF
key = {}
for b in range(nb):
    h = int(hw[b])
    k = (int(xcc[b]), (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xF)   # xcc, se, sh, cu
    key.setdefault(k, []).append((b, h & 0xF, (h >> 4) & 3, (h >> 16) & 0xF))
print("CUs used", len(key), " blocks per CU", sorted(set(len(v) for v in key.values())))
for k in list(key)[:12]:
    print(k, key[k], "  (block, wave_id, simd_id, tg_id)")
