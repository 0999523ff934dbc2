"""Where are the wrong entries?  Real native step at an odd H/16, stopped (a) right after the fused last layer and (b) after
decoder stage 2's backward; bad entries of d r2 / dz / d u2 vs float64 autograd, with their coordinates."""
import sys, importlib, ctypes as C, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_synthetic
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
latent, layers, b, t, wseed = 32, 3, 1, 4, 43
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 112
x = torch.from_numpy(vad.synth.clips(wseed+100, 0, b, t, 3, hw, hw)); N = b*t; h16 = hw//16; Hh = hw//2
ref = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
load_synthetic(vad, ref, wseed); ref = ref.double().train()
xe = x.double().view(N, 3, hw, hw); cur = xe
for mod in ref.encoder.encoder: cur = mod(cur)
hs, _ = ref.convlstm(cur.view(b, t, latent, h16, h16)); cur = hs.reshape(N, latent, h16, h16)
dec = list(ref.decoder.decoder); us, bnout, rs = [], [], []
for j in range(3):
    cur = dec[3*j](cur); cur.retain_grad(); us.append(cur); cur = dec[3*j+1](cur); cur.retain_grad(); bnout.append(cur); cur = dec[3*j+2](cur); cur.retain_grad(); rs.append(cur)
F.mse_loss(torch.tanh(dec[9](cur)), xe).backward()
def run(stop):
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, m, wseed); m = m.cuda(); tr = vad.VideoTrainer(m)
    l.vad_debug_set_train_stop(stop)
    try: tr.forward_backward(x.cuda()); torch.cuda.synchronize()
    finally: l.vad_debug_set_train_stop(-1)
    out = (C.c_longlong * 64)(); n = l.vad_vid_train_debug_layout(b, t, hw, hw, latent, latent, layers, out, 64); o = list(out[:n])
    return tr._ws.view(torch.float32), o[-4], o[-3], o[-2]
def report(name, got, ref_):
    got, ref_ = np.asarray(got, np.float64), ref_.detach().numpy()
    scale = np.abs(ref_).max(); dev = np.abs(got - ref_) / scale; bad = np.argwhere(dev > 1e-3)
    print(f"{name}: max {dev.max():.2e}, {len(bad)} of {dev.size} entries beyond 1e-3")
    if len(bad):
        for ax, nm in enumerate(("frame", "chan", "y", "x")):
            vals, cnt = np.unique(bad[:, ax], return_counts=True)
            print(f"      {nm}: {len(vals)} distinct; " + ", ".join(f"{v}:{c}" for v, c in list(zip(vals, cnt))[:16]) + (" ..." if len(vals) > 16 else ""))
        i = tuple(bad[np.argmax(dev[tuple(bad.T)])]); print(f"      worst at {i}: got {got[i]:.6e} want {ref_[i]:.6e}")
W, g0o, g1o, g2o = run(20)
report(f"[stop after last layer] d r2 ({N}x32x{Hh}x{Hh})", W[g0o:g0o+N*Hh*Hh*32].view(N,Hh,Hh,32).permute(0,3,1,2).cpu().numpy(), rs[2].grad)
W, g0o, g1o, g2o = run(2)
hj = h16 << 2
dz_ref = bnout[2].grad        # gradient wrt the BatchNorm output = routed/act' gradient
report(f"[stop after j=2] dz ({N}x32x{Hh}x{Hh})", W[g1o:g1o+N*Hh*Hh*32].view(N,Hh,Hh,32).permute(0,3,1,2).cpu().numpy(), dz_ref)
du = W[g2o:g2o+N*hj*hj*4*32].view(N,hj,hj,2,2,32).permute(0,5,1,3,2,4).reshape(N,32,2*hj,2*hj).cpu().numpy()
report(f"[stop after j=2] d u2", du, us[2].grad)
