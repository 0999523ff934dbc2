"""Gradient stream of a REAL native step, stopped after each decoder backward stage, against float64 autograd
intermediates (retain_grad): d u_j (space-to-depth, g2) and d(input of stage j) (g0)."""
import sys, importlib, ctypes as C, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_synthetic
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
latent, layers, b, t, wseed = 32, 3, 1, 4, 43
def rel(g, r):
    g, r = np.asarray(g, np.float64), np.asarray(r, np.float64); return float(np.abs(g-r).max()/max(np.abs(r).max(),1e-12))
for hw in [int(a) for a in sys.argv[1:]] or [112, 64]:
    x = torch.from_numpy(vad.synth.clips(wseed+100, 0, b, t, 3, hw, hw)); N = b*t; h16 = hw//16
    ref = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, ref, wseed); ref = ref.double().train()
    xe = x.double().view(N, 3, hw, hw); cur = xe
    for mod in ref.encoder.encoder: cur = mod(cur)
    hs, _ = ref.convlstm(cur.view(b, t, latent, h16, h16)); hq = hs.reshape(N, latent, h16, h16); hq.retain_grad()
    dec = list(ref.decoder.decoder); cur = hq; us, rs = [], []
    for j in range(3):
        cur = dec[3*j](cur); cur.retain_grad(); us.append(cur); cur = dec[3*j+1](cur); cur = dec[3*j+2](cur); cur.retain_grad(); rs.append(cur)
    rec = torch.tanh(dec[9](cur)); F.mse_loss(rec, xe).backward()
    decC = [latent, 128, 64, 32]
    print(f"hw={hw} (H/16={h16})")
    for j in (2, 1, 0):
        m = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
        load_synthetic(vad, m, wseed); m = m.cuda(); tr = vad.VideoTrainer(m)
        l.vad_debug_set_train_stop(j)
        try:
            tr.forward_backward(x.cuda()); torch.cuda.synchronize()
        finally:
            l.vad_debug_set_train_stop(-1)
        out = (C.c_longlong * 64)(); n = l.vad_vid_train_debug_layout(b, t, hw, hw, latent, latent, layers, out, 64); o = list(out[:n])
        g0o, g1o, g2o = o[-4], o[-3], o[-2]; W = tr._ws.view(torch.float32)
        ci, co, hj = decC[j], decC[j+1], h16 << j
        du = W[g2o:g2o+N*hj*hj*4*co].view(N,hj,hj,2,2,co).permute(0,5,1,3,2,4).reshape(N,co,2*hj,2*hj).cpu().numpy()
        din = W[g0o:g0o+N*hj*hj*ci].view(N,hj,hj,ci).permute(0,3,1,2).cpu().numpy()
        din_ref = (hq if j == 0 else rs[j-1]).grad
        dw = dict(m.named_parameters())[f"decoder.decoder.{3*j}.weight"].grad.cpu().numpy()
        print(f"   stop after j={j}: d u{j} {rel(du, us[j].grad):.2e}   d in {rel(din, din_ref):.2e}   dW {rel(dw, dec[3*j].weight.grad):.2e}")
