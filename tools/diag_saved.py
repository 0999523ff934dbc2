"""After ONE real native training step, compare every forward buffer saved in the workspace with a float64 torch forward
(same weights, same clips).  Locates which saved activation the backward reads wrongly at odd H/16."""
import sys, importlib, ctypes as C, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_synthetic
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
latent, layers, b, t, wseed = 32, 3, 1, 4, 43
def rel(g, r):
    g, r = np.asarray(g, np.float64), np.asarray(r, np.float64); return float(np.abs(g-r).max()/max(np.abs(r).max(),1e-12))
for hw in [int(a) for a in sys.argv[1:]] or [112, 64]:
    x = torch.from_numpy(vad.synth.clips(wseed+100, 0, b, t, 3, hw, hw)); N = b*t; h16 = hw//16
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, m, wseed)
    ref = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, ref, wseed); ref = ref.double().train()
    # float64 forward with hooks
    acts = {}
    xe = x.double().view(N, 3, hw, hw); cur = xe; enc = list(ref.encoder.encoder)
    ys, as_ = [], []
    i = 0
    while i < len(enc):
        cur = enc[i](cur); ys.append(cur); cur = enc[i+1](cur); cur = enc[i+2](cur); cur = enc[i+3](cur); as_.append(cur); i += 4
    feats = as_[3].view(b, t, latent, h16, h16)
    hs, _ = ref.convlstm(feats)
    dec = list(ref.decoder.decoder); cur = hs.reshape(N, latent, h16, h16); us, rs = [], []
    for j in range(3):
        cur = dec[3*j](cur); us.append(cur); cur = dec[3*j+1](cur); cur = dec[3*j+2](cur); rs.append(cur)
    m = m.cuda(); tr = vad.VideoTrainer(m)
    tr.forward_backward(x.cuda()); torch.cuda.synchronize()
    out = (C.c_longlong * 64)()
    n = l.vad_vid_train_debug_layout(b, t, hw, hw, latent, latent, layers, out, 64); assert n > 0
    o = list(out[:n]); W = tr._ws.view(torch.float32)
    names = [f"y{k}" for k in range(4)] + [f"a{k}" for k in range(3)] + [f"st_e{k}" for k in range(4)] + [f"cat{q}" for q in range(layers)] + \
            [f"z{q}" for q in range(layers)] + [f"c{q}" for q in range(layers)] + ["hseq"] + [f"u{j}" for j in range(3)] + [f"r{j}" for j in range(3)] + \
            [f"st_d{j}" for j in range(3)] + ["dpre", "g0", "g1", "g2", "END"]
    off = dict(zip(names, o))
    print(f"hw={hw}: layout (floats): " + " ".join(f"{k}={v}" for k, v in off.items() if k[0] in "ur" or k.startswith("st_d") or k in ("hseq","dpre","g0","END")))
    def buf(name, shape): return W[off[name]:off[name]+int(np.prod(shape))].view(*shape).cpu().numpy()
    encC = [32, 64, 128, latent]
    for k in range(4):
        hk = hw >> k
        print(f"   y{k}   {rel(buf(f'y{k}', (N,hk,hk,encC[k])).transpose(0,3,1,2), ys[k].detach()):.2e}", end="")
        if k < 3: print(f"   a{k} {rel(buf(f'a{k}', (N,hk//2,hk//2,encC[k])).transpose(0,3,1,2), as_[k].detach()):.2e}", end="")
        st = buf(f"st_e{k}", (2, encC[k])); mu = ys[k].detach().mean((0,2,3)).numpy(); var = ys[k].detach().var((0,2,3), unbiased=False).numpy()
        print(f"   mean {rel(st[0], mu):.2e} invstd {rel(st[1], 1/np.sqrt(var+1e-5)):.2e}")
    hq = buf("hseq", (N, h16, h16, latent)).transpose(0,3,1,2)
    print(f"   hseq {rel(hq, hs.detach().reshape(N, latent, h16, h16)):.2e}")
    decC = [128, 64, 32]
    for j in range(3):
        hj = h16 << (j+1)
        st = buf(f"st_d{j}", (2, decC[j])); mu = us[j].detach().mean((0,2,3)).numpy(); var = us[j].detach().var((0,2,3), unbiased=False).numpy()
        print(f"   u{j} {rel(buf(f'u{j}', (N,hj,hj,decC[j])).transpose(0,3,1,2), us[j].detach()):.2e}   r{j} {rel(buf(f'r{j}', (N,hj,hj,decC[j])).transpose(0,3,1,2), rs[j].detach()):.2e}"
              f"   mean {rel(st[0], mu):.2e} invstd {rel(st[1], 1/np.sqrt(var+1e-5)):.2e}")
