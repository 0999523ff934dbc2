"""Is the single wrong entry of d u2 a ReLU branch flip?  BatchNorm output v at that pixel (GPU fp32 from the saved
workspace vs float64), and the number of sign disagreements of v per decoder stage / max-pool+LeakyReLU disagreements are
left for later.  Also reports how accurate the GPU's v is overall."""
import sys, importlib, ctypes as C, numpy as np, torch
from pathlib import Path
_R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(_R / "tests")); sys.path.insert(0, str(_R))
from conftest import load_synthetic
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
latent, layers, b, t, wseed = 32, 3, 1, 4, 43
for hw in [int(a) for a in sys.argv[1:]] or [112, 48, 80, 64]:
    x = torch.from_numpy(vad.synth.clips(wseed+100, 0, b, t, 3, hw, hw)); N = b*t; h16 = hw//16
    ref = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, ref, wseed); ref = ref.double().train()
    cur = x.double().view(N, 3, hw, hw)
    for mod in ref.encoder.encoder: cur = mod(cur)
    hs, _ = ref.convlstm(cur.view(b, t, latent, h16, h16)); cur = hs.reshape(N, latent, h16, h16)
    dec = list(ref.decoder.decoder); v64 = []
    for j in range(3):
        cur = dec[3*j](cur); cur = dec[3*j+1](cur); v64.append(cur.detach().clone().numpy()); cur = dec[3*j+2](cur)
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, m, wseed); m = m.cuda(); tr = vad.VideoTrainer(m)
    tr.forward_backward(x.cuda()); torch.cuda.synchronize()
    out = (C.c_longlong * 64)(); n = l.vad_vid_train_debug_layout(b, t, hw, hw, latent, latent, layers, out, 64); o = list(out[:n])
    names = [f"y{k}" for k in range(4)] + [f"a{k}" for k in range(3)] + [f"st_e{k}" for k in range(4)] + [f"cat{q}" for q in range(layers)] + \
            [f"z{q}" for q in range(layers)] + [f"c{q}" for q in range(layers)] + ["hseq"] + [f"u{j}" for j in range(3)] + [f"r{j}" for j in range(3)] + \
            [f"st_d{j}" for j in range(3)] + ["dpre", "g0", "g1", "g2", "END"]
    off = dict(zip(names, o)); W = tr._ws.view(torch.float32); decC = [128, 64, 32]
    print(f"hw={hw} (H/16={h16})")
    for j in range(3):
        hj, c = h16 << (j+1), decC[j]
        u = W[off[f"u{j}"]:off[f"u{j}"]+N*hj*hj*c].view(N,hj,hj,c)
        st = W[off[f"st_d{j}"]:off[f"st_d{j}"]+2*c]
        bn = dec[3*j+1]; ga, be = bn.weight.detach().float().cuda(), bn.bias.detach().float().cuda()
        v = (((u - st[:c]) * st[c:]) * ga + be).permute(0,3,1,2).cpu().numpy().astype(np.float64)       # the kernels' formula
        flips = np.argwhere((v > 0) != (v64[j] > 0))
        print(f"   stage {j}: |v - v64| max {np.abs(v - v64[j]).max():.2e} (|v| max {np.abs(v64[j]).max():.2f});  ReLU sign disagreements: {len(flips)} of {v.size}")
        for f in flips[:6]:
            i = tuple(f); print(f"        at {tuple(int(q) for q in i)}: gpu v = {v[i]:+.3e}   float64 v = {v64[j][i]:+.3e}")
