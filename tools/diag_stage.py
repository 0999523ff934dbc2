"""Per-tensor deviation of first-step gradients from float64, in BACKWARD order, to locate the stage where the odd-size
error enters (open item in tests/test_hip_train_step.py)."""
import sys, importlib, numpy as np, torch
from pathlib import Path
_R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(_R / "tests")); sys.path.insert(0, str(_R))
import test_hip_train_step as T
from conftest import load_synthetic
vad = importlib.import_module("video-anomaly-detection_amd")
latent, layers, b, t, wseed = 32, 3, 1, 4, 43
for hw in (112, 48, 64):
    x = torch.from_numpy(vad.synth.clips(wseed + 100, 0, b, t, 3, hw, hw))
    truth = T._fp64_grads(vad, latent, layers, wseed, x)
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, m, wseed); m = m.cuda()
    tr = vad.VideoTrainer(m, lr=T.LR, weight_decay=T.WD)
    loss, _ = tr.forward_backward(x.cuda())
    m64 = vad.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    load_synthetic(vad, m64, wseed); m64 = m64.double().train()
    l64 = float(torch.nn.MSELoss()(m64(x.double()), x.double()))
    print(f"hw={hw} H/16={hw//16}: loss gpu {float(loss):.9f} fp64 {l64:.9f} rel {abs(float(loss)-l64)/l64:.2e}")
    zero_true = T._bn_fed_biases(m)
    for k, p in reversed(list(m.named_parameters())):
        if k in zero_true: continue
        g = p.grad.detach().cpu().numpy(); tr_ = truth[k]
        scale = max(float(np.abs(tr_).max()), 1e-12)
        dev = np.abs(g - tr_) / scale
        print(f"   {k:34s} max {dev.max():.2e}  frac>1e-4 {np.mean(dev > 1e-4):.3f}  rel-norm {np.linalg.norm(g - tr_)/max(np.linalg.norm(tr_),1e-30):.2e}")
