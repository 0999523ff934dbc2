"""Isolate the odd-size error: conv3x3 forward / data-gradient / weight-gradient and convT ops vs torch at odd H, W,
and eval-mode video scoring vs the CPU oracle at odd H/16."""
import sys, importlib, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hip_helpers as H
from conftest import load_synthetic
from oracle import torch_oracle
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
def ws(n): return torch.empty(max(int(n),1), device="cuda")
def rel(g, r): return float(np.abs(np.asarray(g,np.float64)-np.asarray(r,np.float64)).max()/max(np.abs(r).max(),1e-12))
for (n,h,w,cin,cout) in [(2,2,2,64,128),(2,3,3,64,128),(2,5,5,64,128),(4,7,7,64,128),(2,4,4,64,128),(2,3,4,64,128),(2,4,3,64,128),(4,7,7,128,64)]:
    rng = np.random.default_rng(h*10+w)
    a = rng.standard_normal((n,cin,h,w)).astype(np.float32); g = rng.standard_normal((n,cout,h,w)).astype(np.float32)
    wt = (rng.standard_normal((cout,cin,3,3))/np.sqrt(9*cin)).astype(np.float32)
    at, wtt = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    out_ref = F.conv2d(at, wtt, padding=1); (out_ref*torch.from_numpy(g)).sum().backward()
    ad, gd, wd = H.nhwc(a), H.nhwc(g), H.dev(wt)
    dw = torch.full((cout,cin,3,3), float("nan"), device="cuda"); w_ = ws(l.vad_conv_wgrad_ws_floats(n,h,9,cin,cout))
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), gd.data_ptr(), dw.data_ptr(), w_.data_ptr(), n,h,w,cin,cout,9,0,H.stream()))
    fwd, dgr = ws(l.vad_pack_conv3x3_floats(cout,cin)), ws(l.vad_pack_conv3x3_floats(cin,cout))
    vad.hip.check(l.vad_train_pack_conv3x3(wd.data_ptr(), cout, cin, fwd.data_ptr(), dgr.data_ptr(), H.stream()))
    zi, zo = torch.zeros(cin, device="cuda"), torch.zeros(cout, device="cuda")
    da = torch.full((n,h,w,cin), float("nan"), device="cuda"); out = torch.full((n,h,w,cout), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv3x3(gd.data_ptr(),0,dgr.data_ptr(),zi.data_ptr(),da.data_ptr(),0,n,h,w,cout,cin,0,0,H.stream()))
    vad.hip.check(l.vad_conv3x3(ad.data_ptr(),0,fwd.data_ptr(),zo.data_ptr(),out.data_ptr(),0,n,h,w,cin,cout,0,0,H.stream()))
    print(f"conv3x3 {n}x{h}x{w} {cin}->{cout}: fwd {rel(H.to_nchw(out), out_ref.detach().numpy()):.2e}  dA {rel(H.to_nchw(da), at.grad.numpy()):.2e}  dW {rel(dw.cpu().numpy(), wtt.grad.numpy()):.2e}")
for hw in (32, 48, 64, 80, 112):
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=64, lstm_hidden_dim=64, lstm_num_layers=2)
    st = load_synthetic(vad, m, 77); m = m.cuda().eval()
    x = torch.from_numpy(vad.synth.clips(500+hw, 0, 2, 3, 3, hw, hw))
    with torch.no_grad(): got = m.get_reconstruction_error(x.cuda(), per_frame=True).cpu().numpy()
    want = torch_oracle.vid_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, x, 64, 2)["frame"].numpy()
    print(f"eval scoring {hw}x{hw} (H/16={hw//16}): max rel score err {rel(got, want):.2e}")
