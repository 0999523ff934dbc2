"""Decoder-stage backward ops at the exact shapes where the training step deviates (hw=112: j=2 is 64->32 at 28x28)."""
import sys, importlib, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hip_helpers as H
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
def ws(n): return torch.empty(max(int(n),1), device="cuda")
def rel(g, r): return float(np.abs(np.asarray(g,np.float64)-np.asarray(r,np.float64)).max()/max(np.abs(r).max(),1e-12))
# (a) BatchNorm+ReLU backward, dense vs space-to-depth vs torch
for (n,h,w,c) in [(4,56,56,32),(4,24,24,32),(4,32,32,32),(4,28,28,64),(4,14,14,128)]:
    rng = np.random.default_rng(h+c)
    y = rng.standard_normal((n,c,h,w)).astype(np.float32); dout = rng.standard_normal((n,c,h,w)).astype(np.float32)
    gamma, beta = rng.uniform(0.5,1.5,c).astype(np.float32), (rng.standard_normal(c)*0.1).astype(np.float32)
    yt = torch.from_numpy(y).double().requires_grad_(True)
    z = F.relu(F.batch_norm(yt, None, None, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), training=True, eps=1e-5))
    (z*torch.from_numpy(dout).double()).sum().backward()
    yd, gd, bd, dd = H.nhwc(y), H.dev(gamma), H.dev(beta), H.nhwc(dout)
    stats, w_ = ws(2*c), ws(l.vad_chan_ws_floats(n*h*w, c))
    vad.hip.check(l.vad_bn_stats(yd.data_ptr(), n*h*w, c, 1e-5, 0.1, stats.data_ptr(), None, None, w_.data_ptr(), H.stream()))
    res = []
    for s2d in (0, 1):
        dz = torch.empty(n,h,w,c, device="cuda"); dy = torch.full((n*h*w*c,), float("nan"), device="cuda")
        dg, db, ks = ws(c), ws(c), ws(2*c)
        vad.hip.check(l.vad_bn_act_pool_bwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dd.data_ptr(), 0,0,0,0,
                      dz.data_ptr(), dy.data_ptr(), s2d, dg.data_ptr(), db.data_ptr(), ks.data_ptr(), w_.data_ptr(), n,h,w,c, 2, 0, H.stream()))
        if s2d: dy = dy.view(n,h//2,w//2,2,2,c).permute(0,1,3,2,4,5).reshape(n,h,w,c)
        else: dy = dy.view(n,h,w,c)
        res.append(rel(H.to_nchw(dy), yt.grad.numpy()))
    print(f"bn bwd {n}x{h}x{w}x{c}: dense {res[0]:.2e}  s2d {res[1]:.2e}")
# (b) convT weight / data gradient
for (n,h,w,cin,cout) in [(4,28,28,64,32),(4,14,14,128,64),(4,7,7,32,128),(4,12,12,64,32),(4,6,6,128,64),(4,16,16,64,32),(1,28,28,64,32),(4,28,4,64,32),(4,4,28,64,32)]:
    rng = np.random.default_rng(h*3+cin)
    a = rng.standard_normal((n,cin,h,w)).astype(np.float32); g = rng.standard_normal((n,cout,2*h,2*w)).astype(np.float32)
    wt = (rng.standard_normal((cin,cout,2,2))/np.sqrt(cin)).astype(np.float32)
    at, wtt = torch.from_numpy(a).double().requires_grad_(True), torch.from_numpy(wt).double().requires_grad_(True)
    (F.conv_transpose2d(at, wtt, stride=2)*torch.from_numpy(g).double()).sum().backward()
    ad = H.nhwc(a)
    gs = torch.from_numpy(g).permute(0,2,3,1).reshape(n,h,2,w,2,cout).permute(0,1,3,2,4,5).reshape(n,h,w,4*cout).contiguous().cuda()
    dw = torch.full((cin,cout,2,2), float("nan"), device="cuda"); w_ = ws(l.vad_conv_wgrad_ws_floats(n,h,1,cin,4*cout))
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), gs.data_ptr(), dw.data_ptr(), w_.data_ptr(), n,h,w,cin,4*cout,1,1,H.stream()))
    wd = H.dev(wt); dgr = ws(l.vad_pack_conv1x1_floats(cin, 4*cout))
    vad.hip.check(l.vad_train_pack_convt2x2(wd.data_ptr(), cin, cout, None, dgr.data_ptr(), H.stream()))
    zi = torch.zeros(cin, device="cuda"); da = torch.full((n,h,w,cin), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv1x1(gs.data_ptr(), dgr.data_ptr(), zi.data_ptr(), da.data_ptr(), n*h*w, 4*cout, cin, H.stream()))
    print(f"convT bwd {n}x{h}x{w} {cin}->{cout}: dW {rel(dw.cpu().numpy(), wtt.grad.numpy()):.2e}  dA {rel(H.to_nchw(da), at.grad.numpy()):.2e}")
