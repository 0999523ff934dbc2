"""Replay the decoder's backward at an odd H/16 with the public op-level C calls, in the order train_step.hip issues
them, on activations computed by torch; compare after every call with float64 autograd.  Clean replay => the defect is in
train_step.hip's plumbing; dirty => names the call."""
import sys, importlib, numpy as np, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hip_helpers as H
vad = importlib.import_module("video-anomaly-detection_amd"); l = vad.hip.lib()
def ws(n): return torch.empty(max(int(n),1), device="cuda")
def rel(g, r):
    g, r = np.asarray(g, np.float64), np.asarray(r, np.float64); return float(np.abs(g-r).max()/max(np.abs(r).max(),1e-12))
def nhwc(t): return t.detach().float().permute(0,2,3,1).contiguous().cuda()
torch.manual_seed(0)
N, h16, L = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 7, 32
C = [L, 128, 64, 32]
convs = [nn.ConvTranspose2d(C[j], C[j+1], 2, 2).double() for j in range(3)]
bns = [nn.BatchNorm2d(C[j+1]).double() for j in range(3)]
for b in bns: nn.init.uniform_(b.weight, 0.5, 1.5); nn.init.normal_(b.bias, 0, 0.1)
last = nn.ConvTranspose2d(32, 3, 2, 2).double()
hseq = torch.randn(N, L, h16, h16, dtype=torch.float64, requires_grad=True)
x = torch.rand(N, 3, 16*h16, 16*h16, dtype=torch.float64)*2-1
u, r, cur = [], [], hseq
for j in range(3):
    uj = convs[j](cur); uj.retain_grad(); u.append(uj)
    rj = F.relu(bns[j](uj)); rj.retain_grad(); r.append(rj); cur = rj
rec = torch.tanh(last(cur)); loss = F.mse_loss(rec, x); loss.backward()
# ---- GPU replay
Hh = 8*h16
r2d, xd = nhwc(r[2]), x.float().cuda()
wl, bl = last.weight.detach().float().cuda().contiguous(), last.bias.detach().float().cuda()
g0 = torch.full((N*Hh*Hh*32*4,), float("nan"), device="cuda"); g1 = torch.empty_like(g0); g2 = torch.empty_like(g0)
dpre = torch.empty(N*Hh*Hh*32, device="cuda"); lossd, db3 = ws(1), ws(3)
w3 = ws(l.vad_convt_to3_mse_ws_floats(N, Hh, Hh))
vad.hip.check(l.vad_convt_to3_mse(r2d.data_ptr(), wl.data_ptr(), bl.data_ptr(), xd.data_ptr(), None, g0.data_ptr(), dpre.data_ptr(),
                                  lossd.data_ptr(), db3.data_ptr(), w3.data_ptr(), N, Hh, Hh, H.stream()))
print(f"to3: loss {abs(float(lossd[0])-float(loss))/float(loss):.2e}  d r2 {rel(g0[:N*Hh*Hh*32].view(N,Hh,Hh,32).permute(0,3,1,2).cpu(), r[2].grad):.2e}")
zeros = torch.zeros(1024, device="cuda")
for j in (2, 1, 0):
    ci, co, hj = C[j], C[j+1], h16 << j
    ud = nhwc(u[j]); ind = nhwc(hseq if j == 0 else r[j-1])
    gam, bet = bns[j].weight.detach().float().cuda(), bns[j].bias.detach().float().cuda()
    stats, cw = ws(2*co), ws(l.vad_chan_ws_floats(N*4*hj*hj, co))
    vad.hip.check(l.vad_bn_stats(ud.data_ptr(), N*4*hj*hj, co, 1e-5, 0.1, stats.data_ptr(), None, None, cw.data_ptr(), H.stream()))
    dg, dbt, ks = ws(co), ws(co), ws(2*co)
    vad.hip.check(l.vad_bn_act_pool_bwd(ud.data_ptr(), stats.data_ptr(), gam.data_ptr(), bet.data_ptr(), g0.data_ptr(), 0,0,0,0, g1.data_ptr(), g2.data_ptr(), 1,
                                        dg.data_ptr(), dbt.data_ptr(), ks.data_ptr(), cw.data_ptr(), N, 2*hj, 2*hj, co, 2, 0, H.stream()))
    du = g2[:N*hj*hj*4*co].view(N,hj,hj,2,2,co).permute(0,5,1,3,2,4).reshape(N,co,2*hj,2*hj)
    e_du, e_dg = rel(du.cpu(), u[j].grad), rel(dg.cpu(), bns[j].weight.grad)
    dw = torch.full((ci,co,2,2), float("nan"), device="cuda"); ww = ws(l.vad_conv_wgrad_ws_floats(N,hj,1,ci,4*co))
    vad.hip.check(l.vad_conv_wgrad(ind.data_ptr(), g2.data_ptr(), dw.data_ptr(), ww.data_ptr(), N,hj,hj,ci,4*co,1,1,H.stream()))
    wd = convs[j].weight.detach().float().cuda().contiguous(); dgr = ws(l.vad_pack_conv1x1_floats(ci, 4*co))
    vad.hip.check(l.vad_train_pack_convt2x2(wd.data_ptr(), ci, co, None, dgr.data_ptr(), H.stream()))
    vad.hip.check(l.vad_conv1x1(g2.data_ptr(), dgr.data_ptr(), zeros.data_ptr(), g0.data_ptr(), N*hj*hj, 4*co, ci, H.stream()))
    din_ref = (hseq if j == 0 else r[j-1]).grad
    print(f"j={j} ({ci}->{co} at {hj}x{hj}): d u {e_du:.2e}  dgamma {e_dg:.2e}  dW {rel(dw.cpu(), convs[j].weight.grad):.2e}  d in {rel(g0[:N*hj*hj*ci].view(N,hj,hj,ci).permute(0,3,1,2).cpu(), din_ref):.2e}")
