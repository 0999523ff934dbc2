"""GPU parity of the training-step kernels through the C ABI, one op at a time, against
torch autograd on the CPU in fp32 (the arithmetic the reference's train_video.py:44-65 gets from stock autograd).

Tolerances: forward values 2e-5 absolute on O(1) data; gradients 1e-4 of the tensor's largest magnitude (fp32
summation order over up to ~1e5 terms)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fixed_cpu_threads():
    """CPU autograd is the reference here; its summation order (hence which side of zero a borderline activation falls on)
    depends on the thread count.  Pinned so the reference is the same on every host."""
    before = torch.get_num_threads()
    torch.set_num_threads(4)
    yield
    torch.set_num_threads(before)


def _rng(seed):
    return np.random.default_rng(seed)


def _close(got, ref, rtol=1e-4, what=""):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(float(np.abs(ref).max()), 1e-12)
    err = float(np.abs(got - ref).max()) / scale
    assert np.isfinite(got).all() and err < rtol, f"{what}: max err {err:.3e} of scale {scale:.3e}"


def _ws(n):
    return torch.empty(max(int(n), 1), dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("n,h,w,c,act,pool", [(3, 8, 12, 32, 1, 1), (2, 6, 6, 64, 2, 0), (5, 4, 4, 128, 1, 1),
                                              (2, 16, 16, 96, 2, 0), (64, 32, 32, 32, 1, 1), (4, 56, 56, 32, 2, 0),
                                              (4, 14, 14, 128, 2, 0), (4, 6, 10, 64, 1, 1)])
def test_batchnorm_act_pool_forward_backward(vad, n, h, w, c, act, pool):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(n * 100 + c)
    y = (rng.standard_normal((n, c, h, w)) * rng.uniform(0.5, 2, (1, c, 1, 1)) + rng.standard_normal((1, c, 1, 1))).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), (rng.standard_normal(c) * 0.1).astype(np.float32)
    rm, rv = (rng.standard_normal(c) * 0.1).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32)
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    dout = rng.standard_normal((n, c, oh, ow)).astype(np.float32)

    # torch reference
    yt = torch.from_numpy(y).requires_grad_(True)
    gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    rmt, rvt = torch.from_numpy(rm.copy()), torch.from_numpy(rv.copy())
    z = F.batch_norm(yt, rmt, rvt, gt, bt, training=True, momentum=0.1, eps=1e-5)
    z = F.leaky_relu(z, 0.2) if act == 1 else F.relu(z)
    z = F.max_pool2d(z, 2, 2) if pool else z
    (z * torch.from_numpy(dout)).sum().backward()

    yd, gd, bd = H.nhwc(y), H.dev(gamma), H.dev(beta)
    rmd, rvd = H.dev(rm), H.dev(rv)
    stats, ws = _ws(2 * c), _ws(l.vad_chan_ws_floats(n * h * w, c))
    vad.hip.check(l.vad_bn_stats(yd.data_ptr(), n * h * w, c, 1e-5, 0.1, stats.data_ptr(), rmd.data_ptr(), rvd.data_ptr(),
                                 ws.data_ptr(), H.stream()))
    out = torch.full((n, oh, ow, c), float("nan"), device="cuda")
    vad.hip.check(l.vad_bn_act_pool_fwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, 0, 0, 0,
                                        n, h, w, c, act, pool, H.stream()))
    _close(H.to_nchw(out), z.detach().numpy(), 2e-5, "bn forward")
    _close(rmd.cpu().numpy(), rmt.numpy(), 1e-5, "running_mean")
    _close(rvd.cpu().numpy(), rvt.numpy(), 1e-5, "running_var")

    dy = torch.full((n, h, w, c), float("nan"), device="cuda")
    dg, db, ks = _ws(c), _ws(c), _ws(2 * c)
    doutd = H.nhwc(dout)
    vad.hip.check(l.vad_bn_act_pool_bwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), doutd.data_ptr(), 0, 0, 0, 0,
                                        dy.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), ks.data_ptr(), ws.data_ptr(),
                                        n, h, w, c, act, pool, H.stream()))
    _close(H.to_nchw(dy), yt.grad.numpy(), 1e-4, "dy")
    _close(dg.cpu().numpy(), gt.grad.numpy(), 1e-4, "dgamma")
    _close(db.cpu().numpy(), bt.grad.numpy(), 1e-4, "dbeta")
    if not pool and h % 2 == 0 and w % 2 == 0:      # space-to-depth form of the same gradient
        dy2 = torch.full((n, h // 2, w // 2, 4, c), float("nan"), device="cuda")
        vad.hip.check(l.vad_bn_act_pool_bwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), doutd.data_ptr(), 0, 0, 0, 0,
                                            dy2.data_ptr(), 1, dg.data_ptr(), db.data_ptr(), ks.data_ptr(), ws.data_ptr(),
                                            n, h, w, c, act, pool, H.stream()))
        ref = dy.view(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, c)
        assert torch.equal(dy2, ref)


@pytest.mark.parametrize("mean,std", [(100.0, 0.1), (-30.0, 1.0), (0.0, 1.0), (1000.0, 3.0)])
def test_batchnorm_statistics_survive_large_mean(vad, mean, std):
    """Variance of a channel whose mean dwarfs its spread.  E[x^2] - mean^2 on the raw values cancels catastrophically in
    fp32 (mean/std = 1000 leaves no correct digit); the kernel shifts by a per-channel sample before squaring.  This is the
    defect that made the training step's BatchNorm outputs 10x less accurate than fp32 autograd's and flipped ReLU
    decisions downstream (see tests/test_hip_train_step.py)."""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(int(abs(mean)) + 5)
    n, h, w, c = 4, 14, 14, 64
    y = (mean + std * rng.standard_normal((n, h, w, c))).astype(np.float32)
    yd = H.dev(y)
    stats, ws = _ws(2 * c), _ws(l.vad_chan_ws_floats(n * h * w, c))
    vad.hip.check(l.vad_bn_stats(yd.data_ptr(), n * h * w, c, 1e-5, 0.1, stats.data_ptr(), None, None, ws.data_ptr(), H.stream()))
    y64 = y.astype(np.float64).reshape(-1, c)
    got = stats.cpu().numpy().astype(np.float64)
    assert np.abs(got[:c] - y64.mean(0)).max() < 1e-6 * max(1.0, abs(mean), std)     # fp32 partial sums of ~100 shifted terms
    want = 1.0 / np.sqrt(y64.var(0) + 1e-5)
    assert np.abs(got[c:] / want - 1).max() < 2e-6, f"1/std off by {np.abs(got[c:] / want - 1).max():.2e}"


def test_batchnorm_forward_time_major_strided_destination(vad):
    """The encoder's last stage writes straight into the ConvLSTM operand buffers: frame b*T+t -> slot t*B+b, pixel
    stride cin_x+hid; the backward reads its upstream gradient through the same view."""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(7)
    bsz, t, h, w, c, ps = 3, 4, 4, 4, 32, 80
    n = bsz * t
    y = rng.standard_normal((n, h, w, c)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.standard_normal(c).astype(np.float32) * 0.1
    yd, gd, bd = H.dev(y), H.dev(gamma), H.dev(beta)
    stats, ws = _ws(2 * c), _ws(l.vad_chan_ws_floats(n * h * w, c))
    vad.hip.check(l.vad_bn_stats(yd.data_ptr(), n * h * w, c, 1e-5, 0.1, stats.data_ptr(), None, None, ws.data_ptr(), H.stream()))
    dense = torch.empty(n, h // 2, w // 2, c, device="cuda")
    vad.hip.check(l.vad_bn_act_pool_fwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dense.data_ptr(), 0, 0, 0, 0,
                                        n, h, w, c, 1, 1, H.stream()))
    cat = torch.zeros(t, bsz, h // 2, w // 2, ps, device="cuda")
    vad.hip.check(l.vad_bn_act_pool_fwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), cat.data_ptr(), 0, ps, t, bsz,
                                        n, h, w, c, 1, 1, H.stream()))
    ref = dense.view(bsz, t, h // 2, w // 2, c).permute(1, 0, 2, 3, 4)
    assert torch.equal(cat[..., :c], ref) and float(cat[..., c:].abs().max()) == 0.0
    # backward through the same view == backward from the dense gradient
    dcat = torch.from_numpy(rng.standard_normal((t, bsz, h // 2, w // 2, ps)).astype(np.float32)).cuda()
    ddense = dcat[..., :c].permute(1, 0, 2, 3, 4).reshape(n, h // 2, w // 2, c).contiguous()
    outs = []
    for src, ps_, tt, bb in ((dcat, ps, t, bsz), (ddense, 0, 0, 0)):
        dy = torch.full((n, h, w, c), float("nan"), device="cuda")
        dg, db, ks = _ws(c), _ws(c), _ws(2 * c)
        vad.hip.check(l.vad_bn_act_pool_bwd(yd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), src.data_ptr(), 0, ps_, tt, bb,
                                            dy.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), ks.data_ptr(), ws.data_ptr(),
                                            n, h, w, c, 1, 1, H.stream()))
        outs.append((dy, dg, db))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("nb,hw,hid,first", [(3, 16, 32, False), (2, 64, 128, True), (1, 4, 64, False)])
def test_lstm_gates_forward_backward(vad, nb, hw, hid, first):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(nb + hid)
    z = rng.standard_normal((nb * hw, 4 * hid)).astype(np.float32) * 1.5
    cp = None if first else rng.standard_normal((nb * hw, hid)).astype(np.float32)
    dh1 = rng.standard_normal((nb, hw, hid)).astype(np.float32)
    dh2 = rng.standard_normal((nb, hw, hid + 16)).astype(np.float32)       # strided source (h-part of a wider buffer)
    dcn = rng.standard_normal((nb * hw, hid)).astype(np.float32)

    zt = torch.from_numpy(z).requires_grad_(True)
    cpt = torch.zeros(nb * hw, hid) if first else torch.from_numpy(cp).requires_grad_(True)
    i, f, g, o = torch.split(zt, hid, dim=1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    cn = f * cpt + i * g
    hn = o * torch.tanh(cn)
    dh = torch.from_numpy(dh1).reshape(nb * hw, hid) + torch.from_numpy(dh2[..., 16:]).reshape(nb * hw, hid)
    ((hn * dh).sum() + (cn * torch.from_numpy(dcn)).sum()).backward()

    zd = H.dev(z)
    cpd = None if first else H.dev(cp)
    c_out = torch.empty(nb * hw, hid, device="cuda")
    h1 = torch.empty(nb, hw, hid, device="cuda")
    h2 = torch.zeros(nb, hw, hid + 32, device="cuda")
    vad.hip.check(l.vad_lstm_gates_fwd(zd.data_ptr(), vad.hip.ptr(cpd), c_out.data_ptr(), h1.data_ptr(), 0, 0,
                                       h2.data_ptr() + 4 * 32, 0, hid + 32, nb, hw, hid, H.stream()))
    _close(c_out.cpu().numpy(), cn.detach().numpy(), 2e-6, "c")
    _close(h1.cpu().numpy().reshape(nb * hw, hid), hn.detach().numpy(), 2e-6, "h")
    assert torch.equal(h2[..., 32:], h1) and float(h2[..., :32].abs().max()) == 0.0
    _close(zd.cpu().numpy(), torch.cat([i, f, g, o], 1).detach().numpy(), 2e-6, "gates")

    dz, dcp = torch.empty(nb * hw, 4 * hid, device="cuda"), torch.empty(nb * hw, hid, device="cuda")
    dh1d, dh2d, dcnd = H.dev(dh1), H.dev(dh2), H.dev(dcn)      # keep the device buffers alive across the launch
    vad.hip.check(l.vad_lstm_gates_bwd(zd.data_ptr(), vad.hip.ptr(cpd), c_out.data_ptr(), dh1d.data_ptr(), 0, 0,
                                       dh2d.data_ptr() + 4 * 16, 0, hid + 16, dcnd.data_ptr(), dz.data_ptr(), dcp.data_ptr(),
                                       nb, hw, hid, H.stream()))
    _close(dz.cpu().numpy(), zt.grad.numpy(), 2e-5, "dz")
    if not first:
        _close(dcp.cpu().numpy(), cpt.grad.numpy(), 2e-5, "dc_prev")


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 8, 8, 32, 32), (3, 6, 10, 64, 32), (2, 16, 16, 256, 512), (40, 32, 32, 32, 64),
                                            (1, 2, 2, 128, 128), (2, 3, 3, 64, 128), (4, 7, 7, 128, 64), (2, 5, 3, 64, 128)])
@pytest.mark.parametrize("precision", [0, 1])
def test_conv3x3_weight_and_data_gradients(vad, n, h, w, cin, cout, precision):
    """precision 1: the device packers emit the split-fp16 operand form and the forward / data-gradient convolutions run
    on it; the weight gradient is checked in all three operand forms (exact, bf16, split-fp16) either way."""
    _check_conv3x3_gradients(vad, n, h, w, cin, cout, precision)


def _check_conv3x3_gradients(vad, n, h, w, cin, cout, precision):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(cin + cout + h)
    a = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    g = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    at, wtt = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    (F.conv2d(at, wtt, padding=1) * torch.from_numpy(g)).sum().backward()

    ad, gd = H.nhwc(a), H.nhwc(g)
    dw = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
    ws = _ws(l.vad_conv_wgrad_ws_floats(n, h, 9, cin, cout))
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), gd.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cin, cout, 9, 0, 0, H.stream()))
    _close(dw.cpu().numpy(), wtt.grad.numpy(), 1e-4, "dW")
    # bf16 operands (VAD_PREC_BF16): 16 pixels per MFMA, fp32 accumulation; ~2^-9 per product, averaging over the pixel sum
    dwb = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), gd.data_ptr(), dwb.data_ptr(), ws.data_ptr(), n, h, w, cin, cout, 9, 0, 2, H.stream()))
    _close(dwb.cpu().numpy(), wtt.grad.numpy(), 1e-2, "dW (bf16 operands)")
    # split-fp16 operands (VAD_PREC_SPLIT, round 4): 22-bit products - the exact kernel's result to a few 1e-7 of the largest entry
    dws = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), gd.data_ptr(), dws.data_ptr(), ws.data_ptr(), n, h, w, cin, cout, 9, 0, 1, H.stream()))
    _close(dws.cpu().numpy(), dw.cpu().numpy(), 2e-6, "dW (split-fp16 operands) against the exact kernel")

    # data gradient = forward kernel on the re-packed weight (device-side packing of the live parameter)
    wd = H.dev(wt)
    fwd = _ws(l.vad_pack_conv3x3_floats(cout, cin))
    dgr = _ws(l.vad_pack_conv3x3_floats(cin, cout))
    vad.hip.check(l.vad_train_pack_conv3x3(wd.data_ptr(), cout, cin, fwd.data_ptr(), dgr.data_ptr(), precision, H.stream()))
    zero_in, zero_out = torch.zeros(cin, device="cuda"), torch.zeros(cout, device="cuda")
    da = torch.full((n, h, w, cin), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv3x3(gd.data_ptr(), 0, dgr.data_ptr(), zero_in.data_ptr(), da.data_ptr(), 0, n, h, w, cout, cin, 0, 0, precision, H.stream()))
    _close(H.to_nchw(da), at.grad.numpy(), 1e-4, "dA")
    out = torch.full((n, h, w, cout), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv3x3(ad.data_ptr(), 0, fwd.data_ptr(), zero_out.data_ptr(), out.data_ptr(), 0, n, h, w, cin, cout, 0, 0, precision, H.stream()))
    _close(H.to_nchw(out), F.conv2d(torch.from_numpy(a), torch.from_numpy(wt), padding=1).numpy(), 2e-5, "forward with device pack")


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 4, 4, 128, 128), (3, 8, 6, 128, 64), (2, 16, 16, 64, 32), (8, 32, 32, 64, 32),
                                            (4, 28, 28, 64, 32), (4, 7, 7, 32, 128), (2, 5, 3, 128, 64)])
@pytest.mark.parametrize("precision", [0, 1])
def test_convt2x2_weight_and_data_gradients(vad, n, h, w, cin, cout, precision):
    _check_convt2x2_gradients(vad, n, h, w, cin, cout, precision)


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 32, 48), (5, 12, 80), (2, 64, 256), (1, 8, 272), (1, 4, 768)])
def test_routed_first_layer_weight_gradient_fp32(vad, n, h, w):
    """fp32 form of the routed first-layer weight gradient (csrc/train_ops.hip conv_c3_wgrad_routed_f32_kernel; the bf16 form:
    tests/test_hip_train_bf16.py): dW of Conv2d(3->32) + BatchNorm(batch statistics) + LeakyReLU(0.2) + MaxPool2 from the POOLED
    gradient, one routing byte per pooled element and the Gram matrix of the input patches, against a float64 evaluation that
    takes every decision from the stored fp32 conv output as the kernels do - and against the form it replaces (pass B + the plain
    weight gradient), which must be no closer to float64 than a few 1e-6."""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(11 * n + h + w)
    x = H.dev(rng.uniform(-1, 1, (n, 3, h, w)))
    w0 = (rng.standard_normal((32, 3, 3, 3)) * 0.3).astype(np.float32)
    b0 = (rng.standard_normal(32) * 0.1).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, 32).astype(np.float32), (rng.standard_normal(32) * 0.1).astype(np.float32)
    dout = H.dev(rng.standard_normal((n, h // 2, w // 2, 32)) * 1e-3)
    wp, bo = H.pack_conv3x3(w0, b0)
    y = torch.full((n, h, w, 32), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv3x3_c3(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), y.data_ptr(), n, h, w, 32, 0, 0, H.stream()))
    stats, cws = _ws(64), _ws(l.vad_chan_ws_floats(n * h * w, 32))
    vad.hip.check(l.vad_bn_stats(y.data_ptr(), n * h * w, 32, 1e-5, 0.1, stats.data_ptr(), None, None, cws.data_ptr(), H.stream()))
    mean, invstd = stats[:32].double().cpu(), stats[32:].double().cpu()
    g64, be64 = torch.from_numpy(gamma).double(), torch.from_numpy(beta).double()
    yq = y.double().cpu().permute(0, 3, 1, 2)
    xhat = (yq - mean[None, :, None, None]) * invstd[None, :, None, None]
    zq = F.leaky_relu(xhat * g64[None, :, None, None] + be64[None, :, None, None], 0.2)
    zz = zq.view(n, 32, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 32, h // 2, w // 2, 4)
    am = zz.argmax(-1)
    val = zz.gather(-1, am[..., None])[..., 0]
    gz = dout.double().cpu().permute(0, 3, 1, 2) * torch.where(val > 0, 1.0, 0.2)
    dz = torch.zeros(n, 32, h // 2, w // 2, 4, dtype=torch.float64).scatter_(-1, am[..., None], gz[..., None])
    dz = dz.view(n, 32, h // 2, w // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 32, h, w)
    m_ = n * h * w
    k1, k2 = dz.sum((0, 2, 3)) / m_, (dz * xhat).sum((0, 2, 3)) / m_
    dy = (g64 * invstd)[None, :, None, None] * (dz - k1[None, :, None, None] - xhat * k2[None, :, None, None])
    X = F.unfold(x.double().cpu(), 3, padding=1).permute(0, 2, 1).reshape(-1, 27)
    ref = (dy.permute(0, 2, 3, 1).reshape(-1, 32).t() @ X).reshape(32, 3, 3, 3)
    gd, bd, w0d, b0d = H.dev(gamma), H.dev(beta), H.dev(w0), H.dev(b0)
    dg, db, ks = _ws(32), _ws(32), _ws(64)
    dyd = torch.full((n, h, w, 32), float("nan"), device="cuda")
    vad.hip.check(l.vad_bn_act_pool_bwd(y.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dout.data_ptr(), 0, 0, 0, 0,
                                        dyd.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), ks.data_ptr(), cws.data_ptr(), n, h, w, 32, 1, 1, H.stream()))
    wws = _ws(max(l.vad_conv_c3_wgrad_ws_floats(n, h, 32), l.vad_conv_c3_wgrad_routed_ws_floats(n, h)))
    dw_old = torch.full((32, 3, 3, 3), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_c3_wgrad(x.data_ptr(), dyd.data_ptr(), dw_old.data_ptr(), wws.data_ptr(), n, h, w, 32, H.stream()))
    codes = torch.full((n * (h // 2) * (w // 2), 32), 255, dtype=torch.uint8, device="cuda")
    dg2, db2, ks2 = _ws(32), _ws(32), _ws(64)
    vad.hip.check(l.vad_bn_act_pool_bwd_codes_t(y.data_ptr(), 0, stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dout.data_ptr(), 0, 0, 0, 0,
                                                None, 0, dg2.data_ptr(), db2.data_ptr(), ks2.data_ptr(), cws.data_ptr(), n, h, w, 32, 1, 1,
                                                codes.data_ptr(), H.stream()))
    assert torch.equal(ks2, ks) and torch.equal(dg2, dg) and int(codes.max()) <= 7
    dw_new = torch.full((32, 3, 3, 3), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_c3_wgrad_routed(x.data_ptr(), dout.data_ptr(), 0, codes.data_ptr(), w0d.data_ptr(), b0d.data_ptr(), stats.data_ptr(),
                                             gd.data_ptr(), ks.data_ptr(), dw_new.data_ptr(), wws.data_ptr(), n, h, w, 32, H.stream()))
    scale = float(ref.abs().max())
    e_old = float((dw_old.double().cpu() - ref).abs().max()) / scale
    e_new = float((dw_new.double().cpu() - ref).abs().max()) / scale
    print(f"first-layer dW (fp32) vs float64 ({n}x{h}x{w}): routed {e_new:.2e}, pass B + plain {e_old:.2e} of max |dW|")
    assert bool(torch.isfinite(dw_new).all()) and e_new < 2e-6 and e_old < 2e-6, (e_new, e_old)


@pytest.mark.parametrize("n,h,w,cin,ncols,taps", [(2, 24, 40, 64, 64, 9), (3, 12, 33, 32, 64, 9), (2, 16, 16, 128, 256, 9), (5, 7, 20, 64, 128, 9),
                                                  (2, 9, 64, 32, 128, 9), (7, 5, 70, 64, 64, 9), (2, 24, 40, 64, 256, 1), (3, 12, 33, 32, 128, 1), (4, 16, 16, 128, 128, 1)])
def test_split_weight_gradient_kernel_forms(vad, n, h, w, cin, ncols, taps):
    """VAD_PREC_SPLIT weight gradients: the per-lane kernel (vad_debug_set_wgrad_split(1)) and its LDS-staged, transposed form
    (2, where ncols % 64 == 0 and cin % 64 == 0 or cin == 32: 16- and 32-pixel groups, ragged widths, pixel halves for
    32-channel layers) and the row-ring kernel (3, default for those 3x3 layers: every operand row staged once) against the exact-fp32 kernel (0) on the same operands: 22-bit products, 2e-6 of the largest entry."""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(cin + ncols + w)
    a = H.dev(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    g = H.dev((rng.standard_normal((n, h, w, ncols)) * 0.05).astype(np.float32))
    layout = 0 if taps == 9 else 4
    shape = (ncols, cin, 3, 3) if taps == 9 else (ncols, cin, 1, 1)
    ws = _ws(l.vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols))
    out = {}
    try:
        for mode in (0, 1, 2, 3):
            l.vad_debug_set_wgrad_split(mode)
            dw = torch.full(shape, float("nan"), device="cuda")
            vad.hip.check(l.vad_conv_wgrad(a.data_ptr(), g.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cin, ncols, taps, layout, 1, H.stream()))
            out[mode] = dw.cpu().numpy()
    finally:
        l.vad_debug_set_wgrad_split(3)
    ref = torch.einsum("nhwc,nhwk->kc", a.double()[:, :, :, :], g.double()) if taps == 1 else None
    if taps == 1:
        _close(out[0].reshape(ncols, cin), ref.cpu().numpy(), 1e-5, "exact kernel against float64")
    _close(out[1], out[0], 2e-6, "per-lane split kernel")
    _close(out[2], out[0], 2e-6, "LDS-staged split kernel")
    _close(out[3], out[0], 2e-6, "row-ring split kernel (3x3 layers; else the LDS-staged one again)")
    # exact fp32 (precision 0): the row-ring kernel (default for the 3x3 layers it takes) against the per-wave kernel - the same
    # products, fp32 sums in another order
    dwr = torch.full(shape, float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_wgrad(a.data_ptr(), g.data_ptr(), dwr.data_ptr(), ws.data_ptr(), n, h, w, cin, ncols, taps, layout, 0, H.stream()))
    l.vad_debug_set_wgrad_ring_f32(0)
    try:
        dwp = torch.full(shape, float("nan"), device="cuda")
        vad.hip.check(l.vad_conv_wgrad(a.data_ptr(), g.data_ptr(), dwp.data_ptr(), ws.data_ptr(), n, h, w, cin, ncols, taps, layout, 0, H.stream()))
    finally:
        l.vad_debug_set_wgrad_ring_f32(1)
    assert torch.equal(dwp.cpu(), torch.from_numpy(out[0]))
    _close(dwr.cpu().numpy(), out[0], 2e-6, "exact row-ring kernel")


def _check_convt2x2_gradients(vad, n, h, w, cin, cout, precision):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(cin * 3 + cout + h)
    a = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    g = rng.standard_normal((n, cout, 2 * h, 2 * w)).astype(np.float32)
    wt = (rng.standard_normal((cin, cout, 2, 2)) / np.sqrt(cin)).astype(np.float32)
    at, wtt = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    (F.conv_transpose2d(at, wtt, stride=2) * torch.from_numpy(g)).sum().backward()

    ad = H.nhwc(a)
    g_s2d = torch.from_numpy(g).permute(0, 2, 3, 1).reshape(n, h, 2, w, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(n, h, w, 4 * cout).contiguous().cuda()
    dw = torch.full((cin, cout, 2, 2), float("nan"), device="cuda")
    ws = _ws(l.vad_conv_wgrad_ws_floats(n, h, 1, cin, 4 * cout))
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), g_s2d.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cin, 4 * cout, 1, 1, 0, H.stream()))
    _close(dw.cpu().numpy(), wtt.grad.numpy(), 1e-4, "dW")
    dwb = torch.full((cin, cout, 2, 2), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), g_s2d.data_ptr(), dwb.data_ptr(), ws.data_ptr(), n, h, w, cin, 4 * cout, 1, 1, 2, H.stream()))
    _close(dwb.cpu().numpy(), wtt.grad.numpy(), 1e-2, "dW (bf16 operands)")
    dws = torch.full((cin, cout, 2, 2), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), g_s2d.data_ptr(), dws.data_ptr(), ws.data_ptr(), n, h, w, cin, 4 * cout, 1, 1, 1, H.stream()))
    _close(dws.cpu().numpy(), dw.cpu().numpy(), 2e-6, "dW (split-fp16 operands) against the exact kernel")

    wd = H.dev(wt)
    fwd = _ws(l.vad_pack_convt2x2_floats(cin, cout))
    dgr = _ws(l.vad_pack_conv1x1_floats(cin, 4 * cout))
    vad.hip.check(l.vad_train_pack_convt2x2(wd.data_ptr(), cin, cout, fwd.data_ptr(), dgr.data_ptr(), precision, H.stream()))
    zero_in, zero_out = torch.zeros(cin, device="cuda"), torch.zeros(cout, device="cuda")
    da = torch.full((n, h, w, cin), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv1x1(g_s2d.data_ptr(), dgr.data_ptr(), zero_in.data_ptr(), da.data_ptr(), n * h * w, 4 * cout, cin, H.stream()))
    _close(H.to_nchw(da), at.grad.numpy(), 1e-4, "dA")
    out = torch.full((n, 2 * h, 2 * w, cout), float("nan"), device="cuda")
    vad.hip.check(l.vad_convt2x2(ad.data_ptr(), 0, fwd.data_ptr(), zero_out.data_ptr(), out.data_ptr(), 0, n, h, w, cin, cout, 0, precision, H.stream()))
    _close(H.to_nchw(out), F.conv_transpose2d(torch.from_numpy(a), torch.from_numpy(wt), stride=2).numpy(), 2e-5, "forward with device pack")


@pytest.mark.parametrize("n,h,w,cout", [(2, 8, 8, 32), (3, 12, 20, 32), (1, 6, 6, 64), (10, 64, 64, 32)])
def test_first_layer_forward_pack_and_weight_gradient(vad, n, h, w, cout):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(n + h + cout)
    x = rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    g = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, 3, 3, 3)) / np.sqrt(27)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    wtt = torch.from_numpy(wt).requires_grad_(True)
    ref = F.conv2d(torch.from_numpy(x), wtt, torch.from_numpy(b), padding=1)
    (ref * torch.from_numpy(g)).sum().backward()

    xd, gd = H.dev(x), H.nhwc(g)
    dw = torch.full((cout, 3, 3, 3), float("nan"), device="cuda")
    ws = _ws(l.vad_conv_c3_wgrad_ws_floats(n, h, cout))
    vad.hip.check(l.vad_conv_c3_wgrad(xd.data_ptr(), gd.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, w, cout, H.stream()))
    _close(dw.cpu().numpy(), wtt.grad.numpy(), 1e-4, "dW")
    fwd = torch.zeros(l.vad_pack_conv3x3_c3_floats(cout), device="cuda")
    wd, bd = H.dev(wt), H.dev(b)
    vad.hip.check(l.vad_train_pack_conv3x3_c3(wd.data_ptr(), cout, fwd.data_ptr(), H.stream()))
    out = torch.full((n, h, w, cout), float("nan"), device="cuda")
    vad.hip.check(l.vad_conv3x3_c3(xd.data_ptr(), fwd.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, w, cout, 0, 0, H.stream()))
    _close(H.to_nchw(out), ref.detach().numpy(), 2e-5, "forward with device pack")


@pytest.mark.parametrize("n,h,w", [(2, 4, 4), (3, 8, 12), (6, 64, 64)])
def test_last_layer_convt_tanh_mse_forward_backward(vad, n, h, w):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(n + h)
    a = np.maximum(rng.standard_normal((n, 32, h, w)), 0).astype(np.float32)
    x = rng.uniform(-1, 1, (n, 3, 2 * h, 2 * w)).astype(np.float32)
    wt = (rng.standard_normal((32, 3, 2, 2)) / np.sqrt(32)).astype(np.float32)
    b = (rng.standard_normal(3) * 0.1).astype(np.float32)
    at, wtt, bt = (torch.from_numpy(v).requires_grad_(True) for v in (a, wt, b))
    rec = torch.tanh(F.conv_transpose2d(at, wtt, bt, stride=2))
    loss = F.mse_loss(rec, torch.from_numpy(x))
    loss.backward()

    ad, wd, bd, xd = H.nhwc(a), H.dev(wt), H.dev(b), H.dev(x)
    recon = torch.full((n, 3, 2 * h, 2 * w), float("nan"), device="cuda")
    din, dpre = torch.full((n, h, w, 32), float("nan"), device="cuda"), torch.full((n * h * w, 32), float("nan"), device="cuda")
    lossd, db = _ws(1), _ws(3)
    ws = _ws(l.vad_convt_to3_mse_ws_floats(n, h, w))
    vad.hip.check(l.vad_convt_to3_mse(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), xd.data_ptr(), recon.data_ptr(), din.data_ptr(),
                                      dpre.data_ptr(), lossd.data_ptr(), db.data_ptr(), ws.data_ptr(), n, h, w, H.stream()))
    _close(recon.cpu().numpy(), rec.detach().numpy(), 2e-6, "recon")
    assert abs(float(lossd[0]) - float(loss.detach())) < 1e-6 * float(loss.detach()) + 1e-9
    _close(H.to_nchw(din), at.grad.numpy(), 1e-4, "d input")
    _close(db.cpu().numpy(), bt.grad.numpy(), 1e-4, "d bias")
    dw = torch.full((32, 3, 2, 2), float("nan"), device="cuda")
    ws2 = _ws(l.vad_conv_wgrad_ws_floats(n, h, 1, 32, 32))
    vad.hip.check(l.vad_conv_wgrad(ad.data_ptr(), dpre.data_ptr(), dw.data_ptr(), ws2.data_ptr(), n, h, w, 32, 32, 1, 3, 0, H.stream()))
    _close(dw.cpu().numpy(), wtt.grad.numpy(), 1e-4, "dW")


def test_adam_matches_torch_optim(vad):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(3)
    n = 10007
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-4, weight_decay=1e-5)
    pd, m, v = H.dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 5):
        g = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 0, n)).astype(np.float32)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        gd = H.dev(g)
        vad.hip.check(l.vad_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 1e-5,
                                      step, 1.0, H.stream()))
        d = np.abs(pd.cpu().numpy() - pt.detach().numpy()).max()
        assert d < 2e-7, f"step {step}: parameter difference {d:.3e} (updates are ~1e-4)"
    # the update itself (not just p) agrees: compare the accumulated displacement
    _close(pd.cpu().numpy() - p0, pt.detach().numpy() - p0, 2e-3, "displacement")


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 32, 48), (5, 64, 64), (1, 48, 16)])
@pytest.mark.parametrize("given_drecon", [False, True])
def test_image_last_layer_conv_tanh_backward(vad, n, h, w, given_drecon):
    """ConvAutoencoder's last layer in train mode (models/autoencoder.py:134-135): forward through the scoring tail kernel on
    device-packed weights; backward with the gradient either of nn.MSELoss (computed inside) or handed in (the SSIM /
    combined criterion's)."""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(n * 7 + h + w)
    a = np.maximum(rng.standard_normal((n, 32, h, w)), 0).astype(np.float32)
    x = rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    wt = (rng.standard_normal((3, 32, 3, 3)) / np.sqrt(288)).astype(np.float32)
    b = (rng.standard_normal(3) * 0.1).astype(np.float32)
    R = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    at, wtt, bt = (torch.from_numpy(v).requires_grad_(True) for v in (a, wt, b))
    rec = torch.tanh(F.conv2d(at, wtt, bt, padding=1))
    loss = (rec * torch.from_numpy(R)).sum() if given_drecon else F.mse_loss(rec, torch.from_numpy(x))
    loss.backward()

    ad, xd, wd, bd = H.nhwc(a), H.dev(x), H.dev(wt), H.dev(b)
    fwd = _ws(l.vad_pack_conv3x3_to3_floats(32))
    dgr = torch.zeros(l.vad_pack_conv3x3_c3_floats(32), device="cuda")
    vad.hip.check(l.vad_train_pack_conv3x3_to3(wd.data_ptr(), 32, fwd.data_ptr(), dgr.data_ptr(), H.stream()))
    recon = torch.full((n, 3, h, w), float("nan"), device="cuda")
    parts = _ws(n * l.vad_score_partials(0, h, w))
    vad.hip.check(l.vad_conv3x3_to3_score(ad.data_ptr(), fwd.data_ptr(), bd.data_ptr(), xd.data_ptr(), parts.data_ptr(), recon.data_ptr(), None,
                                          n, h, w, 32, H.stream()))
    _close(recon.cpu().numpy(), rec.detach().numpy(), 2e-6, "recon")
    dpre = torch.full((n, 3, h, w), float("nan"), device="cuda")
    din = torch.full((n, h, w, 32), float("nan"), device="cuda")
    dw, db = torch.full((3, 32, 3, 3), float("nan"), device="cuda"), _ws(3)
    ws = _ws(l.vad_conv3x3_to3_bwd_ws_floats(n, h, w, 32))
    Rd = H.dev(R)
    vad.hip.check(l.vad_conv3x3_to3_tanh_bwd(ad.data_ptr(), recon.data_ptr(), None if given_drecon else xd.data_ptr(),
                                             Rd.data_ptr() if given_drecon else None, dgr.data_ptr(), dpre.data_ptr(), din.data_ptr(),
                                             dw.data_ptr(), db.data_ptr(), ws.data_ptr(), n, h, w, 32, 1.0, H.stream()))
    # grad_mul (the split-fp16 step's gradient scale) is an exact power-of-two rescaling of every output
    din2, dw2, db2, dpre2 = torch.empty_like(din), torch.empty_like(dw), torch.empty_like(db), torch.empty_like(dpre)
    vad.hip.check(l.vad_conv3x3_to3_tanh_bwd(ad.data_ptr(), recon.data_ptr(), None if given_drecon else xd.data_ptr(),
                                             Rd.data_ptr() if given_drecon else None, dgr.data_ptr(), dpre2.data_ptr(), din2.data_ptr(),
                                             dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), n, h, w, 32, 4096.0, H.stream()))
    assert torch.equal(din2, din * 4096.0) and torch.equal(dw2, dw * 4096.0) and torch.equal(db2, db * 4096.0)
    assert l.vad_conv3x3_to3_tanh_bwd(ad.data_ptr(), recon.data_ptr(), None if given_drecon else xd.data_ptr(),
                                      Rd.data_ptr() if given_drecon else None, dgr.data_ptr(), dpre2.data_ptr(), din2.data_ptr(),
                                      dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), n, h, w, 32, 3.0, H.stream()) != 0     # not a power of two
    _close(H.to_nchw(din), at.grad.numpy(), 1e-4, "d input")
    _close(dw.cpu().numpy(), wtt.grad.numpy(), 1e-4, "dW")
    _close(db.cpu().numpy(), bt.grad.numpy(), 1e-4, "d bias")
