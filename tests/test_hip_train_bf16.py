"""GPU tests of the bf16-TENSOR training mode (VAD_PREC_BF16S, `VideoTrainer(precision="bf16")`, BASELINE.json configs[4]'s
dtype): activations and activation gradients are bf16 in HBM, every kernel computes in fp32 on the widened values and rounds
(nearest even) where it stores.

No reference oracle exists for bf16 (the reference trains in fp32), so each kernel is pinned EXACTLY to a kernel that is
itself pinned to the reference arithmetic elsewhere (tests/test_hip_train_ops.py, tests/test_hip_layers.py):
  * pointwise / reduction kernels: bf16 form on bf16 tensors == round_bf16(fp32 form on the same values), bit for bit;
  * MFMA kernels: the bf16-tensor form == the bf16-OPERAND form (VAD_PREC_BF16, fp32 tensors converted while staged) on
    bf16-representable inputs - the matrix instructions then see identical operands in identical order - rounded to bf16.
The end-to-end gates (loss curve vs the fp32 CPU restatement, first-step gradient cosine) are in test_hip_train_step.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16S, BF16 = 3, 2


def _rng(seed):
    return np.random.default_rng(seed)


def _ws(n):
    return torch.empty(max(int(n), 1), dtype=torch.float32, device="cuda")


def _rep(a):
    """fp32 array -> (bf16 device tensor, the same values as an fp32 device tensor): bf16-representable test data."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16)
    return t, t.float()


def _nan16(*shape):
    return torch.full(shape, float("nan"), dtype=torch.bfloat16, device="cuda")


def _nan32(*shape):
    return torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("wide", [1, 0])       # eight / four channels per thread in the bf16 forward and backward-apply passes
@pytest.mark.parametrize("n,h,w,c,act,pool", [(3, 8, 12, 32, 1, 1), (2, 6, 6, 64, 2, 0), (5, 4, 4, 128, 1, 1), (4, 14, 14, 128, 2, 0),
                                              (4, 6, 10, 64, 1, 1), (16, 32, 32, 32, 1, 1)])
def test_batchnorm_passes_on_bf16_tensors(vad, n, h, w, c, act, pool, wide):
    l = vad.hip.lib()
    l.vad_debug_set_bn_wide(wide)
    try:
        _batchnorm_passes_on_bf16_tensors(vad, n, h, w, c, act, pool)
    finally:
        l.vad_debug_set_bn_wide(1)


def _batchnorm_passes_on_bf16_tensors(vad, n, h, w, c, act, pool):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(n * 100 + c + act)
    y16, y32 = _rep(rng.standard_normal((n, h, w, c)) * rng.uniform(0.5, 2, (1, 1, 1, c)) + rng.standard_normal((1, 1, 1, c)))
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    d16, d32 = _rep(rng.standard_normal((n, oh, ow, c)))
    gamma, beta = H.dev(rng.uniform(0.5, 1.5, c)), H.dev(rng.standard_normal(c) * 0.1)
    stats, ws = _ws(2 * c), _ws(l.vad_chan_ws_floats(n * h * w, c))
    vad.hip.check(l.vad_bn_stats(y32.data_ptr(), n * h * w, c, 1e-5, 0.1, stats.data_ptr(), None, None, ws.data_ptr(), H.stream()))
    o32, o16 = _nan32(n, oh, ow, c), _nan16(n, oh, ow, c)
    args = (stats.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    vad.hip.check(l.vad_bn_act_pool_fwd_t(y32.data_ptr(), 0, *args, o32.data_ptr(), 0, 0, 0, 0, n, h, w, c, act, pool, H.stream()))
    vad.hip.check(l.vad_bn_act_pool_fwd_t(y16.data_ptr(), 1, *args, o16.data_ptr(), 0, 0, 0, 0, n, h, w, c, act, pool, H.stream()))
    assert torch.equal(o16, o32.to(torch.bfloat16))
    for s2d in ((0, 1) if (not pool and h % 2 == 0 and w % 2 == 0) else (0,)):
        shape = (n, h // 2, w // 2, 4, c) if s2d else (n, h, w, c)
        dy32, dy16 = _nan32(*shape), _nan16(*shape)
        g32, b32, g16, b16, ks = _ws(c), _ws(c), _ws(c), _ws(c), _ws(2 * c)
        vad.hip.check(l.vad_bn_act_pool_bwd_t(y32.data_ptr(), 0, *args, d32.data_ptr(), 0, 0, 0, 0, dy32.data_ptr(), s2d, g32.data_ptr(), b32.data_ptr(),
                                              ks.data_ptr(), ws.data_ptr(), n, h, w, c, act, pool, H.stream()))
        vad.hip.check(l.vad_bn_act_pool_bwd_t(y16.data_ptr(), 1, *args, d16.data_ptr(), 0, 0, 0, 0, dy16.data_ptr(), s2d, g16.data_ptr(), b16.data_ptr(),
                                              ks.data_ptr(), ws.data_ptr(), n, h, w, c, act, pool, H.stream()))
        assert torch.equal(dy16, dy32.to(torch.bfloat16))
        assert torch.equal(g16, g32) and torch.equal(b16, b32)              # the sums are fp32 sums of the same terms in the same order
    s32, s16 = _ws(c), _ws(c)
    vad.hip.check(l.vad_chan_sum_t(d32.data_ptr(), 0, n * oh * ow, c, s32.data_ptr(), ws.data_ptr(), H.stream()))
    vad.hip.check(l.vad_chan_sum_t(d16.data_ptr(), 1, n * oh * ow, c, s16.data_ptr(), ws.data_ptr(), H.stream()))
    assert torch.equal(s16, s32)


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 8, 8, 32, 32), (3, 6, 10, 64, 32), (2, 16, 16, 256, 512), (6, 32, 32, 32, 64),
                                            (2, 3, 3, 64, 128), (4, 7, 7, 128, 64), (2, 20, 36, 64, 128), (5, 9, 70, 128, 64), (7, 5, 16, 32, 128)])
def test_conv3x3_on_bf16_tensors_equals_the_bf16_operand_kernels(vad, n, h, w, cin, cout):
    """forward + BatchNorm partial sums, data gradient and weight gradient of a 3x3 convolution"""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(cin + cout + h)
    a16, a32 = _rep(rng.standard_normal((n, h, w, cin)))
    g16, g32 = _rep(rng.standard_normal((n, h, w, cout)))
    wt = H.dev(rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin))
    bias = H.dev(rng.standard_normal(cout) * 0.1)
    fwd, dgr = _ws(l.vad_pack_conv3x3_floats(cout, cin)), _ws(l.vad_pack_conv3x3_floats(cin, cout))
    fwd3, dgr3 = torch.empty_like(fwd), torch.empty_like(dgr)
    vad.hip.check(l.vad_train_pack_conv3x3(wt.data_ptr(), cout, cin, fwd.data_ptr(), dgr.data_ptr(), BF16, H.stream()))
    vad.hip.check(l.vad_train_pack_conv3x3(wt.data_ptr(), cout, cin, fwd3.data_ptr(), dgr3.data_ptr(), BF16S, H.stream()))
    assert torch.equal(fwd, fwd3) and torch.equal(dgr, dgr3)                 # one operand layout for both bf16 modes
    o32, o16 = _nan32(n, h, w, cout), _nan16(n, h, w, cout)
    vad.hip.check(l.vad_conv3x3(a32.data_ptr(), 0, fwd.data_ptr(), bias.data_ptr(), o32.data_ptr(), 0, n, h, w, cin, cout, 0, 0, BF16, H.stream()))
    vad.hip.check(l.vad_conv3x3(a16.data_ptr(), 0, fwd.data_ptr(), bias.data_ptr(), o16.data_ptr(), 0, n, h, w, cin, cout, 0, 0, BF16S, H.stream()))
    assert torch.equal(o16, o32.to(torch.bfloat16))
    zero = torch.zeros(max(cin, cout), device="cuda")
    d32, d16 = _nan32(n, h, w, cin), _nan16(n, h, w, cin)
    vad.hip.check(l.vad_conv3x3(g32.data_ptr(), 0, dgr.data_ptr(), zero.data_ptr(), d32.data_ptr(), 0, n, h, w, cout, cin, 0, 0, BF16, H.stream()))
    vad.hip.check(l.vad_conv3x3(g16.data_ptr(), 0, dgr.data_ptr(), zero.data_ptr(), d16.data_ptr(), 0, n, h, w, cout, cin, 0, 0, BF16S, H.stream()))
    assert torch.equal(d16, d32.to(torch.bfloat16))
    ws = _ws(l.vad_conv_wgrad_ws_floats(n, h, 9, cin, cout))
    w32, w16 = _nan32(cout, cin, 3, 3), _nan32(cout, cin, 3, 3)
    vad.hip.check(l.vad_conv_wgrad(a32.data_ptr(), g32.data_ptr(), w32.data_ptr(), ws.data_ptr(), n, h, w, cin, cout, 9, 0, BF16, H.stream()))
    try:
        l.vad_debug_set_wgrad_pairs(0)
        vad.hip.check(l.vad_conv_wgrad(a16.data_ptr(), g16.data_ptr(), w16.data_ptr(), ws.data_ptr(), n, h, w, cin, cout, 9, 0, BF16S, H.stream()))
        assert torch.equal(w16, w32)                                             # parameter gradients stay fp32
        # the other kernel forms, each where it applies (else the call falls through to the next lower one): 1 = paired channels
        # (dword loads, 64 x 64 wave tiles, one item per kernel row), 2 = its LDS-staged work-group form, 3 (default) = the
        # row-ring kernel: the same bf16 products, fp32 sums split differently over the image
        for mode in (1, 2, 3):
            l.vad_debug_set_wgrad_pairs(mode)
            w16p = _nan32(cout, cin, 3, 3)
            vad.hip.check(l.vad_conv_wgrad(a16.data_ptr(), g16.data_ptr(), w16p.data_ptr(), ws.data_ptr(), n, h, w, cin, cout, 9, 0, BF16S, H.stream()))
            assert float((w16p - w32).abs().max()) < 2e-5 * float(w32.abs().max()), mode
    finally:
        l.vad_debug_set_wgrad_pairs(3)
    # and the bf16-operand kernel itself is held to float64 on these inputs (products exact, fp32 sums)
    ref = torch.nn.functional.conv2d(a32.double().permute(0, 3, 1, 2).cpu(), wt.to(torch.bfloat16).double().cpu(), bias.double().cpu(), padding=1)
    assert float((o32.permute(0, 3, 1, 2).cpu().double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 4, 4, 128, 128), (3, 8, 6, 128, 64), (2, 16, 16, 64, 32), (4, 7, 7, 32, 128), (2, 5, 3, 128, 64)])
def test_convt2x2_on_bf16_tensors(vad, n, h, w, cin, cout):
    """forward (+ statistics path), weight gradient from the space-to-depth gradient, 1x1 data gradient on bf16 operands"""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(cin * 3 + cout + h)
    a16, a32 = _rep(rng.standard_normal((n, h, w, cin)))
    g16, g32 = _rep(rng.standard_normal((n, h, w, 4 * cout)))               # space-to-depth gradient
    wt = H.dev(rng.standard_normal((cin, cout, 2, 2)) / np.sqrt(cin))
    bias = H.dev(rng.standard_normal(cout) * 0.1)
    fwd, dgr3 = _ws(l.vad_pack_convt2x2_floats(cin, cout)), _ws(l.vad_pack_conv1x1_floats(cin, 4 * cout))
    vad.hip.check(l.vad_train_pack_convt2x2(wt.data_ptr(), cin, cout, fwd.data_ptr(), dgr3.data_ptr(), BF16S, H.stream()))
    o32, o16 = _nan32(n, 2 * h, 2 * w, cout), _nan16(n, 2 * h, 2 * w, cout)
    vad.hip.check(l.vad_convt2x2(a32.data_ptr(), 0, fwd.data_ptr(), bias.data_ptr(), o32.data_ptr(), 0, n, h, w, cin, cout, 0, BF16, H.stream()))
    vad.hip.check(l.vad_convt2x2(a16.data_ptr(), 0, fwd.data_ptr(), bias.data_ptr(), o16.data_ptr(), 0, n, h, w, cin, cout, 0, BF16S, H.stream()))
    assert torch.equal(o16, o32.to(torch.bfloat16))
    ws = _ws(l.vad_conv_wgrad_ws_floats(n, h, 1, cin, 4 * cout))
    w32, w16 = _nan32(cin, cout, 2, 2), _nan32(cin, cout, 2, 2)
    vad.hip.check(l.vad_conv_wgrad(a32.data_ptr(), g32.data_ptr(), w32.data_ptr(), ws.data_ptr(), n, h, w, cin, 4 * cout, 1, 1, BF16, H.stream()))
    try:
        l.vad_debug_set_wgrad_pairs(0)
        vad.hip.check(l.vad_conv_wgrad(a16.data_ptr(), g16.data_ptr(), w16.data_ptr(), ws.data_ptr(), n, h, w, cin, 4 * cout, 1, 1, BF16S, H.stream()))
        assert torch.equal(w16, w32)
        for mode in (1, 2, 3):
            l.vad_debug_set_wgrad_pairs(mode)
            w16p = _nan32(cin, cout, 2, 2)
            vad.hip.check(l.vad_conv_wgrad(a16.data_ptr(), g16.data_ptr(), w16p.data_ptr(), ws.data_ptr(), n, h, w, cin, 4 * cout, 1, 1, BF16S, H.stream()))
            assert float((w16p - w32).abs().max()) < 2e-5 * float(w32.abs().max()), mode
    finally:
        l.vad_debug_set_wgrad_pairs(3)
    # 1x1 data gradient (K = 4*cout) on bf16 operands: against float64 on the same bf16 values; products are exact, the sum
    # is fp32, the result is rounded to bf16 (2^-9 relative)
    # (pixel counts that are not a multiple of 16 run as one ragged frame: (2, 5, 3) -> 30 pixels)
    zero = torch.zeros(cin, device="cuda")
    guard = torch.full((n * h * w + 32, cin), 7.0, dtype=torch.bfloat16, device="cuda")       # room behind the tensor: must stay untouched
    da = guard[:n * h * w]
    vad.hip.check(l.vad_conv1x1_p(g16.data_ptr(), dgr3.data_ptr(), zero.data_ptr(), da.data_ptr(), n * h * w, 4 * cout, cin, BF16S, H.stream()))
    wq = wt.to(torch.bfloat16).double().permute(2, 3, 1, 0).reshape(4 * cout, cin)          # [q*cout + co][ci]
    ref = g32.double().reshape(-1, 4 * cout) @ wq
    err = (da.double().reshape(-1, cin) - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5).all()), float(err.max())
    assert bool((guard[n * h * w:] == 7.0).all())


@pytest.mark.parametrize("npix,cin,cout", [(512, 64, 128), (256, 128, 32), (1024, 32, 96), (24, 64, 64), (1000, 32, 64)])
def test_conv1x1_on_bf16_tensors(vad, npix, cin, cout):
    """VideoAutoencoder.proj (models/video_autoencoder.py:311) forward and data gradient in the bf16-tensor mode"""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(npix + cin)
    a16, a32 = _rep(rng.standard_normal((npix, cin)))
    wt = H.dev(rng.standard_normal((cout, cin)) / np.sqrt(cin))
    bias = H.dev(rng.standard_normal(cout) * 0.1)
    fwd, dgr = _ws(l.vad_pack_conv1x1_floats(cout, cin)), _ws(l.vad_pack_conv1x1_floats(cin, cout))
    vad.hip.check(l.vad_train_pack_conv1x1_p(wt.data_ptr(), cout, cin, fwd.data_ptr(), dgr.data_ptr(), BF16S, H.stream()))
    out = _nan16(npix, cout)
    vad.hip.check(l.vad_conv1x1_p(a16.data_ptr(), fwd.data_ptr(), bias.data_ptr(), out.data_ptr(), npix, cin, cout, BF16S, H.stream()))
    ref = a32.double() @ wt.to(torch.bfloat16).double().t() + bias.double()
    err = (out.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5).all()), float(err.max())
    g16, g32 = _rep(rng.standard_normal((npix, cout)))
    zero = torch.zeros(cin, device="cuda")
    da = _nan16(npix, cin)
    vad.hip.check(l.vad_conv1x1_p(g16.data_ptr(), dgr.data_ptr(), zero.data_ptr(), da.data_ptr(), npix, cout, cin, BF16S, H.stream()))
    ref = g32.double() @ wt.to(torch.bfloat16).double()
    err = (da.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5).all()), float(err.max())


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 32, 48), (5, 12, 80), (2, 64, 256), (1, 8, 272), (1, 4, 768)])
def test_routed_first_layer_weight_gradient(vad, n, h, w):
    """Round 4: the bf16-tensor step forms the first layer's weight gradient from the POOLED gradient, one routing byte per pooled
    element and the Gram matrix of the input patches (csrc/train_ops.hip conv_c3_wgrad_routed_kernel) instead of writing the dense
    conv-output gradient and reading it back.  Against a float64 evaluation of the same layer (Conv2d(3->32) on bf16 operands ->
    BatchNorm with batch statistics -> LeakyReLU(0.2) -> MaxPool2, every decision taken from the stored bf16 conv output as the
    kernels take it): the routed form and the form it
    replaces (pass B + the plain weight gradient, vad_debug_set_c3_routed(0)) are both within bf16 accuracy, and the routed one -
    which never rounds the dense gradient to bf16 - is the closer of the two or equal to it within noise."""
    import hip_helpers as H
    import torch.nn.functional as F
    l, rng = vad.hip.lib(), _rng(7 * n + h + w)
    x = H.dev(rng.uniform(-1, 1, (n, 3, h, w)))
    w0 = (rng.standard_normal((32, 3, 3, 3)) * 0.3).astype(np.float32)
    b0 = (rng.standard_normal(32) * 0.1).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, 32).astype(np.float32), (rng.standard_normal(32) * 0.1).astype(np.float32)
    dout16, dout32 = _rep(rng.standard_normal((n, h // 2, w // 2, 32)) * 1e-3)
    # forward exactly as the step runs it: bf16 operands, fp32 accumulation, statistics from the fp32 values, y stored as bf16
    wp, bo = H.pack_conv3x3(w0, b0)
    y16 = _nan16(n, h, w, 32)
    vad.hip.check(l.vad_conv3x3_c3_bf16op(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), y16.data_ptr(), n, h, w, 32, H.stream()))
    # float64 evaluation of what the kernels define: statistics of the fp32 conv output, every decision (pooling argmax, sign) and
    # xhat taken from the STORED bf16 y, dy = sc (dz - k1 - xhat k2), dW = sum_p dy[p] (x) X[p] with the exact fp32 frames
    xq, wq = x.to(torch.bfloat16).double().cpu(), torch.from_numpy(w0).to(torch.bfloat16).double()
    yex = F.conv2d(xq, wq, torch.from_numpy(b0).double(), padding=1)
    mean, var = yex.mean((0, 2, 3)), yex.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    mean, invstd = mean.float().double(), invstd.float().double()                      # (the kernels get them as fp32)
    g64, be64 = torch.from_numpy(gamma).double(), torch.from_numpy(beta).double()
    yq = y16.double().cpu().permute(0, 3, 1, 2)
    xhat = (yq - mean[None, :, None, None]) * invstd[None, :, None, None]
    zq = F.leaky_relu(xhat * g64[None, :, None, None] + be64[None, :, None, None], 0.2)
    zz = zq.view(n, 32, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 32, h // 2, w // 2, 4)
    am = zz.argmax(-1)
    val = zz.gather(-1, am[..., None])[..., 0]
    gz = dout32.double().cpu().permute(0, 3, 1, 2) * torch.where(val > 0, 1.0, 0.2)
    dz = torch.zeros(n, 32, h // 2, w // 2, 4, dtype=torch.float64).scatter_(-1, am[..., None], gz[..., None])
    dz = dz.view(n, 32, h // 2, w // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 32, h, w)
    m_ = n * h * w
    k1, k2 = dz.sum((0, 2, 3)) / m_, (dz * xhat).sum((0, 2, 3)) / m_
    dy = (g64 * invstd)[None, :, None, None] * (dz - k1[None, :, None, None] - xhat * k2[None, :, None, None])
    X = F.unfold(x.double().cpu(), 3, padding=1).permute(0, 2, 1).reshape(-1, 27)
    ref = (dy.permute(0, 2, 3, 1).reshape(-1, 32).t() @ X).reshape(32, 3, 3, 3)
    stats = torch.cat([mean, invstd]).float().cuda()
    gd, bd, w0d, b0d = H.dev(gamma), H.dev(beta), H.dev(w0), H.dev(b0)
    dg, db, ks = _ws(32), _ws(32), _ws(64)
    cws = _ws(l.vad_chan_ws_floats(n * h * w, 32))
    # the form it replaces: pass A + pass B (dy, bf16), then the plain weight gradient
    dy16 = _nan16(n, h, w, 32)
    vad.hip.check(l.vad_bn_act_pool_bwd_t(y16.data_ptr(), 1, stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dout16.data_ptr(), 0, 0, 0, 0,
                                          dy16.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), ks.data_ptr(), cws.data_ptr(), n, h, w, 32, 1, 1, H.stream()))
    wws = _ws(max(l.vad_conv_c3_wgrad_ws_floats(n, h, 32), l.vad_conv_c3_wgrad_routed_ws_floats(n, h)))
    dw_old = _nan32(32, 3, 3, 3)
    vad.hip.check(l.vad_conv_c3_wgrad_t(x.data_ptr(), dy16.data_ptr(), 1, dw_old.data_ptr(), wws.data_ptr(), n, h, w, 32, H.stream()))
    # routed: pass A with codes, no dy
    codes = torch.full((n * (h // 2) * (w // 2), 32), 255, dtype=torch.uint8, device="cuda")
    dg2, db2, ks2 = _ws(32), _ws(32), _ws(64)
    vad.hip.check(l.vad_bn_act_pool_bwd_codes_t(y16.data_ptr(), 1, stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dout16.data_ptr(), 0, 0, 0, 0,
                                                None, 0, dg2.data_ptr(), db2.data_ptr(), ks2.data_ptr(), cws.data_ptr(), n, h, w, 32, 1, 1,
                                                codes.data_ptr(), H.stream()))
    assert torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(ks2, ks) and int(codes.max()) <= 7
    assert l.vad_conv_c3_wgrad_routed_ok(h, w, 32) == 1
    dw_new = _nan32(32, 3, 3, 3)
    vad.hip.check(l.vad_conv_c3_wgrad_routed(x.data_ptr(), dout16.data_ptr(), 1, codes.data_ptr(), w0d.data_ptr(), b0d.data_ptr(), stats.data_ptr(),
                                             gd.data_ptr(), ks.data_ptr(), dw_new.data_ptr(), wws.data_ptr(), n, h, w, 32, H.stream()))
    scale = float(ref.abs().max())
    e_old = float((dw_old.double().cpu() - ref).abs().max()) / scale
    e_new = float((dw_new.double().cpu() - ref).abs().max()) / scale
    print(f"first-layer dW vs float64 autograd ({n}x{h}x{w}): routed {e_new:.2e}, pass B + plain {e_old:.2e} of max |dW|")
    assert bool(torch.isfinite(dw_new).all()) and e_new < 2e-2 and e_old < 2e-2, (e_new, e_old)
    assert e_new < 1.5 * e_old + 2e-3, (e_new, e_old)


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 32, 48), (2, 18, 24)])
def test_first_and_last_layer_on_bf16_tensors(vad, n, h, w):
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(n + h)
    x = H.dev(rng.uniform(-1, 1, (n, 3, h, w)))
    w0 = rng.standard_normal((32, 3, 3, 3)).astype(np.float32) * 0.2
    wp, bo = H.pack_conv3x3(w0, rng.standard_normal(32).astype(np.float32) * 0.1)
    o32, o16 = _nan32(n, h, w, 32), _nan16(n, h, w, 32)
    vad.hip.check(l.vad_conv3x3_c3(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), o32.data_ptr(), n, h, w, 32, 0, 0, H.stream()))
    vad.hip.check(l.vad_conv3x3_c3_bf16(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), o16.data_ptr(), n, h, w, 32, H.stream()))
    assert torch.equal(o16, o32.to(torch.bfloat16))
    # ... and with bf16 MFMA operands (the form the bf16-tensor step runs since round 4): on bf16-REPRESENTABLE frames and weights
    # the products are exact in fp32, so the result is the exact-fp32 kernel's up to the summation order of 27 terms - held to
    # one bf16 ulp with almost every element equal; on arbitrary frames it is the same kernel after rounding both operands
    xr, wr = x.to(torch.bfloat16).float(), torch.from_numpy(w0).to(torch.bfloat16).float().numpy()
    wpr, _ = H.pack_conv3x3(wr, np.zeros(32, np.float32))
    o32r, o16op, o16op_any = _nan32(n, h, w, 32), _nan16(n, h, w, 32), _nan16(n, h, w, 32)
    vad.hip.check(l.vad_conv3x3_c3(xr.data_ptr(), wpr.data_ptr(), bo.data_ptr(), o32r.data_ptr(), n, h, w, 32, 0, 0, H.stream()))
    vad.hip.check(l.vad_conv3x3_c3_bf16op(xr.data_ptr(), wpr.data_ptr(), bo.data_ptr(), o16op.data_ptr(), n, h, w, 32, H.stream()))
    vad.hip.check(l.vad_conv3x3_c3_bf16op(x.data_ptr(), wp.data_ptr(), bo.data_ptr(), o16op_any.data_ptr(), n, h, w, 32, H.stream()))
    want = o32r.to(torch.bfloat16)
    same = (o16op == want).float().mean().item()
    ulp = (o16op.float() - want.float()).abs() / (want.float().abs() * 2.0 ** -7 + 1e-30)
    assert same > 0.995 and float(ulp.max()) <= 1.0 + 1e-6, (same, float(ulp.max()))
    assert torch.equal(o16op_any, o16op)                    # rounding the operands first changes nothing: the kernel does exactly that
    # first-layer weight gradient with a bf16 gradient tensor
    g16, g32 = _rep(rng.standard_normal((n, h, w, 32)))
    ws = _ws(l.vad_conv_c3_wgrad_ws_floats(n, h, 32))
    d32, d16 = _nan32(32, 3, 3, 3), _nan32(32, 3, 3, 3)
    vad.hip.check(l.vad_conv_c3_wgrad_t(x.data_ptr(), g32.data_ptr(), 0, d32.data_ptr(), ws.data_ptr(), n, h, w, 32, H.stream()))
    vad.hip.check(l.vad_conv_c3_wgrad_t(x.data_ptr(), g16.data_ptr(), 1, d16.data_ptr(), ws.data_ptr(), n, h, w, 32, H.stream()))
    assert torch.equal(d16, d32)
    # last layer + loss: ConvTranspose2d(32->3) + Tanh + MSELoss forward and backward on a bf16 activation
    h2, w2 = h // 2, w // 2
    r16, r32 = _rep(rng.standard_normal((n, h2, w2, 32)))
    wt, bt = H.dev(rng.standard_normal((32, 3, 2, 2)) * 0.2), H.dev(rng.standard_normal(3) * 0.1)
    tws = _ws(l.vad_convt_to3_mse_ws_floats(n, h2, w2))
    outs = {}
    for io, r in ((0, r32), (1, r16)):
        dt = torch.bfloat16 if io else torch.float32
        rec, loss, db = _nan32(n, 3, h, w), _nan32(1), _nan32(3)
        din = torch.full((n, h2, w2, 32), float("nan"), dtype=dt, device="cuda")
        dpre = torch.full((n * h2 * w2, 32), float("nan"), dtype=dt, device="cuda")
        vad.hip.check(l.vad_convt_to3_mse_t(r.data_ptr(), io, wt.data_ptr(), bt.data_ptr(), x.data_ptr(), rec.data_ptr(), din.data_ptr(),
                                            dpre.data_ptr(), loss.data_ptr(), db.data_ptr(), tws.data_ptr(), n, h2, w2, 1.0, H.stream()))
        outs[io] = (rec, loss, din, dpre, db)
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    assert torch.equal(outs[1][2], outs[0][2].to(torch.bfloat16)) and torch.equal(outs[1][3], outs[0][3].to(torch.bfloat16))
    # the bias gradient sums the STORED dpre: bf16-rounded terms in the bf16 form
    want = outs[1][3].float().reshape(-1, 32)[:, :12].double().sum(0).reshape(4, 3).sum(0)
    assert float((outs[1][4].double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("nb,hw,hid,first", [(3, 16, 64, False), (2, 64, 128, True), (4, 4, 32, False)])
def test_lstm_gates_on_bf16_tensors(vad, nb, hw, hid, first):
    """The gates are stored (and used) as bf16, the cell state stays fp32: the state update must use the gate values as
    stored - the ones the backward reads - so forward and backward stay consistent."""
    import hip_helpers as H
    l, rng = vad.hip.lib(), _rng(nb + hid)
    z16, z32 = _rep(rng.standard_normal((nb * hw, 4 * hid)))
    cp = None if first else H.dev(rng.standard_normal((nb * hw, hid)))
    c_out, h1 = _nan32(nb * hw, hid), _nan16(nb, hw, hid)
    zz = z16.clone()
    vad.hip.check(l.vad_lstm_gates_fwd_t(zz.data_ptr(), 1, vad.hip.ptr(cp), c_out.data_ptr(), h1.data_ptr(), 0, 0, None, 0, 0, nb, hw, hid, H.stream()))
    i, f, g, o = z32[:, :hid], z32[:, hid:2 * hid], z32[:, 2 * hid:3 * hid], z32[:, 3 * hid:]
    gates = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)], 1)
    assert float((zz.float() - gates).abs().max()) < 2.0 ** -8                      # bf16 rounding of values in [-1, 1]
    gi, gf, gg, go = (zz.float()[:, k * hid:(k + 1) * hid] for k in range(4))
    cn = gf * (cp if cp is not None else 0.0) + gi * gg
    assert float((c_out - cn).abs().max()) < 2e-6 * max(1.0, float(cn.abs().max()))  # fp32 state from the STORED gates
    hn = go * torch.tanh(c_out)
    assert float((h1.float().reshape(-1, hid) - hn).abs().max()) < 2.0 ** -8
    # backward: bf16 form == round(fp32 form) on the same stored gates
    d16, d32 = _rep(rng.standard_normal((nb, hw, hid)))
    dcn = H.dev(rng.standard_normal((nb * hw, hid)))
    outs = {}
    for io, (gt, dh) in ((0, (zz.float().contiguous(), d32)), (1, (zz, d16))):
        dz = torch.full((nb * hw, 4 * hid), float("nan"), dtype=torch.bfloat16 if io else torch.float32, device="cuda")
        dcp = _nan32(nb * hw, hid)
        vad.hip.check(l.vad_lstm_gates_bwd_t(gt.data_ptr(), io, vad.hip.ptr(cp), c_out.data_ptr(), dh.data_ptr(), 0, 0, None, 0, 0,
                                             dcn.data_ptr(), dz.data_ptr(), dcp.data_ptr(), nb, hw, hid, H.stream()))
        outs[io] = (dz, dcp)
    assert torch.equal(outs[1][0], outs[0][0].to(torch.bfloat16)) and torch.equal(outs[1][1], outs[0][1])


def test_bf16_tensor_step_is_refused_before_its_first_launch_without_the_statistics_kernels(vad):
    """The bf16-tensor step has no stand-alone statistics pass: it needs the convolutions' own BatchNorm partial sums.  With the
    persistent kernels switched off (`vad_debug_set_conv_variant` bit 0 cleared) the step must be refused BEFORE anything is
    launched - parameters, gradients and running statistics untouched - not abort half way through its launch sequence."""
    m = vad.VideoAutoencoder(latent_dim=32, lstm_hidden_dim=32, lstm_num_layers=1).cuda()
    tr = vad.VideoTrainer(m, precision="bf16")
    x = vad.scoring.synth_frames_device(3, 0, 4, 32, 32).view(2, 2, 3, 32, 32)
    flat, grad, running = tr.flat.clone(), tr.grad.clone(), tr.running.clone()
    l = vad.hip.lib()
    try:
        l.vad_debug_set_conv_variant(0)
        with pytest.raises(vad.hip.VadError, match="BatchNorm partial sums"):
            tr.forward_backward(x)
    finally:
        l.vad_debug_set_conv_variant(1)
    torch.cuda.synchronize()
    assert torch.equal(tr.flat, flat) and torch.equal(tr.grad, grad) and torch.equal(tr.running, running)
    loss, _ = tr.forward_backward(x)                     # and the same trainer steps normally afterwards
    assert torch.isfinite(loss)
