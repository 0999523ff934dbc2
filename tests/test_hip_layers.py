"""GPU parity: every layer kernel of libvad_hip.so (called through the C ABI) against the CPU oracle on
the same seeded inputs.  fp32 tolerances are written next to each check."""
import numpy as np
import pytest
import torch

from conftest import max_abs
from oracle import c_oracle

pytestmark = pytest.mark.gpu

ATOL = 3e-5   # outputs are O(1); fp32 accumulation-order noise over K <= 2304 terms


def _rng(seed):
    return np.random.default_rng(seed)


def _conv_params(rng, cout, cin, k=3):
    w = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    bn = [rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.1, rng.standard_normal(cout) * 0.1,
          rng.uniform(0.25, 1.75, cout)]
    return w, b, [a.astype(np.float32) for a in bn]


def _ref_conv(x, w, b, bn, act, pool):
    y = c_oracle.conv2d(x, w, b, 3)
    if bn is not None:
        y = c_oracle.batchnorm_eval(y, *bn)
    if act == 1:
        y = np.where(y > 0, y, np.float32(0.2) * y)
    elif act == 2:
        y = np.maximum(y, 0)
    if pool:
        y = c_oracle.maxpool2(y)
    return y


@pytest.mark.parametrize("cin,cout,h,w,act,pool", [
    (3, 32, 16, 16, 1, False), (3, 32, 20, 36, 1, True), (3, 32, 19, 23, 2, False), (3, 64, 32, 32, 0, True),
    (32, 32, 16, 16, 1, True), (32, 32, 18, 22, 2, False), (32, 64, 24, 40, 1, False), (64, 64, 16, 32, 1, True),
    (64, 128, 8, 8, 1, False), (128, 128, 12, 20, 2, False), (128, 256, 4, 4, 1, False), (256, 256, 6, 10, 1, True),
    (32, 96, 10, 10, 0, False), (64, 192, 5, 7, 2, False),
])
def test_conv3x3(cin, cout, h, w, act, pool):
    import hip_helpers as H
    rng = _rng(cin * 1000 + cout + h)
    x = rng.standard_normal((2, cin, h, w)).astype(np.float32)
    wt, b, bn = _conv_params(rng, cout, cin)
    got = H.conv3x3(x, wt, b, bn, act, pool)
    ref = _ref_conv(x, wt, b, bn, act, pool)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert max_abs(got, ref) < ATOL
    # Every exact-fp32 form of the layer is the same bits: whatever the cost model picked above, the persistent 32x32x2 kernels
    # only (bit 6: never the gate-split small-grid kernel), the gate-split kernel wherever it applies (bit 7: 8-wave form; cin >= 64,
    # cout % 64 == 0), and one tile per work-group (0).
    l = H.hip.lib()
    try:
        for bits in (1 | 64, 1 | 128, 0):
            l.vad_debug_set_conv_variant(bits)
            assert np.array_equal(got, H.conv3x3(x, wt, b, bn, act, pool)), bits
    finally:
        l.vad_debug_set_conv_variant(1)


@pytest.mark.parametrize("cin,cout,h,w,act,pool,n", [
    (32, 32, 16, 16, 1, True, 2), (32, 32, 18, 22, 2, False, 3), (32, 64, 24, 40, 1, False, 2), (64, 64, 16, 32, 1, True, 5),
    (64, 128, 8, 8, 1, False, 2), (128, 128, 12, 20, 2, False, 2), (128, 256, 4, 4, 1, False, 7), (256, 256, 6, 10, 1, True, 2),
    (32, 96, 10, 10, 0, False, 2), (64, 192, 2, 6, 2, False, 2), (64, 64, 128, 128, 1, True, 3), (128, 128, 32, 32, 2, False, 9),
    (64, 64, 5, 9, 1, False, 3), (128, 256, 3, 3, 0, False, 4), (32, 32, 7, 17, 2, False, 2), (256, 128, 1, 1, 0, False, 3),
])
def test_conv3x3_winograd(cin, cout, h, w, act, pool, n):
    """Winograd F(2x2,3x3) on the exact-fp32 MFMA (opt-in arithmetic): the same layer as `test_conv3x3`, against the same
    oracle.  All-fp32, but 16 products per 2x2 outputs instead of 36 with +-1 / 0.5 transforms around them: not bit-identical to
    the direct kernel, within fp32 rounding of it - held to the direct kernels' own bound against the oracle.  Covers ragged
    tiles (H, W not multiples of 8 / 16), one and two N-tiles per wave, every activation, pooling, frames > work-group groups."""
    import hip_helpers as H
    rng = _rng(cin * 1000 + cout + h + 77)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt, b, bn = _conv_params(rng, cout, cin)
    got = H.conv3x3_wino(x, wt, b, bn, act, pool)
    ref = _ref_conv(x, wt, b, bn, act, pool)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert max_abs(got, ref) < ATOL
    direct = H.conv3x3(x, wt, b, bn, act, pool)
    assert max_abs(got, direct) < ATOL and not np.array_equal(got, direct)       # another rounding order, stated as such


@pytest.mark.parametrize("h,w", [(16, 16), (32, 48), (22, 18), (64, 64)])
def test_conv3x3_c3_fused(h, w):
    """Fused enc1 block == the two separate layers of the oracle (halo recompute, zero padding of conv #2)."""
    import hip_helpers as H
    rng = _rng(h * 100 + w)
    x = rng.uniform(-1, 1, (2, 3, h, w)).astype(np.float32)
    w0, b0, bn0 = _conv_params(rng, 32, 3)
    w1, b1, bn1 = _conv_params(rng, 32, 32)
    ref = _ref_conv(_ref_conv(x, w0, b0, bn0, 1, False), w1, b1, bn1, 1, True)
    got = H.conv3x3_c3_fused(x, w0, b0, bn0, w1, b1, bn1)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert max_abs(got, ref) < ATOL
    l = H.hip.lib()                       # persistent kernel == one-tile-per-work-group kernel, bit for bit
    try:
        l.vad_debug_set_conv_variant(0)
        one_tile = H.conv3x3_c3_fused(x, w0, b0, bn0, w1, b1, bn1)
    finally:
        l.vad_debug_set_conv_variant(1)
    assert np.array_equal(got, one_tile)


def test_conv3x3_no_bn_and_frame_strides():
    import hip_helpers as H
    rng = _rng(5)
    x = rng.standard_normal((3, 32, 8, 8)).astype(np.float32)
    wt, b, _ = _conv_params(rng, 32, 32)
    assert max_abs(H.conv3x3(x, wt, b, None, 0, False), _ref_conv(x, wt, b, None, 0, False)) < ATOL


@pytest.mark.parametrize("cin,cout,h,w,act", [(256, 128, 4, 4, 2), (128, 64, 8, 6, 2), (64, 32, 16, 16, 2),
                                              (32, 32, 9, 13, 2), (32, 64, 3, 5, 0), (128, 128, 2, 2, 1)])
@pytest.mark.parametrize("precision", [0, 1])
def test_convt2x2(vad, cin, cout, h, w, act, precision):
    """precision 1 = split-fp16 operands (weights packed AND kernels launched with precision = VAD_PREC_SPLIT)."""
    import hip_helpers as H
    H.PRECISION = precision
    try:
        _check_convt2x2(H, cin, cout, h, w, act)
    finally:
        H.PRECISION = 0


def _check_convt2x2(H, cin, cout, h, w, act):
    rng = _rng(cin + cout * 7 + h)
    x = rng.standard_normal((3, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cin, cout, 2, 2)) * np.sqrt(1.0 / cin)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    bn = [a.astype(np.float32) for a in (rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.1,
                                         rng.standard_normal(cout) * 0.1, rng.uniform(0.25, 1.75, cout))]
    ref = c_oracle.batchnorm_eval(c_oracle.convt2x2(x, wt, b), *bn)
    ref = np.where(ref > 0, ref, np.float32(0.2) * ref) if act == 1 else (np.maximum(ref, 0) if act == 2 else ref)
    got = H.convt2x2(x, wt, b, bn, act)
    assert np.isfinite(got).all() and max_abs(got, ref) < ATOL


@pytest.mark.parametrize("cin,cout", [(64, 32), (128, 128), (64, 96)])
def test_conv1x1(cin, cout):
    import hip_helpers as H
    rng = _rng(cin + cout)
    x = rng.standard_normal((2, cin, 5, 7)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * np.sqrt(1.0 / cin)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    assert max_abs(H.conv1x1(x, wt, b), c_oracle.conv2d(x, wt, b, 1)) < ATOL


def test_convlstm_step_golden(vad, golden):
    """One ConvLSTMCell step against the REFERENCE's own output (tests/golden/convlstm_unit.npz)."""
    import hip_helpers as H
    g = golden("convlstm_unit.npz")
    st = vad.synth.synthetic_state({"conv.weight": (256, 96, 3, 3), "conv.bias": (256,)}, 31)
    h1, c1 = H.convlstm_step(g["x"], g["h"], g["c"], st["conv.weight"], st["conv.bias"])
    assert max_abs(h1, g["h1"]) < 2e-5 and max_abs(c1, g["c1"]) < 2e-5


@pytest.mark.parametrize("cx,hid,h,w,zero_state", [(32, 64, 8, 8, True), (128, 128, 16, 16, False), (64, 64, 5, 9, False)])
def test_convlstm_step_oracle(cx, hid, h, w, zero_state):
    import hip_helpers as H
    rng = _rng(cx + hid + h)
    x = rng.standard_normal((2, cx, h, w)).astype(np.float32)
    wt = (rng.standard_normal((4 * hid, cx + hid, 3, 3)) * np.sqrt(1.0 / ((cx + hid) * 9))).astype(np.float32)
    b = (rng.standard_normal(4 * hid) * 0.1).astype(np.float32)
    if zero_state:
        hp = cp = None
        h0 = c0 = np.zeros((2, hid, h, w), np.float32)
    else:
        hp = h0 = (rng.standard_normal((2, hid, h, w)) * 0.5).astype(np.float32)
        cp = c0 = rng.standard_normal((2, hid, h, w)).astype(np.float32)
    rh, rc = c_oracle.convlstm_cell(x, h0, c0, wt, b)
    gh, gc = H.convlstm_step(x, hp, cp, wt, b)
    assert max_abs(gh, rh) < 2e-5 and max_abs(gc, rc) < 2e-5


@pytest.mark.parametrize("n,hid,h,w,zero_state", [(2, 128, 16, 16, False), (2, 128, 16, 16, True), (3, 64, 6, 10, False), (70, 64, 4, 4, False),
                                                  (1, 64, 2, 2, True), (2, 64, 3, 5, False), (3, 128, 7, 7, False),
                                                  (40, 128, 16, 16, False), (300, 64, 4, 4, True), (70, 64, 5, 7, False)])   # the fused-cell form (grids that fill the chip)
def test_convlstm_step_winograd(n, hid, h, w, zero_state):
    """ConvLSTMCell step with the gate convolution as ONE two-source Winograd launch (x and h halves of K; the h half skipped at
    t = 0) + the pointwise cell (vad_convlstm_step_wino; VAD_PREC_WINO models): against the oracle at the direct kernels' bound,
    and against the direct fused step."""
    import hip_helpers as H
    rng = _rng(n + hid + h + 5)
    x = rng.standard_normal((n, hid, h, w)).astype(np.float32)
    wt = (rng.standard_normal((4 * hid, 2 * hid, 3, 3)) * np.sqrt(1.0 / (2 * hid * 9))).astype(np.float32)
    b = (rng.standard_normal(4 * hid) * 0.1).astype(np.float32)
    hp = None if zero_state else (rng.standard_normal((n, hid, h, w)) * 0.5).astype(np.float32)
    cp = None if zero_state else rng.standard_normal((n, hid, h, w)).astype(np.float32)
    zeros = np.zeros((n, hid, h, w), np.float32)
    rh, rc = c_oracle.convlstm_cell(x, hp if hp is not None else zeros, cp if cp is not None else zeros, wt, b)
    gh, gc = H.convlstm_step_wino(x, hp, cp, wt, b)
    assert np.isfinite(gh).all() and max_abs(gh, rh) < 2e-5 and max_abs(gc, rc) < 2e-5
    dh, dc = H.convlstm_step(x, hp, cp, wt, b)
    assert max_abs(gh, dh) < 2e-5 and max_abs(gc, dc) < 2e-5


@pytest.mark.parametrize("n,cx,hid,h,w,zero_state", [(2, 128, 128, 16, 16, False), (3, 64, 64, 5, 9, False), (1, 32, 64, 3, 4, False),
                                                     (2, 128, 128, 16, 16, True), (5, 128, 64, 7, 18, False)])
def test_convlstm_small_grid_kernel_is_bit_identical(vad, n, cx, hid, h, w, zero_state):
    """The small-grid ConvLSTM kernels (16x16x4 MFMA, used below one work-group per CU: the reference's batch sizes; bit 6 = never
    / bit 7 = always the gate-split form for the smallest grids) against
    the 32x32x2 kernels (vad_debug_set_conv_variant bit 3 = never small; variant 0 = one tile per work-group): torch.equal,
    not a tolerance - it reproduces their k order exactly (csrc/conv_small.h), which is why a clip scored alone equals the
    same clip inside a large batch.  Ragged maps (partial tiles in both directions) and unequal x / h channel counts."""
    import hip_helpers as H
    l = vad.hip.lib()
    rng = _rng(n + cx + hid + h)
    x = rng.standard_normal((n, cx, h, w)).astype(np.float32)
    wt = (rng.standard_normal((4 * hid, cx + hid, 3, 3)) * np.sqrt(1.0 / ((cx + hid) * 9))).astype(np.float32)
    b = (rng.standard_normal(4 * hid) * 0.1).astype(np.float32)
    hp = None if zero_state else (rng.standard_normal((n, hid, h, w)) * 0.5).astype(np.float32)
    cp = None if zero_state else rng.standard_normal((n, hid, h, w)).astype(np.float32)
    try:
        l.vad_debug_set_conv_variant(1 | 64)           # never the gate-split kernel: the 4-gates-per-wave small-grid kernel
        small = H.convlstm_step(x, hp, cp, wt, b)
        l.vad_debug_set_conv_variant(1 | 128)          # the gate-split kernel (one gate per wave, the cell through LDS)
        gate = H.convlstm_step(x, hp, cp, wt, b)
        l.vad_debug_set_conv_variant(1)                # whatever the cost model picks
        auto = H.convlstm_step(x, hp, cp, wt, b)
        l.vad_debug_set_conv_variant(1 | 8)
        big = H.convlstm_step(x, hp, cp, wt, b)
        l.vad_debug_set_conv_variant(0)
        tile = H.convlstm_step(x, hp, cp, wt, b)
    finally:
        l.vad_debug_set_conv_variant(1)
    for other in (gate, auto, big, tile):
        assert np.array_equal(small[0], other[0]) and np.array_equal(small[1], other[1])
    rh, rc = c_oracle.convlstm_cell(x, hp if hp is not None else np.zeros((n, hid, h, w), np.float32),
                                    cp if cp is not None else np.zeros((n, hid, h, w), np.float32), wt, b)
    assert max_abs(small[0], rh) < 2e-5 and max_abs(small[1], rc) < 2e-5


def test_synth_frames_bit_exact(vad):
    """Device generator == numpy generator, bit for bit, including the anomaly patches and offsets."""
    for first, n, h, w, anomalies in [(0, 4, 32, 48, False), (1000, 6, 64, 64, True), (12345, 2, 256, 256, True)]:
        ref = vad.synth.frames(0xC0FFEE, first, n, 3, h, w, anomalies=anomalies)
        got = vad.scoring.synth_frames_device(0xC0FFEE, first, n, h, w, anomalies=anomalies).cpu().numpy()
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))


def test_transposes():
    import hip_helpers as H
    l = H.hip.lib()
    rng = _rng(3)
    x = rng.standard_normal((3, 40, 5, 7)).astype(np.float32)       # NCHW
    xin = H.dev(x)
    out = torch.empty(3, 5, 7, 40, device="cuda")
    H.hip.check(l.vad_nchw_to_nhwc(xin.data_ptr(), out.data_ptr(), 3, 5, 7, 40, H.stream()))
    assert np.array_equal(out.cpu().numpy(), x.transpose(0, 2, 3, 1))
    back = torch.empty(3, 40, 5, 7, device="cuda")
    H.hip.check(l.vad_nhwc_to_nchw(out.data_ptr(), back.data_ptr(), 3, 5, 7, 40, H.stream()))
    assert np.array_equal(back.cpu().numpy(), x)


def test_bad_arguments_are_rejected():
    import hip_helpers as H
    l = H.hip.lib()
    t = torch.zeros(16, device="cuda")
    assert l.vad_conv3x3(t.data_ptr(), 0, t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 1, 4, 4, 24, 32, 0, 0, 0, None) == -1
    assert b"multiples of 32" in l.vad_last_error()
    assert l.vad_conv3x3(t.data_ptr(), 0, t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 1, 5, 4, 32, 32, 0, 1, 0, None) == -1
    assert l.vad_img_score(t.data_ptr(), 1, 30, 32, 256, t.data_ptr(), t.data_ptr(), 64, 1, t.data_ptr(), None, None, None, None) == -1
    assert l.vad_img_score(t.data_ptr(), 1, 32, 32, 256, t.data_ptr(), t.data_ptr(), 64, 1, t.data_ptr(), None, None, None, None) == -3


def _dec4_case(rng, n, h, w):
    x_in = np.maximum(rng.standard_normal((n, 32, h, w)), 0).astype(np.float32)            # dec3.3 output (post-ReLU)
    wt = (rng.standard_normal((32, 32, 2, 2)) * np.sqrt(1.0 / 32)).astype(np.float32)
    bt = (rng.standard_normal(32) * 0.1).astype(np.float32)
    bn = [a.astype(np.float32) for a in (rng.uniform(0.5, 1.5, 32), rng.standard_normal(32) * 0.1,
                                         rng.standard_normal(32) * 0.1, rng.uniform(0.25, 1.75, 32))]
    w3 = (rng.standard_normal((3, 32, 3, 3)) * np.sqrt(1.0 / (32 * 9))).astype(np.float32)
    b3 = (rng.standard_normal(3) * 0.1).astype(np.float32)
    frames = rng.uniform(-1, 1, (n, 3, 2 * h, 2 * w)).astype(np.float32)
    return x_in, wt, bt, bn, w3, b3, frames


def _dec4_run(H, case, fused, band=0, u8=None):
    """dec4 block + scoring through the C ABI: fused kernel, or the two launches it replaces.  -> (recon, errmap, scores)"""
    import ctypes as C
    x_in, wt, bt, bn, w3, b3, frames = case
    l = H.hip.lib()
    n, _, h, w = x_in.shape
    h2, w2 = 2 * h, 2 * w
    wtp, btp = H.pack_convt(wt, bt, bn)
    w3p = np.empty(l.vad_pack_conv3x3_to3_floats(32), np.float32)
    H.hip.check(l.vad_pack_conv3x3_to3(np.ascontiguousarray(w3).ctypes.data, 32, w3p.ctypes.data))
    w3d, b3d = H.dev(w3p), H.dev(b3)
    xin, xf = H.nhwc(x_in), H.dev(frames)
    recon = torch.full((n, 3, h2, w2), float("nan"), device="cuda")
    emap = torch.full((n, h2, w2), float("nan"), device="cuda")
    scores = torch.full((n,), float("nan"), device="cuda")
    if fused:
        nparts = l.vad_dec4_score_partials(h2, w2)
        parts = torch.full((n, nparts), float("nan"), device="cuda")
        l.vad_debug_set_dec4_band(band)
        try:
            H.hip.check(l.vad_dec4_score(xin.data_ptr(), wtp.data_ptr(), btp.data_ptr(), w3d.data_ptr() + 4 * 8 * 108, b3d.data_ptr(),
                                         xf.data_ptr(), parts.data_ptr(), recon.data_ptr(), emap.data_ptr(), n, h, w, H.stream()))
        finally:
            l.vad_debug_set_dec4_band(0)
    else:
        nparts = l.vad_score_partials(0, h2, w2)
        parts = torch.full((n, nparts), float("nan"), device="cuda")
        act = torch.full((n, h2, w2, 32), float("nan"), device="cuda")
        H.hip.check(l.vad_convt2x2(xin.data_ptr(), 0, wtp.data_ptr(), btp.data_ptr(), act.data_ptr(), 0, n, h, w, 32, 32, 2, 0, H.stream()))
        H.hip.check(l.vad_conv3x3_to3_score(act.data_ptr(), w3d.data_ptr(), b3d.data_ptr(), xf.data_ptr(), parts.data_ptr(),
                                            recon.data_ptr(), emap.data_ptr(), n, h2, w2, 32, H.stream()))
    H.hip.check(l.vad_score_finalize(parts.data_ptr(), nparts, n, h2, w2, scores.data_ptr(), None, 1, H.stream()))
    torch.cuda.synchronize()
    return recon.cpu().numpy(), emap.cpu().numpy(), scores.cpu().numpy()


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 8, 24), (1, 5, 72), (2, 33, 128), (1, 7, 136), (1, 4, 264), (2, 128, 128)])
def test_dec4_fused_kernel(n, h, w):
    """dec4.0 + dec4.3 + score in one kernel (csrc/dec4_fused.hip) against the CPU oracle's ConvTranspose2d + BatchNorm +
    ReLU + Conv2d + Tanh + squared error, against the two launches it replaces (same activations bit for bit; the 3x3 sum is
    ordered differently, so recon agrees to rounding), and against itself under every band height (torch.equal: the order of
    every sum depends on the pixel only).  Widths above 128 exercise the strips, 5 / 7 / 33 rows the ragged bands."""
    import hip_helpers as H
    rng = _rng(n * 1000 + h * 10 + w)
    case = _dec4_case(rng, n, h, w)
    x_in, wt, bt, bn, w3, b3, frames = case
    act = np.maximum(c_oracle.batchnorm_eval(c_oracle.convt2x2(x_in, wt, bt), *bn), 0)
    ref_recon = np.tanh(c_oracle.conv2d(act, w3, b3, 3).astype(np.float64))
    ref_emap = ((frames.astype(np.float64) - ref_recon) ** 2).mean(axis=1)
    ref_scores = ref_emap.mean(axis=(1, 2))
    recon, emap, scores = _dec4_run(H, case, fused=True)
    assert np.isfinite(recon).all() and np.isfinite(emap).all() and np.isfinite(scores).all()
    assert max_abs(recon, ref_recon) < ATOL and max_abs(emap, ref_emap) < ATOL
    assert np.max(np.abs(scores - ref_scores) / ref_scores) < 1e-5
    r2, e2, s2 = _dec4_run(H, case, fused=False)
    assert max_abs(recon, r2) < 2e-6 and np.max(np.abs(scores - s2) / s2) < 1e-6
    for band in (1, 2, 3, 16, 1000):
        rb, eb, sb = _dec4_run(H, case, fused=True, band=band)
        assert np.array_equal(recon, rb) and np.array_equal(emap, eb) and np.array_equal(scores, sb), band
