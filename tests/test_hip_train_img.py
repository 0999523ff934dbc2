"""GPU parity of the native training step of the image autoencoder (`ImageTrainer`, reference train.py:28-52 with the
criteria of train.py:149-158).  Same standard as tests/test_hip_train_step.py: the reference's own step (golden fixture),
float64 gradients with the kernels' branch decisions imposed (exact at any size).

Multi-step TRAJECTORIES are a different matter at train.py's learning rate (1e-3): Adam's first updates are +-lr for every
weight whatever the size of its gradient, so weights whose gradient is below fp32 noise get a random sign.  Measured on
the CPU alone, same code, latent 32, 3 frames of 32x32: fp32 vs float64 losses differ by 2e-3 at step 3 and 1e-2 by step
10-25, parameters by up to 0.2 lr on average per tensor after 3 steps; changing the CPU thread count moves the fp32 loss
by 2e-3 by step 10.  Trajectory comparisons between two fp32 implementations are therefore held to that spread, and the
tight statement about the update is made without the chaos: the parameters after each native step must equal
torch.optim.Adam applied to the kernels' OWN gradients (test_image_optimizer_applies_its_own_gradients_like_torch_adam)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_synthetic

pytestmark = pytest.mark.gpu

LR, WD = 1e-3, 1e-5        # train.py:256, 159


@pytest.fixture(autouse=True)
def _fixed_cpu_threads():
    before = torch.get_num_threads()
    torch.set_num_threads(4)
    yield
    torch.set_num_threads(before)


def _make(vad, latent, wseed):
    m = vad.ConvAutoencoder(in_channels=3, latent_dim=latent)
    load_synthetic(vad, m, wseed)
    return m


def _criterion(vad, loss, alpha, window, channels=3, double=False):
    if loss == "mse":
        return nn.MSELoss()
    crit = vad.SSIMLoss(window_size=window) if loss == "ssim" else vad.CombinedLoss(alpha=alpha, window_size=window)
    if double:
        ssim = crit if loss == "ssim" else crit.ssim
        ssim.window = vad.losses._gaussian_window(window, channels).double()
    return crit


def _bn_fed_biases(model):
    """conv / convT biases directly followed by BatchNorm: exactly-zero true gradient in train mode"""
    names = []
    for prefix, block in list(model.encoder.named_children()) + list(model.decoder.named_children()):
        mods = list(block)
        for i, m in enumerate(mods[:-1]):
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and isinstance(mods[i + 1], nn.BatchNorm2d):
                owner = "encoder" if prefix.startswith("enc") else "decoder"
                names.append(f"{owner}.{prefix}.{i}.bias")
    return set(names)


def _record_decisions(vad, tr, x):
    """Branch decisions of every BatchNorm backward, in the order train_step_img.hip issues them."""
    l = vad.hip.lib()
    n, _, h, w = x.shape
    c, d = [32, 64, 128, tr.latent], [128, 64, 32, 32]
    sizes = []
    for j in (3, 2, 1, 0):
        hj, wj = (h // 16) << (j + 1), (w // 16) << (j + 1)
        if j < 3:
            sizes.append((("dc", j), (n, hj, wj, d[j])))
        sizes.append((("dt", j), (n, hj, wj, d[j])))
    for i in (3, 2, 1, 0):
        sizes.append((("eb", i), (n, (h >> i) // 2, (w >> i) // 2, c[i])))
        sizes.append((("ea", i), (n, h >> i, w >> i, c[i])))
    total = sum(int(np.prod(s)) for _, s in sizes)
    buf = torch.zeros(total, dtype=torch.uint8, device="cuda")
    vad.hip.check(l.vad_debug_set_train_decisions(buf.data_ptr(), total))
    try:
        loss, _ = tr.forward_backward(x)
        torch.cuda.synchronize()
        assert l.vad_debug_train_decisions_used() == total
    finally:
        l.vad_debug_set_train_decisions(None, 0)
    out, off, host = {}, 0, buf.cpu()
    for key, shape in sizes:
        k = int(np.prod(shape))
        out[key] = host[off:off + k].view(*shape).permute(0, 3, 1, 2).contiguous()
        off += k
    return float(loss), out


def _conditioned_float64(vad, latent, wseed, x, decisions, loss, alpha, window):
    m = _make(vad, latent, wseed).double().train()
    cur = x.double()
    report = []
    for i in range(4):
        blk = list(getattr(m.encoder, f"enc{i + 1}"))          # conv, bn, lrelu, conv, bn, lrelu, pool
        v = blk[1](blk[0](cur))
        da = decisions[("ea", i)]
        pos = (da & 4) > 0
        with torch.no_grad():
            diff = (v > 0) != pos
            report.append((f"ea{i}", int(diff.sum()), float(v[diff].abs().max()) if diff.any() else 0.0, pos.numel()))
        cur = v * torch.where(pos, 1.0, 0.2).double()
        v = blk[4](blk[3](cur))
        nn_, c, hh, ww = v.shape
        win = v.view(nn_, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(nn_, c, hh // 2, ww // 2, 4)
        db = decisions[("eb", i)]
        am, pos = (db & 3).long(), (db & 4) > 0
        chosen = win.gather(-1, am.unsqueeze(-1)).squeeze(-1)
        with torch.no_grad():
            best, am64 = win.max(-1)
            diff_am = am64 != am
            diff_sign = (~diff_am) & ((chosen > 0) != pos)
            margins = torch.cat([(best - chosen)[diff_am].abs().reshape(-1), chosen[diff_sign].abs().reshape(-1)])
            report.append((f"eb{i}", int(diff_am.sum() + diff_sign.sum()), float(margins.max()) if margins.numel() else 0.0, db.numel()))
        cur = chosen * torch.where(pos, 1.0, 0.2).double()
    for j in range(4):
        blk = list(getattr(m.decoder, f"dec{j + 1}"))          # convT, bn, relu, conv, bn|tanh, ...
        v = blk[1](blk[0](cur))
        mask = (decisions[("dt", j)] & 4) > 0
        with torch.no_grad():
            diff = (v > 0) != mask
            report.append((f"dt{j}", int(diff.sum()), float(v[diff].abs().max()) if diff.any() else 0.0, mask.numel()))
        cur = v * mask.double()
        if j < 3:
            v = blk[4](blk[3](cur))
            mask = (decisions[("dc", j)] & 4) > 0
            with torch.no_grad():
                diff = (v > 0) != mask
                report.append((f"dc{j}", int(diff.sum()), float(v[diff].abs().max()) if diff.any() else 0.0, mask.numel()))
            cur = v * mask.double()
        else:
            cur = torch.tanh(blk[3](cur))
    val = _criterion(vad, loss, alpha, window, double=True)
    val = val.double() if isinstance(val, nn.Module) else val
    out = val(cur, x.double())
    out.backward()
    return float(out.detach()), {k: p.grad.detach().numpy() for k, p in m.named_parameters()}, report


@pytest.mark.parametrize("latent,n,hw,loss,alpha,window,wseed", [
    (32, 2, 32, "mse", 0.5, 11, 81), (64, 3, (48, 80), "combined", 0.3, 11, 82), (256, 1, 64, "ssim", 0.5, 11, 83),
    (32, 2, 16, "mse", 0.5, 11, 84), (32, 2, (32, 64), "combined", 0.5, 7, 85), (96, 2, 96, "mse", 0.5, 11, 86)])
@pytest.mark.parametrize("precision", ["fp32", "split", "winograd"])
def test_image_train_step_gradients_match_decision_conditioned_float64(vad, latent, n, hw, loss, alpha, window, wseed, precision):
    h, w = hw if isinstance(hw, tuple) else (hw, hw)
    x = torch.from_numpy(vad.synth.frames(wseed + 100, 0, n, 3, h, w))
    m = _make(vad, latent, wseed).cuda()
    tr = vad.ImageTrainer(m, lr=LR, weight_decay=WD, loss=loss, ssim_weight=alpha, window_size=window, precision=precision)
    loss_gpu, decisions = _record_decisions(vad, tr, x.cuda())
    got = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    loss64, want, report = _conditioned_float64(vad, latent, wseed, x, decisions, loss, alpha, window)
    for stage, ndiff, margin, total in report:
        assert ndiff <= max(3, total // 100000), f"{stage}: {ndiff} of {total} branch decisions differ from float64"
        assert margin < 2e-4, f"{stage}: a differing decision has margin {margin:.3e}"
    assert abs(loss_gpu - loss64) < 5e-6 * abs(loss64), (loss_gpu, loss64)
    bound = 2e-4         # every mode (measured 6e-6 .. 1.3e-5; split: since its backward runs on gradients scaled into the fp16 range - unscaled it needed 1e-3)
    zero_true, worst = _bn_fed_biases(m), 0.0
    assert len(zero_true) == 15
    for k, r in want.items():
        if k in zero_true:
            assert float(np.abs(got[k]).max()) == 0.0, k                     # written as exact zeros
            continue
        scale = max(float(np.abs(r).max()), 1e-12)
        err = float(np.abs(got[k] - r).max()) / scale
        worst = max(worst, err)
        assert err < bound, f"grad {k}: {err:.3e} of max |g| {scale:.3e} (differing decisions {[(s, c_) for s, c_, _, _ in report if c_]})"
    print(f"[{precision},{latent},{n},{hw},{loss}] worst gradient deviation {worst:.2e}; differing decisions {[(s, c_) for s, c_, _, _ in report if c_]}")


@pytest.mark.parametrize("tag", ["mse", "combined"])
def test_image_train_step_matches_reference_golden(vad, golden, tag):
    """The REFERENCE's ConvAutoencoder.train() + its own criterion classes + torch.optim.Adam(lr 1e-3, wd 1e-5), three steps
    on one seeded batch (tests/golden/make_golden.py:train_img_fixture)."""
    g = golden("train_img_l32.npz")
    latent, n, hw, wseed, xseed, steps, stride = (int(g[k]) for k in ("latent", "n", "hw", "wseed", "xseed", "steps", "stride"))
    x = torch.from_numpy(vad.synth.frames(xseed, 0, n, 3, hw, hw)).cuda()
    m = _make(vad, latent, wseed)
    init = {k: v.detach().clone().numpy() for k, v in m.state_dict().items()}
    m = m.cuda()
    tr = vad.ImageTrainer(m, lr=LR, weight_decay=WD, loss=tag, ssim_weight=0.5)
    loss0, _ = tr.forward_backward(x)
    keys = [str(k) for k in g["param_keys"]]
    got = {k: p.grad.detach().cpu().numpy().reshape(-1) for k, p in m.named_parameters()}
    assert list(got.keys()) == keys
    zero_true = _bn_fed_biases(m)
    for i, k in enumerate(keys):
        if k in zero_true:
            continue
        # 2e-3 on norms / 5e-2 on the few sampled elements, not the 3e-4 of a tie-free configuration: with the round-2 K order (channels
        # 0,4,2,6,1,5,3,7 per 8-group) ONE ReLU / pooling decision of this fixture falls on the other side of a rounding-level
        # tie than in the reference's summation order, and one flipped decision moves the small BatchNorm-affine gradients
        # upstream of it by ~4e-3 of their largest entry (module docstring; the decision-conditioned
        # float64 test above is the exact statement and holds at 2e-4)
        ref_n, ref_s = float(g[f"{tag}_grad_norms"][i]), g[f"{tag}_grad_{i}"]
        assert abs(float(np.linalg.norm(got[k].astype(np.float64))) - ref_n) < 2e-3 * ref_n + 1e-12, k
        assert np.abs(got[k][::stride] - ref_s).max() < 5e-2 * max(float(np.abs(ref_s).max()), 1e-12), k
    tr.optimizer_step()
    losses = [float(loss0)] + [float(tr.step(x)) for _ in range(steps - 1)]
    ref_losses = [float(r) for r in g[f"{tag}_losses"]]
    assert abs(losses[0] - ref_losses[0]) < 3e-5 * ref_losses[0]                      # before any update: pure forward
    for a, r in zip(losses[1:], ref_losses[1:]):
        assert abs(a - r) < 5e-3 * r, (losses, ref_losses)                               # intrinsic fp32 spread, see module docstring
    st = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for i, k in enumerate(str(k) for k in g["state_keys"]):
        ref = g[f"{tag}_state_{i}"]
        got_s = st[k].reshape(-1)[::stride] if st[k].ndim else st[k].reshape(1)
        if k.endswith("num_batches_tracked"):
            assert int(got_s[0]) == int(ref[0]) == int(init[k]) + steps
        elif "running_" in k:
            assert np.abs(got_s - ref).max() < 2e-2 * max(1.0, np.abs(ref).max()), k
        else:
            d = np.abs(got_s - ref)
            assert d.max() <= 2.05 * LR * steps and d.mean() < 0.5 * LR, f"{k}: max {d.max():.3e} mean {d.mean():.3e}"


def test_image_loss_curve_and_checkpoint_roundtrip(vad, tmp_path):
    """25 Adam steps with the combined criterion against CPU autograd: the loss falls and stays within the intrinsic fp32
    spread of this trajectory (module docstring); a checkpoint taken from the native trainer after 10 steps, loaded into
    the CPU module + torch.optim.Adam, reproduces the loss of step 10."""
    latent, n, hw, wseed, steps = 32, 4, 32, 87, 25
    x = torch.from_numpy(vad.synth.frames(wseed + 100, 0, n, 3, hw, hw))
    crit = vad.CombinedLoss(alpha=0.5)
    ref = _make(vad, latent, wseed).train()
    opt, want = torch.optim.Adam(ref.parameters(), lr=LR, weight_decay=WD), []
    for _ in range(steps):
        loss = crit(ref(x), x)
        opt.zero_grad()
        loss.backward()
        opt.step()
        want.append(float(loss.detach()))
    m = _make(vad, latent, wseed).cuda()
    tr = vad.ImageTrainer(m, lr=LR, weight_decay=WD, loss="combined", ssim_weight=0.5)
    xd, got = x.cuda(), []
    for s in range(steps):
        if s == 10:
            torch.save({"model_state_dict": m.state_dict(), "optimizer_state_dict": tr.state_dict()}, tmp_path / "ck.pth")
        got.append(float(tr.step(xd)))
    rel = [abs(a - r) / r for a, r in zip(got, want)]
    assert max(rel[:3]) < 5e-3 and max(rel) < 5e-2 and got[-1] < 0.6 * got[0], (max(rel), got[0], got[-1])
    ck = torch.load(tmp_path / "ck.pth", map_location="cpu", weights_only=True)
    res = _make(vad, latent, wseed)
    res.load_state_dict(ck["model_state_dict"])
    res.train()
    opt2 = torch.optim.Adam(res.parameters(), lr=LR, weight_decay=WD)
    opt2.load_state_dict(ck["optimizer_state_dict"])
    loss = crit(res(x), x)
    assert abs(float(loss.detach()) - got[10]) < 1e-4 * got[10]          # the resumed model reproduces step 10's loss
    # the trained weights are what eval-mode scoring uses afterwards
    m.eval()
    with torch.no_grad():
        s = m.get_reconstruction_error(xd)
    assert torch.isfinite(s).all()


@pytest.mark.parametrize("loss", ["mse", "combined"])
def test_image_optimizer_applies_its_own_gradients_like_torch_adam(vad, loss):
    """The update itself, without trajectory chaos: after every native step the parameters must equal what
    torch.optim.Adam(lr, weight_decay) makes of the SAME gradients (the kernels' own) on a CPU shadow copy.  Covers the flat
    parameter / gradient layout, bias corrections, weight decay and the step counter over four steps."""
    latent, n, hw, wseed = 64, 2, 32, 88
    x = torch.from_numpy(vad.synth.frames(wseed + 100, 0, n, 3, hw, hw)).cuda()
    m = _make(vad, latent, wseed).cuda()
    shadow = [p.detach().cpu().clone().requires_grad_(True) for p in m.parameters()]
    opt = torch.optim.Adam(shadow, lr=LR, weight_decay=WD)
    tr = vad.ImageTrainer(m, lr=LR, weight_decay=WD, loss=loss)
    for step in range(4):
        tr.forward_backward(x)
        for sp, p in zip(shadow, m.parameters()):
            sp.grad = p.grad.detach().cpu().clone()
        opt.step()
        tr.optimizer_step()
        worst = max(float((p.detach().cpu() - sp.detach()).abs().max()) for sp, p in zip(shadow, m.parameters()))
        assert worst < 3e-7, f"step {step}: parameters differ from torch.optim.Adam on the same gradients by {worst:.3e}"


def test_image_trainer_rejects_bad_arguments(vad):
    with pytest.raises(vad.hip.VadError, match="GPU"):
        vad.ImageTrainer(vad.ConvAutoencoder(latent_dim=32))
    with pytest.raises(vad.hip.VadError, match="loss must be"):
        vad.ImageTrainer(vad.ConvAutoencoder(latent_dim=32).cuda(), loss="l1")
    # "bf16" names the bf16-TENSOR mode of VideoTrainer everywhere: the image trainer has no such form and says so by name
    with pytest.raises(vad.hip.VadError, match="bf16_operands"):
        vad.ImageTrainer(vad.ConvAutoencoder(latent_dim=32).cuda(), precision="bf16")
    assert vad.ImageTrainer(vad.ConvAutoencoder(latent_dim=32).cuda(), precision="bf16_operands").precision == "bf16_operands"
    tr = vad.ImageTrainer(vad.ConvAutoencoder(latent_dim=32).cuda())
    with pytest.raises(vad.hip.VadError, match="multiples of 16"):
        tr.step(torch.zeros(2, 3, 24, 32, device="cuda"))
