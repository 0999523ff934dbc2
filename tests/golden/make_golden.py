"""Generate tests/golden/*.npz by running the REFERENCE's own model code (build container only).

The reference (/root/reference) never travels to the GPU box, so its outputs are captured here as
small data fixtures: seeded inputs (regenerable from video-anomaly-detection_amd/synth.py), the
deterministic synthetic state dict (same generator, so weights are not stored) and the reference's
outputs.  The reference modules are loaded by file path under private names, because this repository
has its own drop-in `models` package.

    python tests/golden/make_golden.py                     # rewrites every fixture
    python tests/golden/make_golden.py clip_loop_64.npz    # rewrites the named ones
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))
synth = importlib.import_module("video-anomaly-detection_amd.synth")


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


ref_ae = _load("_ref_autoencoder", REF / "models" / "autoencoder.py")
ref_vae = _load("_ref_video_autoencoder", REF / "models" / "video_autoencoder.py")
ref_losses = _load("_ref_losses", REF / "utils" / "losses.py")


def _shapes(model) -> dict:
    return {k: tuple(v.shape) for k, v in model.state_dict().items()}


def _load_synth(model, seed: int):
    st = synth.synthetic_state(_shapes(model), seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=True)
    model.eval()
    return model


def _contract(model) -> dict:
    sh = _shapes(model)
    return {"keys": np.array(list(sh.keys())), "shapes": np.array([",".join(map(str, s)) for s in sh.values()]),
            "nparams": np.array(sum(p.numel() for p in model.parameters()))}


def _save(name: str, **arrays):
    path = HERE / name
    np.savez_compressed(path, **arrays)
    print(f"{name}: {path.stat().st_size / 1024:.0f} KiB")


def image_fixture(name, latent, wseed, xseed, n, hw, intermediates=False, subsample=None, in_ch=3):
    torch.manual_seed(0)
    model = _load_synth(ref_ae.ConvAutoencoder(in_channels=in_ch, latent_dim=latent), wseed)
    x = torch.from_numpy(synth.frames(xseed, 0, n, in_ch, hw, hw))
    out = {}
    hooks = []
    if intermediates:
        for blk in ["encoder.enc1", "encoder.enc2", "encoder.enc3", "encoder.enc4",
                    "decoder.dec1", "decoder.dec2", "decoder.dec3"]:
            mod = model.get_submodule(blk)
            hooks.append(mod.register_forward_hook(
                lambda m, i, o, blk=blk: out.__setitem__("act." + blk, o.detach().numpy().copy())))
    with torch.no_grad():
        recon = model(x)
        for h in hooks:
            h.remove()
        emap = model.get_reconstruction_error(x, per_pixel=True)
        scores = model.get_reconstruction_error(x, per_pixel=False)
        latent_t = model.get_latent(x)
    if subsample:
        out["recon_sub"] = recon.numpy()[:, :, ::subsample, ::subsample]
        out["errmap_sub"] = emap.numpy()[:, :, ::subsample, ::subsample]
    else:
        out["recon"] = recon.numpy()
        out["errmap"] = emap.numpy()
    out["latent"] = latent_t.numpy()
    out["scores"] = scores.numpy()
    _save(name, latent_dim=np.array(latent), in_channels=np.array(in_ch), wseed=np.array(wseed), xseed=np.array(xseed),
          n=np.array(n), hw=np.array(hw), **_contract(model), **out)


def video_fixture(name, latent, hid, layers, wseed, xseed, b, t, hw, in_ch=3):
    torch.manual_seed(0)
    model = _load_synth(ref_vae.VideoAutoencoder(in_channels=in_ch, latent_dim=latent, lstm_hidden_dim=hid,
                                                 lstm_num_layers=layers), wseed)
    x = torch.from_numpy(synth.clips(xseed, 0, b, t, in_ch, hw, hw))
    with torch.no_grad():
        recon = model(x)
        seq = model.get_reconstruction_error(x)
        frame = model.get_reconstruction_error(x, per_frame=True)
        emap = model.get_reconstruction_error(x, per_pixel=True)
        both = model.get_reconstruction_error(x, per_frame=True, per_pixel=True)
    assert both.shape == emap.shape   # per_pixel wins when both flags are set
    _save(name, latent_dim=np.array(latent), hid=np.array(hid), layers=np.array(layers), in_channels=np.array(in_ch),
          wseed=np.array(wseed), xseed=np.array(xseed), b=np.array(b), t=np.array(t), hw=np.array(hw), **_contract(model),
          recon=recon.numpy(), seq=seq.numpy(), frame=frame.numpy(), errmap=emap.numpy())


def convlstm_fixture(name):
    torch.manual_seed(0)
    cell = ref_vae.ConvLSTMCell(input_dim=32, hidden_dim=64, kernel_size=3)
    st = synth.synthetic_state({k: tuple(v.shape) for k, v in cell.state_dict().items()}, 31)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 32, 8, 8), dtype=np.float32)
    h = rng.standard_normal((2, 64, 8, 8), dtype=np.float32) * 0.5
    c = rng.standard_normal((2, 64, 8, 8), dtype=np.float32)
    with torch.no_grad():
        h1, c1 = cell(torch.from_numpy(x), (torch.from_numpy(h), torch.from_numpy(c)))
    stack = ref_vae.ConvLSTM(input_dim=32, hidden_dims=[64, 64], kernel_size=3, num_layers=2)
    st2 = synth.synthetic_state({k: tuple(v.shape) for k, v in stack.state_dict().items()}, 32)
    stack.load_state_dict({k: torch.from_numpy(v) for k, v in st2.items()})
    xs = rng.standard_normal((2, 3, 32, 8, 8), dtype=np.float32)
    with torch.no_grad():
        out, (hl, cl) = stack(torch.from_numpy(xs))
    _save(name, x=x, h=h, c=c, h1=h1.numpy(), c1=c1.numpy(), xs=xs, seq_out=out.numpy(), h_last=hl.numpy(),
          c_last=cl.numpy())


def auroc_fixture(name, patch=10):
    """configs[0]: 64 synthetic 256x256 frames in batches of 16 through the reference model, AUROC by the
    same sklearn call evaluate.compute_auroc makes (reference evaluate.py:56-74).  Labelled frames carry a saturated
    `patch` x `patch` square: at 10 pixels it moves a score by about one standard deviation of the normal frames'
    scores, so the AUROC lands well inside (0.5, 1) and a wrong ranking changes it (a 32-pixel patch gave 1.0)."""
    from sklearn.metrics import roc_auc_score
    seed = 0xC0FFEE
    torch.manual_seed(0)
    model = _load_synth(ref_ae.ConvAutoencoder(in_channels=3, latent_dim=256), 7)
    labels = synth.frame_label(seed, np.arange(64))
    scores = []
    with torch.no_grad():
        for s in range(0, 64, 16):   # batch_size=16: reference evaluate.py:240
            x = torch.from_numpy(synth.frames(seed, s, 16, 3, 256, 256, anomalies=patch))
            scores.extend(model.get_reconstruction_error(x, per_pixel=False).numpy())
    scores = np.array(scores, dtype=np.float32)
    auroc = roc_auc_score(labels, scores)
    print(f"  configs[0] AUROC {auroc:.4f} (patch {patch})")
    assert 0.6 < auroc < 0.95, "the AUROC fixture must not be degenerate"
    _save(name, seed=np.array(seed), wseed=np.array(7), patch=np.array(patch), labels=labels, scores=scores,
          auroc=np.array(auroc))


def clip_loop_fixture(name, n=10, t=6, hw=64, batch=4, wseed=24, xseed=204):
    """Row a12: the clip loop of the reference's evaluate_video.evaluate (evaluate_video.py:137-154) over `n` seeded
    clips in batches of `batch` (its --batch_size default, evaluate_video.py:416; the last batch is ragged): per batch
    `get_reconstruction_error(sequences, per_frame=False)` extended into the clip scores and, because the batch carries
    frame labels, a second forward `per_frame=True` extended into the frame scores.  Default model dimensions."""
    torch.manual_seed(0)
    model = _load_synth(ref_vae.VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128,
                                                 lstm_num_layers=2), wseed)
    labels = synth.frame_label(xseed, np.arange(n))
    all_scores, all_frame_scores = [], []
    with torch.no_grad():
        for s in range(0, n, batch):
            k = min(batch, n - s)
            sequences = torch.from_numpy(synth.clips(xseed, s, k, t, 3, hw, hw))
            seq_errors = model.get_reconstruction_error(sequences, per_frame=False)
            all_scores.extend(seq_errors.cpu().numpy())
            frame_errors = model.get_reconstruction_error(sequences, per_frame=True)
            for frame_err in frame_errors.cpu().numpy():
                all_frame_scores.extend(frame_err)
    _save(name, n=np.array(n), t=np.array(t), hw=np.array(hw), batch=np.array(batch), wseed=np.array(wseed),
          xseed=np.array(xseed), labels=labels, seq_scores=np.array(all_scores, dtype=np.float32),
          frame_scores=np.array(all_frame_scores, dtype=np.float32).reshape(n, t))


def losses_fixture(name):
    x = torch.from_numpy(synth.frames(77, 0, 2, 3, 32, 32))
    y = torch.from_numpy(synth.frames(78, 0, 2, 3, 32, 32)) * 0.25 + x * 0.75
    _save(name, ssim=ref_losses.SSIMLoss()(y, x).numpy(), combined=ref_losses.CombinedLoss(alpha=0.5)(y, x).numpy(),
          combined_03=ref_losses.CombinedLoss(alpha=0.3, window_size=7)(y, x).numpy())


def init_fixture(name):
    """Parameter counts and init statistics of the reference constructors."""
    torch.manual_seed(1234)
    a = ref_ae.ConvAutoencoder()
    v = ref_vae.VideoAutoencoder()
    _save(name, img_nparams=np.array(sum(p.numel() for p in a.parameters())),
          vid_nparams=np.array(sum(p.numel() for p in v.parameters())),
          img_keys=np.array(list(a.state_dict().keys())), vid_keys=np.array(list(v.state_dict().keys())),
          img_w_std=np.array([float(a.state_dict()[k].std()) for k in a.state_dict() if k.endswith("0.weight") and a.state_dict()[k].dim() == 4]))





def train_fixture(name, latent=32, layers=2, b=2, t=3, hw=32, wseed=61, xseed=161, steps=3, stride=7):
    """Row f-1: the reference's own training step (train_video.py:44-65,175): VideoAutoencoder.train(), nn.MSELoss,
    torch.optim.Adam(lr 1e-4, weight_decay 1e-5), `steps` steps on one seeded batch.  Stored: the losses, the norm and
    a strided sample of every first-step gradient, a strided sample of the state dict after the last step."""
    m = ref_vae.VideoAutoencoder(in_channels=3, latent_dim=latent, lstm_hidden_dim=latent, lstm_num_layers=layers)
    _load_synth(m, wseed)
    m.train()
    x = torch.from_numpy(synth.clips(xseed, 0, b, t, 3, hw, hw))
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)
    crit = torch.nn.MSELoss()
    arrays, losses = {}, []
    for s in range(steps):
        loss = crit(m(x), x)
        opt.zero_grad()
        loss.backward()
        if s == 0:
            keys = [k for k, _ in m.named_parameters()]
            arrays["param_keys"] = np.array(keys)
            arrays["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
            for i, (_, p) in enumerate(m.named_parameters()):
                arrays[f"grad_{i}"] = p.grad.detach().reshape(-1)[::stride].numpy().copy()
        opt.step()
        losses.append(float(loss.detach()))
    st = m.state_dict()
    arrays["state_keys"] = np.array(list(st.keys()))
    for i, (k, v) in enumerate(st.items()):
        arrays[f"state_{i}"] = (v.detach().reshape(-1)[::stride] if v.dim() else v.detach().reshape(1)).numpy().copy()
    _save(name, losses=np.array(losses), latent=np.array(latent), layers=np.array(layers), b=np.array(b), t=np.array(t),
          hw=np.array(hw), wseed=np.array(wseed), xseed=np.array(xseed), steps=np.array(steps), stride=np.array(stride), **arrays)


def train_img_fixture(name, latent=32, n=3, hw=32, wseed=71, xseed=171, steps=3, stride=23):
    """The reference's image training step (train.py:28-52,149-159): ConvAutoencoder.train(), nn.MSELoss (the default
    criterion) and CombinedLoss(alpha=0.5) from the reference's utils/losses.py, torch.optim.Adam(lr 1e-3, weight_decay 1e-5),
    `steps` steps on one seeded batch.  Stored per criterion: losses, norms + strided samples of the first-step gradients,
    strided samples of the final state dict."""
    arrays = {}
    x = torch.from_numpy(synth.frames(xseed, 0, n, 3, hw, hw))
    for tag, crit in (("mse", torch.nn.MSELoss()), ("combined", ref_losses.CombinedLoss(alpha=0.5))):
        m = ref_ae.ConvAutoencoder(in_channels=3, latent_dim=latent)
        _load_synth(m, wseed)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        losses = []
        for s in range(steps):
            loss = crit(m(x), x)
            opt.zero_grad()
            loss.backward()
            if s == 0:
                arrays["param_keys"] = np.array([k for k, _ in m.named_parameters()])
                arrays[f"{tag}_grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
                for i, (_, p) in enumerate(m.named_parameters()):
                    arrays[f"{tag}_grad_{i}"] = p.grad.detach().reshape(-1)[::stride].numpy().copy()
            opt.step()
            losses.append(float(loss.detach()))
        arrays[f"{tag}_losses"] = np.array(losses)
        st = m.state_dict()
        arrays["state_keys"] = np.array(list(st.keys()))
        for i, (k, v) in enumerate(st.items()):
            arrays[f"{tag}_state_{i}"] = (v.detach().reshape(-1)[::stride] if v.dim() else v.detach().reshape(1)).numpy().copy()
    _save(name, latent=np.array(latent), n=np.array(n), hw=np.array(hw), wseed=np.array(wseed), xseed=np.array(xseed),
          steps=np.array(steps), stride=np.array(stride), **arrays)


def trained_fixture(name, latent=64, epochs=12):
    """Precision gate of SURVEY.md section 8(d): a TRAINED, low-residual model.  Recipe: the reference's own synthetic
    dataset generator (utils/download_data.py:85-184, loaded by file path; numpy + PIL only), its model class, Adam(lr
    1e-3, wd 1e-5) + MSELoss as in train.py:149-159, batch 10, `epochs` epochs on the 50 training images.  Stored:
    the trained state dict (latent_dim 64 keeps it at ~2.7 MB), the 30 test images as uint8 and the reference's
    scores / AUROC on them."""
    import tempfile
    from PIL import Image
    from sklearn.metrics import roc_auc_score
    dl = _load("_ref_download_data", REF / "utils" / "download_data.py")
    with tempfile.TemporaryDirectory() as tmp:
        root = dl.create_synthetic_test_data(tmp, "synthetic")

        def load_dir(d):
            files = sorted(Path(d).glob("*.png"))
            return np.stack([np.asarray(Image.open(f).convert("RGB").resize((256, 256))) for f in files])
        train_u8 = load_dir(root / "train" / "good")
        good_u8 = load_dir(root / "test" / "good")
        bad_u8 = load_dir(root / "test" / "defect")

    def to_t(u8):   # ToTensor + Normalize(0.5, 0.5) (reference utils/dataset.py:65-70)
        return torch.from_numpy(synth.u8_to_unit(u8.transpose(0, 3, 1, 2)))
    torch.manual_seed(0)
    model = ref_ae.ConvAutoencoder(in_channels=3, latent_dim=latent)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    crit = torch.nn.MSELoss()
    xtr = to_t(train_u8)
    model.train()
    for ep in range(epochs):
        perm = torch.randperm(len(xtr))
        tot = 0.0
        for i in range(0, len(xtr), 10):
            xb = xtr[perm[i:i + 10]]
            opt.zero_grad()
            loss = crit(model(xb), xb)
            loss.backward()
            opt.step()
            tot += float(loss) * len(xb)
        print(f"  epoch {ep + 1}: loss {tot / len(xtr):.6f}")
    model.eval()
    test_u8 = np.concatenate([good_u8, bad_u8])
    labels = np.array([0] * len(good_u8) + [1] * len(bad_u8))
    with torch.no_grad():
        xt = to_t(test_u8)
        scores = torch.cat([model.get_reconstruction_error(xt[i:i + 16]) for i in range(0, len(xt), 16)]).numpy()
        emap0 = model.get_reconstruction_error(xt[:1], per_pixel=True).numpy()
    st = {("w." + k): v.numpy() for k, v in model.state_dict().items()}
    print(f"  scores good {scores[labels == 0].mean():.5f} defect {scores[labels == 1].mean():.5f} "
          f"auroc {roc_auc_score(labels, scores):.4f}")
    _save(name, latent_dim=np.array(latent), test_u8=test_u8, labels=labels, scores=scores.astype(np.float32),
          auroc=np.array(roc_auc_score(labels, scores)), errmap0_sub=emap0[:, :, ::8, ::8], **st)


FIXTURES = {
    "img_l32_32.npz": lambda n: image_fixture(n, latent=32, wseed=11, xseed=101, n=3, hw=32, intermediates=True),
    "img_l256_64.npz": lambda n: image_fixture(n, latent=256, wseed=12, xseed=102, n=2, hw=64),
    "img_l256_256.npz": lambda n: image_fixture(n, latent=256, wseed=13, xseed=103, n=1, hw=256, subsample=8),
    # constructor arguments the kernels' channel tiling does not divide (zero-padded by the packers) and fewer than 3 planes:
    # the reference takes any positive width (models/autoencoder.py:161, models/video_autoencoder.py:290-296)
    "img_l100_32.npz": lambda n: image_fixture(n, latent=100, wseed=14, xseed=104, n=2, hw=32),
    "img_c1_l24_32.npz": lambda n: image_fixture(n, latent=24, wseed=15, xseed=105, n=2, hw=32, in_ch=1),
    # more than 3 planes (round 4: the library's generic first / last layers, csrc/wide_io.hip)
    "img_c5_l32_32.npz": lambda n: image_fixture(n, latent=32, wseed=16, xseed=106, n=3, hw=32, in_ch=5),
    "vid_c4_l32_32.npz": lambda n: video_fixture(n, latent=32, hid=32, layers=2, wseed=28, xseed=208, b=2, t=3, hw=32, in_ch=4),
    "vid_l48_h96_32.npz": lambda n: video_fixture(n, latent=48, hid=96, layers=2, wseed=25, xseed=205, b=2, t=3, hw=32),
    "vid_l100_32.npz": lambda n: video_fixture(n, latent=100, hid=100, layers=1, wseed=26, xseed=206, b=1, t=3, hw=32),
    "vid_c2_l32_h40_32.npz": lambda n: video_fixture(n, latent=32, hid=40, layers=1, wseed=27, xseed=207, b=1, t=2, hw=32, in_ch=2),
    "vid_default_64.npz": lambda n: video_fixture(n, latent=128, hid=128, layers=2, wseed=21, xseed=201, b=2, t=3, hw=64),
    "vid_proj_32.npz": lambda n: video_fixture(n, latent=32, hid=64, layers=1, wseed=22, xseed=202, b=1, t=4, hw=32),
    "vid_l3_32.npz": lambda n: video_fixture(n, latent=64, hid=64, layers=3, wseed=23, xseed=203, b=2, t=2, hw=32),
    "convlstm_unit.npz": convlstm_fixture,
    "auroc_cfg0.npz": auroc_fixture,
    "clip_loop_64.npz": clip_loop_fixture,
    "losses.npz": losses_fixture,
    "init.npz": init_fixture,
    "img_trained_l64.npz": trained_fixture,
    "train_vid_l32.npz": train_fixture,
    "train_img_l32.npz": train_img_fixture,
}

if __name__ == "__main__":
    torch.set_num_threads(8)
    for fixture in (sys.argv[1:] or list(FIXTURES)):     # optional arguments: the fixtures to rewrite (default: all)
        FIXTURES[fixture](fixture)
