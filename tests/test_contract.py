"""CPU-only checks of the drop-in boundary: module tree / state_dict contract against the reference's own
key list (captured in tests/golden), the C ABI surface, host-side weight packing, and the harness helpers."""
import ctypes as C
import importlib
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import IMG_GOLDENS, REPO, VID_GOLDENS, in_channels_of, load_synthetic, rel_err


def test_import_surface_matches_reference():
    """`from models import ConvAutoencoder`, `from models.video_autoencoder import VideoAutoencoder`,
    `from utils import CombinedLoss, SSIMLoss` (reference evaluate.py:22, evaluate_video.py:25, train.py:24)."""
    import models
    import models.video_autoencoder as mv
    import utils
    assert set(models.__all__) >= {"ConvAutoencoder", "Encoder", "Decoder"}
    for name in ("ConvLSTMCell", "ConvLSTM", "VideoEncoder", "VideoDecoder", "VideoAutoencoder"):
        assert hasattr(mv, name)
    assert utils.SSIMLoss and utils.CombinedLoss


def test_option_a_layout_imports(tmp_path):
    """INTEGRATION.md option A: the reference's tree with `models/` replaced, `utils/losses.py` replaced and the reference's
    own `utils/__init__.py` / `dataset.py` / `download_data.py` kept.  The kept files are stood in for by stubs written
    here (the reference never travels): an `__init__.py` with the reference's three import lines (utils/__init__.py:5-7)
    and modules defining the names it imports.  The scripts' import lines (evaluate.py:22-23, train.py:23-24,
    evaluate_video.py:25) must resolve, in this layout and with this repository's own `utils/__init__.py` instead."""
    import shutil
    import subprocess
    import sys
    for own_init in (False, True):
        root = tmp_path / ("own" if own_init else "kept")
        (root / "utils").mkdir(parents=True)
        shutil.copytree(REPO / "models", root / "models")
        (root / "video-anomaly-detection_amd").symlink_to(REPO / "video-anomaly-detection_amd")
        shutil.copy(REPO / "utils" / "losses.py", root / "utils" / "losses.py")
        (root / "utils" / "dataset.py").write_text("class MVTecDataset:\n    pass\n\ndef get_dataloaders(*a, **k):\n    return None\n")
        (root / "utils" / "download_data.py").write_text(
            "def create_synthetic_test_data(*a, **k):\n    return None\n\ndef download_with_kagglehub(*a, **k):\n    return None\n")
        if own_init:
            shutil.copy(REPO / "utils" / "__init__.py", root / "utils" / "__init__.py")
        else:
            (root / "utils" / "__init__.py").write_text(
                "from .dataset import MVTecDataset, get_dataloaders\n"
                "from .download_data import create_synthetic_test_data, download_with_kagglehub\n"
                "from .losses import SSIMLoss, CombinedLoss\n")
        code = ("from models import ConvAutoencoder\nfrom utils import MVTecDataset, CombinedLoss\n"
                "from utils import SSIMLoss, get_dataloaders, create_synthetic_test_data\n"
                "from models.video_autoencoder import VideoAutoencoder\nimport utils.losses\n"
                "assert ConvAutoencoder.__module__.startswith('video-anomaly-detection_amd')\n"
                "assert CombinedLoss.__module__.startswith('video-anomaly-detection_amd')\nprint('ok')\n")
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                           env={"PATH": "/usr/bin:/bin", "PYTHONPATH": str(root)})
        assert r.returncode == 0 and r.stdout.strip() == "ok", (own_init, r.stderr[-2000:])


def test_eval_with_autograd_enabled_warns_once(vad):
    """eval() + grad enabled leaves the HIP path for the torch composition: a RuntimeWarning says so, once per model."""
    import warnings
    m = vad.ConvAutoencoder(latent_dim=32).eval()
    x = torch.zeros(1, 3, 16, 16)
    with pytest.warns(RuntimeWarning, match="no_grad"):
        m.get_reconstruction_error(x)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m(x)                                  # second call: silent
        m.train()
        m(torch.zeros(2, 3, 16, 16))          # train mode is the documented autograd path: never warns
    v = vad.VideoAutoencoder(latent_dim=32, lstm_hidden_dim=32, lstm_num_layers=1).eval()
    with pytest.warns(RuntimeWarning, match="no_grad"):
        v(torch.zeros(1, 2, 3, 16, 16))


def test_state_dict_contract_image(vad, golden):
    g = golden("init.npz")
    m = vad.ConvAutoencoder()
    assert list(m.state_dict().keys()) == list(g["img_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["img_nparams"]) == 1_546_147
    for name in IMG_GOLDENS:
        f = golden(name)
        mm = vad.ConvAutoencoder(in_channels=in_channels_of(f), latent_dim=int(f["latent_dim"]))
        sd = mm.state_dict()
        assert list(sd.keys()) == list(f["keys"])
        assert [",".join(map(str, v.shape)) for v in sd.values()] == list(f["shapes"])
        load_synthetic(vad, mm, 1)   # strict load of a full state dict


def test_state_dict_contract_video(vad, golden):
    g = golden("init.npz")
    m = vad.VideoAutoencoder()
    assert list(m.state_dict().keys()) == list(g["vid_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["vid_nparams"]) == 2_709_411
    for name in VID_GOLDENS:
        f = golden(name)
        mm = vad.VideoAutoencoder(in_channels=in_channels_of(f), latent_dim=int(f["latent_dim"]), lstm_hidden_dim=int(f["hid"]),
                                  lstm_num_layers=int(f["layers"]))
        sd = mm.state_dict()
        assert list(sd.keys()) == list(f["keys"])
        assert [",".join(map(str, v.shape)) for v in sd.values()] == list(f["shapes"])
    assert isinstance(vad.VideoAutoencoder(latent_dim=128, lstm_hidden_dim=128).proj, torch.nn.Identity)
    assert isinstance(vad.VideoAutoencoder(latent_dim=32, lstm_hidden_dim=64).proj, torch.nn.Conv2d)


def test_init_follows_reference(vad):
    """Xavier-normal conv weights, zero biases, identity BN (reference models/autoencoder.py:170-179)."""
    torch.manual_seed(0)
    m = vad.ConvAutoencoder()
    w = m.encoder.enc2[3].weight
    assert abs(float(w.std()) - (2.0 / ((64 + 64) * 9)) ** 0.5) < 2e-3
    assert float(m.encoder.enc2[3].bias.abs().max()) == 0.0
    bn = m.decoder.dec1[1]
    assert torch.all(bn.weight == 1) and torch.all(bn.bias == 0)


def test_torch_training_path_matches_oracle_on_cpu(vad, golden):
    """train()/autograd keeps the stock torch.nn composition; in eval mode WITH grad enabled it is the same
    composition and must agree with the reference's golden scores (this is not the HIP path)."""
    g = golden("img_l32_32.npz")
    m = vad.ConvAutoencoder(latent_dim=32)
    load_synthetic(vad, m, int(g["wseed"]))
    m.eval()
    x = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, int(g["n"]), 3, 32, 32))
    s = m.get_reconstruction_error(x)           # grad enabled -> torch composition
    assert s.requires_grad and rel_err(s.detach().numpy(), g["scores"]) < 2e-6
    v = golden("vid_l3_32.npz")
    mv = vad.VideoAutoencoder(latent_dim=64, lstm_hidden_dim=64, lstm_num_layers=3)
    load_synthetic(vad, mv, int(v["wseed"]))
    mv.eval()
    xv = torch.from_numpy(vad.synth.clips(int(v["xseed"]), 0, int(v["b"]), int(v["t"]), 3, 32, 32))
    assert rel_err(mv.get_reconstruction_error(xv, per_frame=True).detach().numpy(), v["frame"]) < 2e-6


def test_inference_has_no_cpu_fallback(vad):
    m = vad.ConvAutoencoder(latent_dim=32).eval()
    with torch.no_grad(), pytest.raises(vad.hip.VadError, match="no CPU fallback"):
        m.get_reconstruction_error(torch.zeros(1, 3, 32, 32))
    v = vad.VideoAutoencoder(latent_dim=64, lstm_hidden_dim=64).eval()
    with torch.no_grad(), pytest.raises(vad.hip.VadError, match="no CPU fallback"):
        v(torch.zeros(1, 2, 3, 32, 32))


def test_input_contract_is_checked_before_anything_is_packed(vad):
    """Shape, dtype and the uint8 / in_channels combination are rejected at the top of every inference entry point
    (capture included) - before weights are packed or uploaded - and the message names only formats the model accepts."""
    m1 = vad.ConvAutoencoder(in_channels=1, latent_dim=24).eval()
    v2 = vad.VideoAutoencoder(in_channels=2, latent_dim=32, lstm_hidden_dim=40).eval()
    with torch.no_grad():
        with pytest.raises(vad.hip.VadError, match="uint8 input needs in_channels == 3") as e:
            m1.get_reconstruction_error(torch.zeros(2, 32, 32, 1, dtype=torch.uint8))
        assert "uint8 input [" not in str(e.value).split("expected", 1)[1]      # no uint8 form advertised for this model
        with pytest.raises(vad.hip.VadError, match="uint8 input needs in_channels == 3"):
            m1.capture(torch.zeros(2, 32, 32, 1, dtype=torch.uint8))
        with pytest.raises(vad.hip.VadError, match="uint8 frames needs in_channels == 3"):
            v2.score_windows(torch.zeros(9, 32, 32, 2, dtype=torch.uint8), sequence_length=4)
        with pytest.raises(vad.hip.VadError, match=r"expected float input \[B,T,2,H,W\], got"):
            v2.get_reconstruction_error(torch.zeros(1, 3, 3, 32, 32))
        with pytest.raises(vad.hip.VadError, match=r"expected float input \[B,3,H,W\] or uint8 input \[B,H,W,3\]"):
            vad.ConvAutoencoder(latent_dim=32).eval().get_reconstruction_error(torch.zeros(1, 1, 32, 32))
    assert m1._hip.packed is None and v2._hip.packed is None


def test_losses_match_reference(vad, golden):
    g = golden("losses.npz")
    x = torch.from_numpy(vad.synth.frames(77, 0, 2, 3, 32, 32))
    y = torch.from_numpy(vad.synth.frames(78, 0, 2, 3, 32, 32)) * 0.25 + x * 0.75
    assert abs(float(vad.SSIMLoss()(y, x)) - float(g["ssim"])) < 1e-6
    assert abs(float(vad.CombinedLoss(alpha=0.5)(y, x)) - float(g["combined"])) < 1e-6
    assert abs(float(vad.CombinedLoss(alpha=0.3, window_size=7)(y, x)) - float(g["combined_03"])) < 1e-6


# ------------------------------------------------------------------------------------------ C ABI
def _declared_symbols():
    text = (REPO / "include" / "vad_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vad_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(vad):
    """libvad_hip.so loads on a machine without a GPU and exports exactly the header's surface."""
    lib = vad.hip.lib()
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vad_hip.h but not exported"
        assert name in vad.hip.SIGNATURES, f"{name} has no ctypes signature in hip.py"
    assert set(vad.hip.SIGNATURES) == set(declared)
    assert lib.vad_abi_version() == vad.hip.ABI_VERSION == 3


def test_argument_errors_without_gpu(vad):
    lib = vad.hip.lib()
    assert lib.vad_img_packed_floats(3, 256) > 1_500_000
    assert lib.vad_img_packed_floats(1, 256) == 0          # the C ABI is 3-plane; the modules widen 1- / 2-channel models
    # any positive width is taken (models/autoencoder.py:161): slots are zero-padded to the kernels' channel tiling
    assert lib.vad_img_packed_floats(3, 100) == lib.vad_img_packed_floats(3, 128) > lib.vad_img_packed_floats(3, 96)
    assert lib.vad_img_packed_floats(3, 0) == 0 and lib.vad_img_packed_floats(3, vad.hip.MAX_WIDTH + 1) == 0
    assert lib.vad_vid_packed_floats(128, 128, 2) > 2_600_000
    assert lib.vad_vid_packed_floats(128, 100, 2) == lib.vad_vid_packed_floats(128, 128, 2) + lib.vad_pack_conv1x1_floats(128, 128) + 128
    assert lib.vad_vid_packed_floats(100, 100, 2) == lib.vad_vid_packed_floats(128, 128, 2)      # no proj: one common width
    assert lib.vad_vid_packed_floats(128, 0, 2) == 0 and lib.vad_vid_packed_floats(128, 128, 9) == 0
    assert lib.vad_img_workspace_bytes(16, 250, 256, 256) == 0
    assert lib.vad_img_workspace_bytes(16, 256, 256, 256) >= 2 * 16 * 256 * 256 * 32 * 4
    assert lib.vad_vid_nparams(2, 0) == 48 and lib.vad_vid_nparams(1, 1) == 48
    assert lib.vad_score_partials(0, 256, 256) == 64 and lib.vad_score_partials(1, 256, 256) == 256   # 32x32 tiles; 128 rows x 2 segments
    assert lib.vad_score_partials(7, 256, 256) < 0 and b"kind" in lib.vad_last_error()


def test_odd_widths_are_zero_padded_by_the_packer(vad):
    """latent_dim 40 (image): the blob equals the blob of the latent-64 model whose extra channels have zero weights, zero
    bias and identity-free BatchNorm (scale 0) - i.e. padding happens in the packer and nowhere else."""
    lib = vad.hip.lib()
    m = vad.ConvAutoencoder(latent_dim=40)
    st = load_synthetic(vad, m, 5)
    big = vad.ConvAutoencoder(latent_dim=64)
    sd = {k: torch.zeros_like(v) for k, v in big.state_dict().items()}
    for k, v in st.items():
        t = torch.from_numpy(np.asarray(v))
        if t.dim() == 0:
            sd[k] = t
            continue
        sd[k][tuple(slice(0, d) for d in t.shape)] = t
        if k.endswith("running_var"):
            sd[k][t.shape[0]:] = 1.0
    big.load_state_dict(sd)

    def blob(model, latent):
        params = [np.ascontiguousarray(v.detach().numpy(), dtype=np.float32) for k, v in model.state_dict().items()
                  if not k.endswith("num_batches_tracked")]
        out = np.empty(lib.vad_img_packed_floats(3, latent), np.float32)
        vad.hip.check(lib.vad_img_pack(vad.hip.pointer_array(params), len(params), 3, latent, 0, out.ctypes.data), "pack")
        return out
    a, b = blob(m, 40), blob(big, 64)
    assert a.shape == b.shape
    assert np.array_equal(a[4:].view(np.uint32), b[4:].view(np.uint32))        # header word 2 holds the real latent_dim
    assert a[:4].view(np.uint32)[2] == 40 and b[:4].view(np.uint32)[2] == 64


def test_bn_folding_and_packing_layout(vad):
    """vad_pack_conv3x3: out[tap][cin/8][cout][8] = w[co][ci][tap] * gamma/sqrt(var+eps); bias folded in fp64."""
    lib = vad.hip.lib()
    rng = np.random.default_rng(0)
    cout, cin = 32, 64
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    bn = [rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32),
          rng.standard_normal(cout).astype(np.float32), rng.uniform(0.2, 2.0, cout).astype(np.float32)]
    out = np.empty(lib.vad_pack_conv3x3_floats(cout, cin), np.float32)
    bo = np.empty(cout, np.float32)
    ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in bn])
    vad.hip.check(lib.vad_pack_conv3x3(w.ctypes.data, b.ctypes.data, ptrs, cout, cin, 0, out.ctypes.data, bo.ctypes.data))
    s = bn[0].astype(np.float64) / np.sqrt(bn[3].astype(np.float64) + 1e-5)
    ref = (w.astype(np.float64) * s[:, None, None, None]).reshape(cout, cin // 8, 8, 9).transpose(3, 1, 0, 2)
    assert np.array_equal(out.reshape(9, cin // 8, cout, 8), ref.astype(np.float32))
    assert np.array_equal(bo, ((b.astype(np.float64) - bn[2]) * s + bn[1]).astype(np.float32))
    # no BN: plain re-ordering
    vad.hip.check(lib.vad_pack_conv3x3(w.ctypes.data, b.ctypes.data, None, cout, cin, 0, out.ctypes.data, bo.ctypes.data))
    assert np.array_equal(out.reshape(9, cin // 8, cout, 8), w.reshape(cout, cin // 8, 8, 9).transpose(3, 1, 0, 2))
    assert np.array_equal(bo, b)
    # convT: [q][cin/8][cout][8] from IOHW
    wt = rng.standard_normal((64, 32, 2, 2)).astype(np.float32)
    ot = np.empty(lib.vad_pack_convt2x2_floats(64, 32), np.float32)
    bt = np.empty(32, np.float32)
    vad.hip.check(lib.vad_pack_convt2x2(wt.ctypes.data, b.ctypes.data, None, 64, 32, 0, ot.ctypes.data, bt.ctypes.data))
    assert np.array_equal(ot.reshape(4, 8, 32, 8), wt.reshape(8, 8, 32, 4).transpose(3, 0, 2, 1))


def test_winograd_packing_is_the_same_convolution(vad):
    """vad_pack_conv3x3_wino (no GPU needed): the packed U = G g G^T, laid out [16][cin/8][cout][8] with BatchNorm folded, is the
    filter transform of Winograd F(2x2,3x3).  Pinned two ways: against numpy's G g G^T, and end to end - a numpy restatement of
    the kernel's arithmetic (input transform B^T d B, 16 element-wise products summed over channels, output transform A^T m A)
    on the packed operands reproduces the C oracle's direct convolution + folded BatchNorm on ragged sizes."""
    from oracle import c_oracle
    lib = vad.hip.lib()
    rng = np.random.default_rng(5)
    cout, cin, n, h, w_ = 32, 16, 2, 6, 10
    wt = (rng.standard_normal((cout, cin, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    bn = [rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32),
          rng.standard_normal(cout).astype(np.float32), rng.uniform(0.2, 2.0, cout).astype(np.float32)]
    out = np.empty(lib.vad_pack_conv3x3_wino_floats(cout, cin), np.float32)
    bo = np.empty(cout, np.float32)
    ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in bn])
    vad.hip.check(lib.vad_pack_conv3x3_wino(wt.ctypes.data, b.ctypes.data, ptrs, cout, cin, out.ctypes.data, bo.ctypes.data))
    assert out.size == 16 * cin * cout
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    s = bn[0].astype(np.float64) / np.sqrt(bn[3].astype(np.float64) + 1e-5)
    U = np.einsum("ra,oiab,cb->rcoi", G, wt.astype(np.float64) * s[:, None, None, None], G)          # [fr][fc][co][ci]
    got = out.reshape(16, cin // 8, cout, 8).transpose(0, 2, 1, 3).reshape(4, 4, cout, cin)
    assert np.allclose(got, U, rtol=0, atol=1e-7) and np.array_equal(bo, ((b.astype(np.float64) - bn[2]) * s + bn[1]).astype(np.float32))
    x = rng.standard_normal((n, cin, h, w_)).astype(np.float32)
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    xp = np.zeros((n, cin, h + 2 + 1, w_ + 2 + 1))                    # zero padding (+1: partial last tiles on odd sizes)
    xp[:, :, 1:h + 1, 1:w_ + 1] = x
    y = np.zeros((n, cout, h + 1, w_ + 1))
    for ty in range(0, h, 2):
        for tx in range(0, w_, 2):
            d = xp[:, :, ty:ty + 4, tx:tx + 4]
            V = np.einsum("ri,ncij,sj->nrsc", BT, d, BT)
            M = np.einsum("nrsc,rsoc->nrso", V, got.astype(np.float64))
            y[:, :, ty:ty + 2, tx:tx + 2] = np.einsum("ir,nrso,js->noij", AT, M, AT)
    y = y[:, :, :h, :w_] + bo[None, :, None, None]
    ref = c_oracle.batchnorm_eval(c_oracle.conv2d(x, wt, b, 3), *bn)
    assert np.abs(y - ref).max() < 2e-5
    assert lib.vad_pack_conv3x3(wt.ctypes.data, b.ctypes.data, None, cout, cin, 4, out.ctypes.data, bo.ctypes.data) == -1   # the direct packer refuses the mode
    assert b"vad_pack_conv3x3_wino" in lib.vad_last_error()


def test_model_pack_consumes_state_dict_in_order(vad):
    lib = vad.hip.lib()
    m = vad.ConvAutoencoder(latent_dim=64)
    load_synthetic(vad, m, 3)
    params = importlib.import_module("video-anomaly-detection_amd.autoencoder")._HipScorer.float_params(m)
    assert len(params) == 92
    blob = np.full(lib.vad_img_packed_floats(3, 64), np.nan, np.float32)
    vad.hip.check(lib.vad_img_pack(vad.hip.pointer_array(params), 92, 3, 64, 0, blob.ctypes.data))
    assert np.isfinite(blob).all()
    # the blob header names the arithmetic mode it was packed for (ABI 2: precision is an argument, never process state)
    assert lib.vad_blob_precision(blob.ctypes.data) == 0
    split = np.empty_like(blob)
    vad.hip.check(lib.vad_img_pack(vad.hip.pointer_array(params), 92, 3, 64, 1, split.ctypes.data))
    assert lib.vad_blob_precision(split.ctypes.data) == 1 and not np.array_equal(split[4:], blob[4:])
    wino = np.empty_like(blob)                    # VAD_PREC_WINO: same layout (a 3x3 slot holds either form), its own tag
    vad.hip.check(lib.vad_img_pack(vad.hip.pointer_array(params), 92, 3, 64, 4, wino.ctypes.data))
    assert lib.vad_blob_precision(wino.ctypes.data) == 4 and np.isfinite(wino).all() and not np.array_equal(wino[4:], blob[4:])
    assert lib.vad_blob_precision(blob[8:].ctypes.data) == -1 and b"not a packed model blob" in lib.vad_last_error()
    assert lib.vad_img_pack(vad.hip.pointer_array(params), 92, 3, 64, 7, blob.ctypes.data) == -1 and b"precision" in lib.vad_last_error()
    assert not hasattr(lib, "vad_set_precision")
    assert lib.vad_img_pack(vad.hip.pointer_array(params[:-1]), 91, 3, 64, 0, blob.ctypes.data) == -1
    v = vad.VideoAutoencoder(latent_dim=32, lstm_hidden_dim=64, lstm_num_layers=1)
    load_synthetic(vad, v, 4)
    vp = importlib.import_module("video-anomaly-detection_amd.autoencoder")._HipScorer.float_params(v)
    assert len(vp) == lib.vad_vid_nparams(1, 1)
    vb = np.full(lib.vad_vid_packed_floats(32, 64, 1), np.nan, np.float32)
    vad.hip.check(lib.vad_vid_pack(vad.hip.pointer_array(vp), len(vp), 32, 64, 1, 0, vb.ctypes.data))
    assert np.isfinite(vb).all()


# ------------------------------------------------------------------------------------------ harness
def test_synth_is_counter_based(vad):
    a = vad.synth.frames(5, 10, 4, 3, 16, 16)
    b = vad.synth.frames(5, 12, 2, 3, 16, 16)
    assert np.array_equal(a[2:], b) and a.dtype == np.float32 and a.min() >= -1 and a.max() <= 1
    u8 = vad.synth.frames_u8(0xC0FFEE, 0, 1, 3, 4, 4)
    assert u8.dtype == np.uint8 and u8[0, 0, 0, :4].tolist() == vad.synth.frames_u8(0xC0FFEE, 0, 1, 3, 4, 4)[0, 0, 0, :4].tolist()
    assert np.array_equal(vad.synth.clips(9, 1, 2, 3, 3, 8, 8).reshape(6, 3, 8, 8), vad.synth.frames(9, 3, 6, 3, 8, 8))
    lab = vad.synth.frame_label(1, np.arange(1000))
    assert 400 < lab.sum() < 600
    an = vad.synth.frames_u8(1, 0, 8, 3, 64, 64, anomalies=True)
    for i in range(8):
        assert ((an[i] == 255).all(axis=0).sum() >= 32 * 32) == bool(lab[i])


def test_roc_auc_matches_sklearn(vad):
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(1)
    for _ in range(5):
        y = rng.integers(0, 2, 200)
        s = np.round(rng.standard_normal(200) + y * 0.7, 1)       # rounding creates ties
        assert abs(vad.scoring.roc_auc(y, s) - roc_auc_score(y, s)) < 1e-12
    with pytest.raises(ValueError):
        vad.scoring.roc_auc([1, 1], [0.1, 0.2])


def test_compute_auroc_contract(vad):
    """Same call shape and result tuple as the reference's evaluate.compute_auroc (evaluate.py:46-91)."""
    class Fake:
        def get_reconstruction_error(self, images, per_pixel=False):
            return images.mean(dim=(1, 2, 3))
    batches = [{"image": torch.full((4, 3, 2, 2), float(i)) + torch.arange(4).view(4, 1, 1, 1), "label": np.array([0, 0, 1, 1]),
                "defect_type": ["good", "good", "scratch", "spot"]} for i in range(2)]
    auroc, labels, scores, per = vad.scoring.compute_auroc(Fake(), batches, "cpu")
    assert labels.tolist() == [0, 0, 1, 1] * 2 and scores.shape == (8,)
    assert set(per) == {"good", "scratch", "spot"} and per["good"]["count"] == 4 and per["spot"]["is_anomaly"] == 1
    assert 0.5 < auroc <= 1.0


def test_block_partition(vad):
    bp = vad.scoring.block_partition
    assert bp(100_000, 8, 0) == (0, 12_500, 12_500) and bp(100_000, 8, 7) == (87_500, 12_500, 12_500)
    assert bp(10, 4, 3) == (9, 1, 3) and bp(10, 4, 2) == (6, 3, 3) and bp(2, 4, 3) == (2, 0, 1)
    for n, w in [(1, 1), (7, 3), (64, 8), (65, 8), (3, 8)]:
        got = [i for r in range(w) for i in range(bp(n, w, r)[0], bp(n, w, r)[0] + bp(n, w, r)[1])]
        assert got == list(range(n))
