"""Seeded random sweep of the scoring entry points against the CPU oracle: frame sizes (multiples of 16, non-square),
batch sizes that do not divide the chunk, chunk sizes, latent / hidden sizes, number of ConvLSTM layers, clip lengths,
the three arithmetic modes (exact fp32, split-fp16, Winograd), in_channels other than 3 and uint8 ingest.  Every case is an exact statement of the drop-in contract: scores within 1e-5
relative of the reference arithmetic, identical under re-chunking."""
import os

import numpy as np
import pytest
import torch

from conftest import load_synthetic, rel_err
from oracle import torch_oracle

pytestmark = pytest.mark.gpu
SEEDS = range(int(os.environ.get("VAD_FUZZ_SEEDS", "14")))     # VAD_FUZZ_SEEDS=100 for a deeper ad-hoc sweep


def _state(st):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}


@pytest.mark.parametrize("seed", SEEDS)
def test_image_random_configuration(vad, seed):
    rng = np.random.default_rng(1000 + seed)
    latent = int(rng.choice([32, 64, 96, 128, 256, 24, 48, 100]))      # any positive width (models/autoencoder.py:161)
    h, w = (int(16 * rng.integers(1, 9)) for _ in range(2))
    b = int(rng.integers(1, 12))
    chunk = int(rng.integers(1, 9))
    precision = ("fp32", "winograd", "split")[seed % 3]
    cin = 3 if seed % 4 else int(rng.choice([1, 2, 4, 6, 17, 32]))       # every fourth case: not an RGB model (models/autoencoder.py:161)
    m = vad.ConvAutoencoder(in_channels=cin, latent_dim=latent)
    st = load_synthetic(vad, m, 300 + seed)
    m = m.cuda().eval()
    m.precision, m.chunk = precision, chunk
    x = vad.synth.frames(2000 + seed, 0, b, cin, h, w)
    ref = torch_oracle.img_scores(_state(st), torch.from_numpy(x))
    tag = f"cin {cin} latent {latent} {h}x{w} b {b} chunk {chunk} {precision}"
    with torch.no_grad():
        out = m.score_all(torch.from_numpy(x).cuda())
        m.chunk = 128
        again = m.get_reconstruction_error(torch.from_numpy(x).cuda())
        if cin != 3:                                                          # (uint8 frames are 3-channel images)
            assert rel_err(out["scores"].cpu().numpy(), ref["scores"].numpy()) < 1e-5, tag
            assert np.abs(out["recon"].cpu().numpy() - ref["recon"].numpy()).max() < 5e-5, tag
            assert torch.equal(again, out["scores"]), tag
            return
        u8 = np.clip(np.round((x * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8).transpose(0, 2, 3, 1).copy()
        from_u8 = m.get_reconstruction_error(torch.from_numpy(u8).cuda())
        as_f32 = m.get_reconstruction_error(((torch.from_numpy(u8).permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5).cuda())
    assert rel_err(out["scores"].cpu().numpy(), ref["scores"].numpy()) < 1e-5, tag
    assert np.abs(out["recon"].cpu().numpy() - ref["recon"].numpy()).max() < 5e-5, tag
    assert torch.equal(again, out["scores"]), tag                         # chunking never changes a bit
    assert torch.equal(from_u8, as_f32), tag                             # uint8 ingest == the same frames as fp32


@pytest.mark.parametrize("seed", SEEDS)
def test_video_random_configuration(vad, seed):
    rng = np.random.default_rng(5000 + seed)
    latent = int(rng.choice([32, 64, 128, 24, 48, 100]))               # any positive widths (models/video_autoencoder.py:290-296)
    hid = int(rng.choice([64, 128, 32, 96])) if seed % 2 else latent
    layers = int(rng.integers(1, 4))
    h, w = (int(16 * rng.integers(1, 6)) for _ in range(2))
    b, t = int(rng.integers(1, 5)), int(rng.integers(1, 7))
    # split ConvLSTM step: x and h halves of equal width; winograd: any widths (layer 0 stays direct when they pad differently)
    precision = "split" if (seed % 3 == 2 and hid == latent) else ("winograd" if seed % 3 == 1 else "fp32")
    cin = 3 if seed % 4 != 1 else int(rng.choice([1, 2, 4, 5, 9]))           # every fourth case: not an RGB model (models/video_autoencoder.py:290)
    m = vad.VideoAutoencoder(in_channels=cin, latent_dim=latent, lstm_hidden_dim=hid, lstm_num_layers=layers)
    st = load_synthetic(vad, m, 700 + seed)
    m = m.cuda().eval()
    m.precision, m.chunk = precision, int(rng.integers(1, 4))
    x = vad.synth.clips(6000 + seed, 0, b, t, cin, h, w)
    ref = torch_oracle.vid_scores(_state(st), torch.from_numpy(x), hid, layers)
    with torch.no_grad():
        out = m.score_all(torch.from_numpy(x).cuda())
    tag = f"cin {cin} latent {latent} hid {hid} layers {layers} {h}x{w} b {b} t {t} {precision}"
    assert rel_err(out["frame"].cpu().numpy(), ref["frame"].numpy()) < 1e-5, tag
    assert rel_err(out["seq"].cpu().numpy(), ref["seq"].numpy()) < 1e-5, tag
    assert np.abs(out["recon"].cpu().numpy() - ref["recon"].numpy()).max() < 5e-5, tag
