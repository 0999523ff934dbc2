"""GPU parity of the whole scoring path, through the drop-in modules (ctypes -> libvad_hip.so), against
(1) the golden vectors captured from the reference itself and (2) the CPU oracle on fresh seeded inputs.

Tolerance: BASELINE.json's north_star asks for per-frame scores within 1e-4 relative of the reference CPU
path; the exact-fp32 MFMA path is held to 1e-5 here."""
import numpy as np
import pytest
import torch

from conftest import IMG_GOLDENS, VID_GOLDENS, auroc_slack, in_channels_of, load_synthetic, max_abs, rel_err
from oracle import torch_oracle

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-5
ACT_ATOL = 5e-5


def _img_model(vad, latent, wseed, in_ch=3):
    m = vad.ConvAutoencoder(in_channels=in_ch, latent_dim=latent)
    st = load_synthetic(vad, m, wseed)
    return m.cuda().eval(), st


def _vid_model(vad, latent, hid, layers, wseed, in_ch=3):
    m = vad.VideoAutoencoder(in_channels=in_ch, latent_dim=latent, lstm_hidden_dim=hid, lstm_num_layers=layers)
    st = load_synthetic(vad, m, wseed)
    return m.cuda().eval(), st


@pytest.mark.parametrize("name", IMG_GOLDENS)
def test_image_matches_reference_golden(vad, golden, name):
    """The reference's own outputs, including a latent_dim the channel tiling does not divide (100: zero-padded by the
    packer) and a 1-channel model (models/autoencoder.py:161 takes any in_channels / latent_dim)."""
    g = golden(name)
    cin = in_channels_of(g)
    m, _ = _img_model(vad, int(g["latent_dim"]), int(g["wseed"]), cin)
    x = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, int(g["n"]), cin, int(g["hw"]), int(g["hw"]))).cuda()
    before = vad.hip.calls["img_score"]
    with torch.no_grad():
        recon = m(x)
        emap = m.get_reconstruction_error(x, per_pixel=True)
        scores = m.get_reconstruction_error(x)
        lat = m.get_latent(x)
        allo = m.score_all(x)
    assert vad.hip.calls["img_score"] == before + 5          # the native path really ran
    assert recon.shape == x.shape and emap.shape == (x.shape[0], 1, *x.shape[2:]) and scores.shape == (x.shape[0],)
    assert lat.shape == g["latent"].shape
    assert rel_err(scores.cpu().numpy(), g["scores"]) < SCORE_RTOL
    assert max_abs(recon.cpu().numpy(), g["recon"]) < ACT_ATOL
    assert max_abs(emap.cpu().numpy(), g["errmap"]) < ACT_ATOL
    assert max_abs(lat.cpu().numpy(), g["latent"]) < 2e-4
    # one pass == three passes, bit for bit
    assert torch.equal(allo["scores"], scores) and torch.equal(allo["errmap"], emap) and torch.equal(allo["recon"], recon)


def test_image_256_frame_matches_reference_golden(vad, golden):
    g = golden("img_l256_256.npz")
    m, _ = _img_model(vad, 256, int(g["wseed"]))
    x = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, 1, 3, 256, 256)).cuda()
    with torch.no_grad():
        out = m.score_all(x)
    assert rel_err(out["scores"].cpu().numpy(), g["scores"]) < SCORE_RTOL
    assert max_abs(out["recon"].cpu().numpy()[:, :, ::8, ::8], g["recon_sub"]) < ACT_ATOL
    assert max_abs(out["errmap"].cpu().numpy()[:, :, ::8, ::8], g["errmap_sub"]) < ACT_ATOL


def test_image_config0_scores_and_auroc(vad, golden):
    """configs[0] on the GPU: the reference's 64 golden scores (batches of 16, evaluate.py:240) and AUROC."""
    g = golden("auroc_cfg0.npz")
    seed = int(g["seed"])
    m, _ = _img_model(vad, 256, int(g["wseed"]))
    labels = vad.synth.frame_label(seed, np.arange(64))
    patch = int(g["patch"])                                      # small patch: AUROC 0.82, not the trivially separable 1.0
    assert 0.6 < float(g["auroc"]) < 0.95
    batches = [{"image": vad.scoring.synth_frames_device(seed, s, 16, anomalies=patch), "label": labels[s:s + 16],
                "defect_type": ["defect" if l else "good" for l in labels[s:s + 16]]} for s in range(0, 64, 16)]
    auroc, lab, scores, per_defect = vad.scoring.compute_auroc(m, batches, "cuda")
    assert rel_err(scores, g["scores"]) < SCORE_RTOL
    # same ranking up to the pairs the reference itself separates by less than 2e-6 relative (float32 ulps)
    slack, close_pairs = auroc_slack(labels, g["scores"], 1e-6)
    assert close_pairs <= 4 and abs(auroc - float(g["auroc"])) <= slack + 1e-12
    assert set(per_defect) == {"good", "defect"} and per_defect["good"]["count"] + per_defect["defect"]["count"] == 64


def test_image_batch_and_chunk_independence(vad):
    """A frame's score must not depend on batch size, position or chunking: bit-exact (this is what makes
    multi-GPU sharding parity trivial)."""
    m, _ = _img_model(vad, 64, 3)
    x = vad.scoring.synth_frames_device(9, 0, 11, 64, 64)
    with torch.no_grad():
        full = m.get_reconstruction_error(x)
        m.chunk = 4
        chunked = m.get_reconstruction_error(x)
        singles = torch.cat([m.get_reconstruction_error(x[i:i + 1]) for i in range(11)])
        rev = m.get_reconstruction_error(x.flip(0)).flip(0)
    assert torch.equal(full, chunked) and torch.equal(full, singles) and torch.equal(full, rev)


def test_image_fresh_inputs_vs_oracle(vad):
    m, st = _img_model(vad, 128, 41)
    x = vad.synth.frames(555, 7, 3, 3, 96, 80)
    ref = torch_oracle.img_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, torch.from_numpy(x))
    with torch.no_grad():
        out = m.score_all(torch.from_numpy(x).cuda())
    assert rel_err(out["scores"].cpu().numpy(), ref["scores"].numpy()) < SCORE_RTOL
    assert max_abs(out["recon"].cpu().numpy(), ref["recon"].numpy()) < ACT_ATOL


def test_weight_update_invalidates_packed_cache(vad):
    m, _ = _img_model(vad, 32, 1)
    x = vad.scoring.synth_frames_device(1, 0, 2, 32, 32)
    with torch.no_grad():
        a = m.get_reconstruction_error(x)
        load_synthetic(vad, m, 2)          # in-place load_state_dict
        b = m.get_reconstruction_error(x)
        m.decoder.dec4[3].bias.add_(0.05)  # in-place parameter edit
        c = m.get_reconstruction_error(x)
    assert not torch.equal(a, b) and not torch.equal(b, c)


def test_inference_requires_gpu_tensor_and_train_mode_uses_autograd(vad):
    m, _ = _img_model(vad, 32, 1)
    with torch.no_grad(), pytest.raises(vad.hip.VadError):
        m.get_reconstruction_error(torch.zeros(1, 3, 32, 32))
    with torch.no_grad(), pytest.raises(vad.hip.VadError):
        m.get_reconstruction_error(torch.zeros(1, 3, 40, 32).cuda())
    x = vad.scoring.synth_frames_device(1, 0, 2, 32, 32)
    m.train()
    loss = ((m(x) - x) ** 2).mean()
    loss.backward()
    assert m.encoder.enc1[0].weight.grad is not None


@pytest.mark.parametrize("name", VID_GOLDENS)
def test_video_matches_reference_golden(vad, golden, name):
    """The reference's own outputs for the default model, `proj` variants, three ConvLSTM layers, widths the channel
    tiling does not divide (latent 48 / hidden 96 with proj; latent = hidden = 100 without) and a 2-channel model."""
    g = golden(name)
    cin = in_channels_of(g)
    m, _ = _vid_model(vad, int(g["latent_dim"]), int(g["hid"]), int(g["layers"]), int(g["wseed"]), cin)
    x = torch.from_numpy(vad.synth.clips(int(g["xseed"]), 0, int(g["b"]), int(g["t"]), cin, int(g["hw"]), int(g["hw"]))).cuda()
    before = vad.hip.calls["vid_score"]
    with torch.no_grad():
        recon = m(x)
        seq = m.get_reconstruction_error(x)
        frame = m.get_reconstruction_error(x, per_frame=True)
        emap = m.get_reconstruction_error(x, per_pixel=True)
        both = m.get_reconstruction_error(x, per_frame=True, per_pixel=True)
        allo = m.score_all(x)
    assert vad.hip.calls["vid_score"] == before + 6
    assert rel_err(seq.cpu().numpy(), g["seq"]) < SCORE_RTOL
    assert rel_err(frame.cpu().numpy(), g["frame"]) < SCORE_RTOL
    assert max_abs(recon.cpu().numpy(), g["recon"]) < ACT_ATOL
    assert max_abs(emap.cpu().numpy(), g["errmap"]) < ACT_ATOL
    assert torch.equal(both, emap)                                  # per_pixel wins (reference :373)
    assert torch.equal(allo["seq"], seq) and torch.equal(allo["frame"], frame) and torch.equal(allo["recon"], recon)


def test_clip_loop_matches_reference_golden(vad, golden):
    """Row a12: `scoring.score_clips` (the build's counterpart of the clip loop of evaluate_video.evaluate,
    reference evaluate_video.py:137-154) against the clip / frame scores the REFERENCE's loop produced for the same
    seeded clips in batches of 4 with a ragged last batch (tests/golden/clip_loop_64.npz, make_golden.clip_loop_fixture)."""
    g = golden("clip_loop_64.npz")
    n, t, hw, batch = int(g["n"]), int(g["t"]), int(g["hw"]), int(g["batch"])
    m, _ = _vid_model(vad, 128, 128, 2, int(g["wseed"]))
    labels = vad.synth.frame_label(int(g["xseed"]), np.arange(n))

    def loader():
        for s in range(0, n, batch):
            k = min(batch, n - s)
            yield {"frames": torch.from_numpy(vad.synth.clips(int(g["xseed"]), s, k, t, 3, hw, hw)), "label": labels[s:s + k]}

    before = vad.hip.calls["vid_score"]
    seq, lab = vad.scoring.score_clips(m, loader(), "cuda")
    seq2, lab2, frm = vad.scoring.score_clips(m, loader(), "cuda", per_frame=True)
    assert vad.hip.calls["vid_score"] == before + 2 * 3          # one native pass per batch, also with per_frame
    assert seq.shape == (n,) and frm.shape == (n, t) and seq.dtype == np.float32 and frm.dtype == np.float32
    assert rel_err(seq, g["seq_scores"]) < SCORE_RTOL and rel_err(frm, g["frame_scores"]) < SCORE_RTOL
    assert np.array_equal(seq, seq2) and np.array_equal(lab, g["labels"]) and np.array_equal(lab2, g["labels"])
    assert vad.scoring.roc_auc(lab, seq) == vad.scoring.roc_auc(g["labels"], g["seq_scores"])


def test_image_config1_full_size_vs_oracle(vad):
    """BASELINE configs[1] at its stated size - 512 frames x 256x256 in ONE launch group of 512 (the persistent grids, frame
    strides and tile variants the headline benchmark runs, evaluate.py:56-64 at the bench's batch): frames 0 / 255 / 511
    against the CPU oracle at 1e-5, and bit-equality with the same frames scored alone (a launch group of 1: the latency
    tilings and the gate-split kernel forms)."""
    m, st = _img_model(vad, 256, 7)
    assert int(m.chunk) == 512
    seed = 0xC0FFEE + 1
    x = vad.scoring.synth_frames_device(seed, 0, 512)
    idx = [0, 255, 511]
    with torch.no_grad():
        scores = m.get_reconstruction_error(x)
        alone = [m.get_reconstruction_error(x[i:i + 1]) for i in idx]
        emap = m.get_reconstruction_error(x[idx], per_pixel=True)
    assert scores.shape == (512,) and bool(torch.isfinite(scores).all())
    for i, a in zip(idx, alone):
        assert torch.equal(a, scores[i:i + 1]), i
    xs = torch.from_numpy(np.concatenate([vad.synth.frames(seed, i, 1, 3, 256, 256) for i in idx]))
    assert torch.equal(xs, x[idx].cpu())                           # device generator == numpy generator
    torch.set_num_threads(16)
    ref = torch_oracle.img_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, xs)
    assert rel_err(scores[idx].cpu().numpy(), ref["scores"].numpy()) < SCORE_RTOL
    assert max_abs(emap.cpu().numpy(), ref["errmap"].numpy()) < ACT_ATOL
    # size-independent property at the full size: the score vector of the batch in reverse order is the reverse, bit for bit
    with torch.no_grad():
        rev = m.get_reconstruction_error(x.flip(0).contiguous())
    assert torch.equal(rev.flip(0), scores)


def test_video_config2_full_size_vs_oracle(vad):
    """BASELINE configs[2] at its stated size - 64 clips x 10 frames x 256x256, default model, one launch group of 64
    clips (the grids and tile variants the benchmark runs): first / middle / last clip against the CPU oracle at 1e-5,
    and bit-equality with the same clips scored alone."""
    m, st = _vid_model(vad, 128, 128, 2, 8)
    seed = 0xC0FFEE + 2
    x = vad.scoring.synth_frames_device(seed, 0, 640).view(64, 10, 3, 256, 256)
    with torch.no_grad():
        out = m.score_seq_and_frames(x)
        idx = [0, 31, 63]
        alone = [m.score_seq_and_frames(x[i:i + 1]) for i in idx]
    assert out["seq"].shape == (64,) and out["frame"].shape == (64, 10)
    for i, a in zip(idx, alone):
        assert torch.equal(a["seq"], out["seq"][i:i + 1]) and torch.equal(a["frame"], out["frame"][i:i + 1])
    tst = {k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}
    xs = torch.from_numpy(np.stack([vad.synth.clips(seed, i, 1, 10, 3, 256, 256)[0] for i in idx]))
    assert torch.equal(xs, x[idx].cpu())                           # device generator == numpy generator
    torch.set_num_threads(16)
    ref = torch_oracle.vid_scores(tst, xs, 128, 2)
    assert rel_err(out["seq"][idx].cpu().numpy(), ref["seq"].numpy()) < SCORE_RTOL
    assert rel_err(out["frame"][idx].cpu().numpy(), ref["frame"].numpy()) < SCORE_RTOL


@pytest.mark.parametrize("clips", [24, 40, 48, 72])
def test_video_scores_do_not_depend_on_the_convlstm_kernel_form(vad, clips):
    """At 256x256 the ConvLSTM step is run by the small-grid (16x16x4) kernel or by the large (32x32x2) one, chosen per launch
    from the number of clips (conv_mfma.hip, vad_convlstm_step: below one work-group per CU, and above it wherever the large
    form would leave most CUs waiting for the fullest one - 40, 48, 72 clips).  Whatever the choice, a clip's scores are the
    bits it gets inside the 64-clip launch group of configs[2] (large kernel) and alone (small kernel)."""
    m, _ = _vid_model(vad, 128, 128, 2, 8)
    x = vad.scoring.synth_frames_device(0xC0FFEE + 2, 0, 72 * 3).view(72, 3, 3, 256, 256)
    with torch.no_grad():
        big = m.score_seq_and_frames(x[:64])
        got = m.score_seq_and_frames(x[:clips])
        alone = m.score_seq_and_frames(x[clips - 1:clips])
    k = min(clips, 64)
    assert torch.equal(got["seq"][:k], big["seq"][:k]) and torch.equal(got["frame"][:k], big["frame"][:k])
    assert torch.equal(got["seq"][-1:], alone["seq"]) and torch.equal(got["frame"][-1:], alone["frame"])


@pytest.mark.parametrize("latent,hid,layers,hw", [(128, 128, 2, 64), (64, 64, 3, 32), (32, 64, 1, 32), (100, 100, 2, 48)])
def test_convlstm_x_halves_ahead_of_the_recurrence_change_no_bit(vad, latent, hid, layers, hw):
    """Small launch groups compute bias + the x half of every ConvLSTM step's gate pre-activations ahead of the recurrence (one
    batched convolution for layer 0 - per SOURCE frame when windows overlap - and one launch per step on a helper stream for
    the layers above) and the step kernel resumes the accumulator chain with the h half: the chain is cut at a chunk boundary
    and stored as fp32, so every output is the same bits as with the split switched off (vad_debug_set_conv_variant bit 5),
    with the layer wavefront on or off."""
    l = vad.hip.lib()
    m, _ = _vid_model(vad, latent, hid, layers, 9)
    x = torch.from_numpy(vad.synth.clips(55, 0, 3, 5, 3, hw, hw)).cuda()
    frames = torch.from_numpy(vad.synth.frames(56, 0, 9, 3, hw, hw)).cuda()
    outs = []
    try:
        for bits, wf in ((1, 1), (1 | 32, 1), (1, 0), (1 | 32, 0), (1 | 64, 1), (1 | 128, 1), (1 | 128 | 32, 0)):   # bit 6 / 7: never / always the gate-split kernels
            l.vad_debug_set_conv_variant(bits)
            l.vad_debug_set_lstm_wavefront(wf)
            with torch.no_grad():
                outs.append((m.score_all(x), m.score_windows(frames, sequence_length=4, stride=1, recon=True)))
    finally:
        l.vad_debug_set_conv_variant(1)
        l.vad_debug_set_lstm_wavefront(1)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            for k in a:
                assert torch.equal(a[k], b[k]), k


def test_video_causal_and_clip_independent(vad):
    """Reference properties: perturbing frames >= k leaves frame scores < k bit-identical;
    a clip's scores do not depend on the rest of the batch."""
    m, _ = _vid_model(vad, 64, 64, 2, 5)
    x = torch.from_numpy(vad.synth.clips(77, 0, 5, 6, 3, 32, 32)).cuda()
    with torch.no_grad():
        base = m.get_reconstruction_error(x, per_frame=True)
        y = x.clone()
        y[:, 4:] = -y[:, 4:]
        pert = m.get_reconstruction_error(y, per_frame=True)
        solo = m.get_reconstruction_error(x[2:3], per_frame=True)
        m.chunk = 2
        chunked = m.get_reconstruction_error(x, per_frame=True)
    assert torch.equal(base[:, :4], pert[:, :4]) and not torch.equal(base[:, 4:], pert[:, 4:])
    assert torch.equal(base[2:3], solo) and torch.equal(base, chunked)


@pytest.mark.parametrize("t,stride,chunk", [(4, 1, 64), (6, 3, 2), (5, 5, 3)])
def test_video_dense_windows_equal_per_window_clips(vad, t, stride, chunk):
    """Row f-2: sliding windows over one video with the encoder shared between overlapping windows give bit-identical
    scores / maps / reconstructions to scoring every window as its own clip (reference evaluate_video.py:322-352)."""
    m, st = _vid_model(vad, 64, 64, 2, 6)
    f = 17
    frames = torch.from_numpy(vad.synth.frames(99, 0, f, 3, 32, 48)).cuda()
    nw = (f - t) // stride + 1
    clips = torch.stack([frames[k * stride:k * stride + t] for k in range(nw)])
    with torch.no_grad():
        ref = m.score_all(clips)
        m.window_chunk = chunk
        got = m.score_windows(frames, sequence_length=t, stride=stride, errmap=True, recon=True)
    assert got["seq"].shape == (nw,) and got["frame"].shape == (nw, t)
    for k in ("seq", "frame", "errmap", "recon"):
        assert torch.equal(got[k], ref[k]), k
    o = torch_oracle.vid_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, clips[-1:].cpu(), 64, 2)
    assert rel_err(got["frame"][-1:].cpu().numpy(), o["frame"].numpy()) < SCORE_RTOL
    with torch.no_grad(), pytest.raises(vad.hip.VadError):
        m.score_windows(frames[:3], sequence_length=4)


def test_trained_model_precision_gate(vad, golden):
    """SURVEY.md section 8(d) precision gate on the GPU: trained reference weights (strict state-dict load), the
    reference's synthetic test images fed as raw uint8 frames, scores within 1e-5 of the reference's own and the same
    AUROC / ranking."""
    g = golden("img_trained_l64.npz")
    m = vad.ConvAutoencoder(in_channels=3, latent_dim=int(g["latent_dim"]))
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=True)
    m = m.cuda().eval()
    xu = torch.from_numpy(g["test_u8"]).cuda()                                   # [30,256,256,3] uint8
    with torch.no_grad():
        out = m.score_all(xu)
    s = out["scores"].cpu().numpy()
    assert rel_err(s, g["scores"]) < SCORE_RTOL
    assert vad.scoring.roc_auc(g["labels"], s) == pytest.approx(float(g["auroc"]), abs=1e-12)
    assert np.array_equal(np.argsort(s), np.argsort(g["scores"]))
    assert max_abs(out["errmap"].cpu().numpy()[:1, :, ::8, ::8], g["errmap0_sub"]) < ACT_ATOL


def test_split_precision_mode_holds_parity(vad, golden):
    """Opt-in split-fp16 arithmetic (3 x fp16 MFMA, fp32 accumulate; `model.precision = "split"` -> VAD_PREC_SPLIT in
    every call): every score gate of the exact path — reference golden vectors on random weights, the trained-model gate,
    the ConvLSTM video model — within 1e-5 relative, i.e. 10x inside north_star's 1e-4 bar."""
    g = golden("img_l256_64.npz")
    m, _ = _img_model(vad, 256, int(g["wseed"]))
    m.precision = "split"
    x = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, int(g["n"]), 3, 64, 64)).cuda()
    with torch.no_grad():
        out = m.score_all(x)
    assert rel_err(out["scores"].cpu().numpy(), g["scores"]) < SCORE_RTOL
    assert max_abs(out["recon"].cpu().numpy(), g["recon"]) < ACT_ATOL
    t = golden("img_trained_l64.npz")
    mt = vad.ConvAutoencoder(in_channels=3, latent_dim=64)
    mt.load_state_dict({k[2:]: torch.from_numpy(t[k]) for k in t.files if k.startswith("w.")}, strict=True)
    mt = mt.cuda().eval()
    mt.precision = "split"
    with torch.no_grad():
        s = mt.get_reconstruction_error(torch.from_numpy(t["test_u8"]).cuda()).cpu().numpy()
    assert rel_err(s, t["scores"]) < SCORE_RTOL
    assert np.array_equal(np.argsort(s), np.argsort(t["scores"]))
    v = golden("vid_default_64.npz")
    mv, _ = _vid_model(vad, 128, 128, 2, int(v["wseed"]))
    mv.precision = "split"
    xv = torch.from_numpy(vad.synth.clips(int(v["xseed"]), 0, int(v["b"]), int(v["t"]), 3, 64, 64)).cuda()
    with torch.no_grad():
        ov = mv.score_all(xv)
    assert rel_err(ov["frame"].cpu().numpy(), v["frame"]) < SCORE_RTOL
    assert max_abs(ov["recon"].cpu().numpy(), v["recon"]) < ACT_ATOL
    # the mode belongs to the model: a default model scored afterwards (and the first one flipped back) is exact again
    m2, _ = _img_model(vad, 256, int(g["wseed"]))
    with torch.no_grad():
        s2 = m2.get_reconstruction_error(x)
        m.precision = "fp32"
        s3 = m.get_reconstruction_error(x)
    assert rel_err(s2.cpu().numpy(), g["scores"]) < SCORE_RTOL and torch.equal(s2, s3)


def test_winograd_mode_holds_parity(vad, golden):
    """Opt-in Winograd arithmetic (`model.precision = "winograd"` -> VAD_PREC_WINO: every 3x3 convolution behind the first
    layer as F(2x2,3x3) on the exact-fp32 MFMA, csrc/conv_wino.hip).  All-fp32 but another rounding order than the direct
    kernels, so not bit-identical to them; every score gate of the exact path must hold at the same 1e-5: the reference's golden
    vectors on random weights (latent 256 and a latent the channel tiling pads), the trained-model gate on raw uint8 frames
    (scores, ranking, AUROC), configs[0]'s 64 scores, and the ConvLSTM video model (its encoder's convolutions)."""
    for name in ("img_l256_64.npz", "img_l100_32.npz", "img_c1_l24_32.npz"):
        g = golden(name)
        cin = in_channels_of(g)
        m, _ = _img_model(vad, int(g["latent_dim"]), int(g["wseed"]), cin)
        x = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, int(g["n"]), cin, int(g["hw"]), int(g["hw"]))).cuda()
        with torch.no_grad():
            exact = m.score_all(x)
            m.precision = "winograd"
            out = m.score_all(x)
            lat = m.get_latent(x)
        assert rel_err(out["scores"].cpu().numpy(), g["scores"]) < SCORE_RTOL, name
        assert max_abs(out["recon"].cpu().numpy(), g["recon"]) < ACT_ATOL and max_abs(out["errmap"].cpu().numpy(), g["errmap"]) < ACT_ATOL
        assert max_abs(lat.cpu().numpy(), g["latent"]) < 2e-4
        assert not torch.equal(out["recon"], exact["recon"])                    # it IS other arithmetic, and says so
    t = golden("img_trained_l64.npz")
    mt = vad.ConvAutoencoder(in_channels=3, latent_dim=64)
    mt.load_state_dict({k[2:]: torch.from_numpy(t[k]) for k in t.files if k.startswith("w.")}, strict=True)
    mt = mt.cuda().eval()
    mt.precision = "winograd"
    with torch.no_grad():
        s = mt.get_reconstruction_error(torch.from_numpy(t["test_u8"]).cuda()).cpu().numpy()
    assert rel_err(s, t["scores"]) < SCORE_RTOL
    assert np.array_equal(np.argsort(s), np.argsort(t["scores"]))
    assert vad.scoring.roc_auc(t["labels"], s) == pytest.approx(float(t["auroc"]), abs=1e-12)
    a = golden("auroc_cfg0.npz")
    ma, _ = _img_model(vad, 256, int(a["wseed"]))
    ma.precision = "winograd"
    with torch.no_grad():                                                        # batches of 16, evaluate.py:240
        sa = torch.cat([ma.get_reconstruction_error(vad.scoring.synth_frames_device(int(a["seed"]), i, 16, anomalies=int(a["patch"])))
                        for i in range(0, 64, 16)]).cpu().numpy()
    assert rel_err(sa, a["scores"]) < SCORE_RTOL
    for name in ("vid_default_64.npz", "vid_l48_h96_32.npz"):
        v = golden(name)
        mv, _ = _vid_model(vad, int(v["latent_dim"]), int(v["hid"]), int(v["layers"]), int(v["wseed"]))
        mv.precision = "winograd"
        xv = torch.from_numpy(vad.synth.clips(int(v["xseed"]), 0, int(v["b"]), int(v["t"]), 3, int(v["hw"]), int(v["hw"]))).cuda()
        with torch.no_grad():
            ov = mv.score_all(xv)
        assert rel_err(ov["frame"].cpu().numpy(), v["frame"]) < SCORE_RTOL, name
        assert rel_err(ov["seq"].cpu().numpy(), v["seq"]) < SCORE_RTOL
        assert max_abs(ov["recon"].cpu().numpy(), v["recon"]) < ACT_ATOL
    # frames whose 16th is odd (48 x 80: 3 x 5 ConvLSTM maps, partial Winograd tiles) against the oracle, and dense windows ==
    # per-window clips, bit for bit, in this mode
    mo, sto = _vid_model(vad, 128, 128, 2, 13)
    mo.precision = "winograd"
    xo = torch.from_numpy(vad.synth.clips(31, 0, 2, 3, 3, 48, 80)).cuda()
    with torch.no_grad():
        oo = mo.score_all(xo)
        fr = vad.scoring.synth_frames_device(32, 0, 7, 48, 80)
        dense = mo.score_windows(fr, sequence_length=3, stride=2)
        each = mo.score_seq_and_frames(torch.stack([fr[k:k + 3] for k in range(0, 5, 2)]))
    ro = torch_oracle.vid_scores({k: torch.from_numpy(np.asarray(v)) for k, v in sto.items()}, xo.cpu(), 128, 2)
    assert rel_err(oo["frame"].cpu().numpy(), ro["frame"].numpy()) < SCORE_RTOL and max_abs(oo["recon"].cpu().numpy(), ro["recon"].numpy()) < ACT_ATOL
    assert torch.equal(dense["frame"], each["frame"]) and torch.equal(dense["seq"], each["seq"])
    # clip-independence across the two ConvLSTM forms of this mode: 40 clips in one launch group fill the chip (the cell fused
    # into the gate convolution's epilogue), a clip alone does not (one N-tile per wave + the pointwise cell launch): same bits
    xc = vad.scoring.synth_frames_device(33, 0, 40 * 3, 64, 64).view(40, 3, 3, 64, 64)
    mo.chunk = 64
    with torch.no_grad():
        big = mo.score_seq_and_frames(xc)
        alone = [mo.score_seq_and_frames(xc[i:i + 1]) for i in (0, 21, 39)]
    for i, a in zip((0, 21, 39), alone):
        assert torch.equal(a["frame"], big["frame"][i:i + 1]) and torch.equal(a["seq"], big["seq"][i:i + 1]), i
    rc = torch_oracle.vid_scores({k: torch.from_numpy(np.asarray(v)) for k, v in sto.items()}, xc[[0, 39]].cpu(), 128, 2)
    assert rel_err(big["frame"][[0, 39]].cpu().numpy(), rc["frame"].numpy()) < SCORE_RTOL
    # frame-independence holds in this mode too: a frame's score does not depend on its batch or position
    g = golden("img_l256_64.npz")
    m, _ = _img_model(vad, 256, int(g["wseed"]))
    m.precision = "winograd"
    xs = vad.scoring.synth_frames_device(21, 0, 37, 64, 64)
    with torch.no_grad():
        whole = m.get_reconstruction_error(xs)
        m.chunk = 5
        parts = m.get_reconstruction_error(xs)
        single = torch.cat([m.get_reconstruction_error(xs[i:i + 1]) for i in (0, 17, 36)])
    assert torch.equal(whole, parts) and torch.equal(single, whole[[0, 17, 36]])


def test_winograd_mode_through_capture_windows_and_threads(vad):
    """The Winograd mode through the less-travelled entry points: hipGraph capture / replay (image; video at a one-window, a
    4-clip and a chip-filling launch group: wavefront on helper streams, the two-launch and the fused-cell ConvLSTM forms inside
    a capture), uint8 dense windows == the same frames as fp32, and three models in three arithmetic modes scored concurrently
    from three threads on three streams - every result bit-equal to its single-threaded value."""
    import threading
    mi, _ = _img_model(vad, 64, 3)
    mi.precision = "winograd"
    xa, xb = vad.scoring.synth_frames_device(9, 0, 16, 64, 64), vad.scoring.synth_frames_device(9, 100, 16, 64, 64)
    with torch.no_grad():
        g = mi.capture(xa, scores=True, errmap=True)
        eb = mi.score_all(xb)
        ob = g.replay(xb)
    assert torch.equal(ob["scores"], eb["scores"]) and torch.equal(ob["errmap"], eb["errmap"])
    mv, _ = _vid_model(vad, 128, 128, 2, 5)
    mv.precision = "winograd"
    for b, t in ((4, 16), (1, 16), (40, 3)):
        ca = vad.scoring.synth_frames_device(11, 0, b * t, 32, 32).view(b, t, 3, 32, 32)
        cb = vad.scoring.synth_frames_device(11, 500, b * t, 32, 32).view(b, t, 3, 32, 32)
        with torch.no_grad():
            gv = mv.capture(ca, seq=True, frame=True, recon=True)
            want = mv.score_all(cb)
            for _ in range(2):
                got = gv.replay(cb)
                for k in ("seq", "frame", "recon"):
                    assert torch.equal(got[k], want[k]), (b, t, k)
    u8n = vad.synth.frames_u8(21, 0, 9, 3, 48, 32)
    with torch.no_grad():
        d = mv.score_windows(torch.from_numpy(np.ascontiguousarray(u8n.transpose(0, 2, 3, 1))).cuda(), sequence_length=4, stride=1)
        d2 = mv.score_windows(torch.from_numpy(vad.synth.u8_to_unit(u8n)).cuda(), sequence_length=4, stride=1)
    assert torch.equal(d["frame"], d2["frame"]) and torch.equal(d["seq"], d2["seq"])
    models = []
    for prec in ("fp32", "split", "winograd"):
        m, _ = _img_model(vad, 64, 3)
        m.precision = prec
        models.append(m)
    x = vad.scoring.synth_frames_device(5, 0, 24, 64, 64)
    with torch.no_grad():
        single = [m.get_reconstruction_error(x).clone() for m in models]
    res, errs = {}, []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(20):
                    r = models[i].get_reconstruction_error(x)
                st.synchronize()
            res[i] = r
        except Exception as e:                              # noqa: BLE001 - reported below
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=120)
    assert not errs and not any(th.is_alive() for th in ths), errs
    for i in range(3):
        assert torch.equal(res[i], single[i]), i


def test_two_threads_with_different_precision_do_not_interfere(vad):
    """SURVEY.md section 8(b) threading contract (the reference's UI calls one global model from worker threads,
    main.py:50,274): an exact-fp32 model and a split-fp16 model scored CONCURRENTLY from two Python threads on two
    streams give, every time, bit-for-bit what each gives alone.  With the arithmetic mode as process state (ABI 1) the
    two raced: a blob packed for one mode was launched under the other."""
    import threading
    ma, _ = _img_model(vad, 64, 3)
    mb, _ = _img_model(vad, 64, 3)
    mb.precision = "split"
    va, _ = _vid_model(vad, 64, 64, 2, 5)
    va.precision = "split"
    x = vad.scoring.synth_frames_device(9, 0, 24, 64, 64)
    xc = vad.scoring.synth_frames_device(10, 0, 24, 32, 32).view(4, 6, 3, 32, 32)
    with torch.no_grad():
        want = {"a": ma.get_reconstruction_error(x), "b": mb.get_reconstruction_error(x), "v": va.score_seq_and_frames(xc)["frame"]}
    assert not torch.equal(want["a"], want["b"])                    # the two modes do differ in the last bits
    torch.cuda.synchronize()
    rounds, errors, start = 40, [], threading.Barrier(3)

    def worker(tag, fn):
        try:
            stream = torch.cuda.Stream()
            start.wait()
            with torch.no_grad(), torch.cuda.stream(stream):
                for i in range(rounds):
                    got = fn()
                    stream.synchronize()
                    if not torch.equal(got, want[tag]):
                        errors.append(f"{tag}: round {i} differs from the single-threaded result")
                        return
        except Exception as e:                                      # noqa: BLE001 - reported by the main thread
            errors.append(f"{tag}: {e!r}")

    threads = [threading.Thread(target=worker, args=("a", lambda: ma.get_reconstruction_error(x))),
               threading.Thread(target=worker, args=("b", lambda: mb.get_reconstruction_error(x))),
               threading.Thread(target=worker, args=("v", lambda: va.score_seq_and_frames(xc)["frame"]))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads) and not errors, errors


def test_captured_graph_replays_bit_identically(vad):
    """hipGraph capture / replay of one scoring call (vad_graph_*, `model.capture`) at the reference's call sizes: batch 16
    images (evaluate.py:240), 4 clips x 16 frames (evaluate_video.py:416: the small-grid ConvLSTM kernel and the two-stream
    layer wavefront inside the capture), one window (evaluate_video.py:344).  A replay on NEW frames equals the eager call
    on those frames, bit for bit, every time."""
    mi, _ = _img_model(vad, 64, 3)
    xa, xb = vad.scoring.synth_frames_device(9, 0, 16, 64, 64), vad.scoring.synth_frames_device(9, 100, 16, 64, 64)
    with torch.no_grad():
        g = mi.capture(xa, scores=True, errmap=True)
        ea, eb = mi.score_all(xa), mi.score_all(xb)
        n0 = vad.hip.calls["img_score"]
        for _ in range(3):
            ob = g.replay(xb)
            assert torch.equal(ob["scores"], eb["scores"]) and torch.equal(ob["errmap"], eb["errmap"])
        oa = g.replay(xa)
        assert torch.equal(oa["scores"], ea["scores"]) and not torch.equal(ea["scores"], eb["scores"])
        assert vad.hip.calls["img_score"] == n0 and vad.hip.calls["graph_replay"] >= 4      # no eager launch behind a replay
    mv, _ = _vid_model(vad, 128, 128, 2, 5)
    for b, t in ((4, 16), (1, 16)):
        ca = vad.scoring.synth_frames_device(11, 0, b * t, 32, 32).view(b, t, 3, 32, 32)
        cb = vad.scoring.synth_frames_device(11, 500, b * t, 32, 32).view(b, t, 3, 32, 32)
        with torch.no_grad():
            gv = mv.capture(ca, seq=True, frame=True, recon=True)
            want = mv.score_all(cb)
            for _ in range(3):
                got = gv.replay(cb)
                for k in ("seq", "frame", "recon"):
                    assert torch.equal(got[k], want[k]), (b, t, k)
    torch.cuda.synchronize()


def test_wide_channel_models_all_entry_points_and_modes(vad, golden):
    """Models with MORE than 3 planes (round 4; csrc/wide_io.hip): the reference's own scores / reconstructions for a 5-channel
    image model and a 4-channel video model (tests/golden/make_golden.py) through the eager call, hipGraph capture / replay,
    `get_latent`, chunking, the opt-in arithmetic modes and dense windows."""
    g = golden("img_c5_l32_32.npz")
    cin = in_channels_of(g)
    assert cin == 5
    mi, sti = _img_model(vad, int(g["latent_dim"]), int(g["wseed"]), cin)
    n, hw = int(g["n"]), int(g["hw"])
    xa = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, n, cin, hw, hw)).cuda()
    xb = torch.from_numpy(vad.synth.frames(int(g["xseed"]) + 1, 0, n, cin, hw, hw)).cuda()
    with torch.no_grad():
        ea, eb = mi.score_all(xa), mi.score_all(xb)
        assert ea["recon"].shape == xa.shape and ea["errmap"].shape == (n, 1, hw, hw)
        assert rel_err(ea["scores"].cpu().numpy(), g["scores"]) < SCORE_RTOL
        assert max_abs(ea["recon"].cpu().numpy(), g["recon"]) < ACT_ATOL
        assert max_abs(mi.get_latent(xa).cpu().numpy(), g["latent"]) < 2e-4
        assert torch.equal(mi(xa), ea["recon"]) and torch.equal(mi.get_reconstruction_error(xa), ea["scores"])
        assert torch.equal(mi.get_reconstruction_error(xa, per_pixel=True), ea["errmap"])
        mi.chunk = 2                                                          # chunking never changes a bit
        assert torch.equal(mi.get_reconstruction_error(xa), ea["scores"])
        mi.chunk = 128
        cap = mi.capture(xa, scores=True, errmap=True, recon=True, latent=True)
        for _ in range(2):
            ob = cap.replay(xb)
            for k in ("scores", "errmap", "recon"):
                assert torch.equal(ob[k], eb[k]), k
        for mode in ("split", "winograd"):
            mi.precision = mode
            assert rel_err(mi.get_reconstruction_error(xa).cpu().numpy(), g["scores"]) < SCORE_RTOL, mode
        mi.precision = "fp32"
        with pytest.raises(vad.hip.VadError, match="uint8 .* needs in_channels == 3"):
            mi.get_reconstruction_error(torch.zeros(n, hw, hw, cin, dtype=torch.uint8).cuda())
    # a frame size and a batch the golden does not have, against the CPU oracle
    xo = vad.synth.frames(991, 0, 3, cin, 48, 80)
    ref = torch_oracle.img_scores({k: torch.from_numpy(np.asarray(v)) for k, v in sti.items()}, torch.from_numpy(xo))
    with torch.no_grad():
        assert rel_err(mi.get_reconstruction_error(torch.from_numpy(xo).cuda()).cpu().numpy(), ref["scores"].numpy()) < SCORE_RTOL

    gv = golden("vid_c4_l32_32.npz")
    cv = in_channels_of(gv)
    assert cv == 4
    mv, stv = _vid_model(vad, int(gv["latent_dim"]), int(gv["hid"]), int(gv["layers"]), int(gv["wseed"]), cv)
    b, t, hw = int(gv["b"]), int(gv["t"]), int(gv["hw"])
    ca = torch.from_numpy(vad.synth.clips(int(gv["xseed"]), 0, b, t, cv, hw, hw)).cuda()
    cb = torch.from_numpy(vad.synth.clips(int(gv["xseed"]) + 1, 0, b, t, cv, hw, hw)).cuda()
    with torch.no_grad():
        wa, wb = mv.score_all(ca), mv.score_all(cb)
        assert wa["recon"].shape == ca.shape
        assert rel_err(wa["frame"].cpu().numpy(), gv["frame"]) < SCORE_RTOL and rel_err(wa["seq"].cpu().numpy(), gv["seq"]) < SCORE_RTOL
        assert max_abs(wa["recon"].cpu().numpy(), gv["recon"]) < ACT_ATOL
        mv.chunk = 1
        assert torch.equal(mv.score_all(ca)["frame"], wa["frame"])
        mv.chunk = 64
        cap = mv.capture(ca, seq=True, frame=True, errmap=True, recon=True)
        for _ in range(2):
            ob = cap.replay(cb)
            for k in ("seq", "frame", "errmap", "recon"):
                assert torch.equal(ob[k], wb[k]), k
        for mode in ("split", "winograd"):
            mv.precision = mode
            assert rel_err(mv.score_all(ca)["frame"].cpu().numpy(), gv["frame"]) < SCORE_RTOL, mode
        mv.precision = "fp32"
        frames = torch.from_numpy(vad.synth.frames(78, 0, 9, cv, hw, hw)).cuda()
        tw = 4
        dense = mv.score_windows(frames, sequence_length=tw, stride=2, errmap=True, recon=True)
        clips = torch.stack([frames[k:k + tw] for k in range(0, 9 - tw + 1, 2)])
        each = mv.score_all(clips)
    assert dense["recon"].shape == clips.shape
    for k in ("seq", "frame", "errmap", "recon"):
        assert torch.equal(dense[k], each[k]), k
    # 32 planes (the largest width the library takes) against the oracle; 33 is refused before anything is packed
    m32 = vad.ConvAutoencoder(in_channels=32, latent_dim=32)
    st32 = load_synthetic(vad, m32, 77)
    m32 = m32.cuda().eval()
    x32 = vad.synth.frames(992, 0, 2, 32, 32, 32)
    ref = torch_oracle.img_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st32.items()}, torch.from_numpy(x32))
    with torch.no_grad():
        assert rel_err(m32.get_reconstruction_error(torch.from_numpy(x32).cuda()).cpu().numpy(), ref["scores"].numpy()) < SCORE_RTOL
        with pytest.raises(vad.hip.VadError, match="in_channels"):
            vad.ConvAutoencoder(in_channels=33, latent_dim=32).cuda().eval().get_reconstruction_error(torch.zeros(1, 33, 32, 32).cuda())


def test_one_and_two_channel_models_capture_latent_and_windows(vad, golden):
    """1- and 2-channel models (models/autoencoder.py:161, models/video_autoencoder.py:290 take any in_channels) through the
    entry points the per-call narrowing does not cover by itself: hipGraph capture / replay (the captured kernels work on 3
    planes: the input buffer is widened once outside the capture, the outputs are narrowed and rescaled after each replay),
    `get_latent`, and `score_windows`.  Every result equals the eager call or the reference's golden values."""
    g = golden("img_c1_l24_32.npz")
    mi, _ = _img_model(vad, int(g["latent_dim"]), int(g["wseed"]), 1)
    n, hw = int(g["n"]), int(g["hw"])
    xa = torch.from_numpy(vad.synth.frames(int(g["xseed"]), 0, n, 1, hw, hw)).cuda()
    xb = torch.from_numpy(vad.synth.frames(int(g["xseed"]) + 1, 0, n, 1, hw, hw)).cuda()
    with torch.no_grad():
        cap = mi.capture(xa, scores=True, errmap=True, recon=True, latent=True)
        ea, eb = mi.score_all(xa), mi.score_all(xb)
        lat_b = mi.get_latent(xb)
        for _ in range(2):
            ob = cap.replay(xb)
            assert ob["recon"].shape == xb.shape and ob["errmap"].shape == (n, 1, hw, hw)
            for k in ("scores", "errmap", "recon"):
                assert torch.equal(ob[k], eb[k]), k
            assert torch.equal(ob["latent"], lat_b)
        oa = cap.replay(xa)
        assert torch.equal(oa["scores"], ea["scores"]) and not torch.equal(ea["scores"], eb["scores"])
    assert rel_err(oa["scores"].cpu().numpy(), g["scores"]) < SCORE_RTOL            # the reference's own scores, through a replay
    assert max_abs(oa["recon"].cpu().numpy(), g["recon"]) < ACT_ATOL

    gv = golden("vid_c2_l32_h40_32.npz")
    mv, stv = _vid_model(vad, int(gv["latent_dim"]), int(gv["hid"]), int(gv["layers"]), int(gv["wseed"]), 2)
    b, t, hw = int(gv["b"]), int(gv["t"]), int(gv["hw"])
    ca = torch.from_numpy(vad.synth.clips(int(gv["xseed"]), 0, b, t, 2, hw, hw)).cuda()
    cb = torch.from_numpy(vad.synth.clips(int(gv["xseed"]) + 1, 0, b, t, 2, hw, hw)).cuda()
    with torch.no_grad():
        cap = mv.capture(ca, seq=True, frame=True, errmap=True, recon=True)
        wa, wb = mv.score_all(ca), mv.score_all(cb)
        for _ in range(2):
            ob = cap.replay(cb)
            assert ob["recon"].shape == cb.shape
            for k in ("seq", "frame", "errmap", "recon"):
                assert torch.equal(ob[k], wb[k]), k
        oa = cap.replay(ca)
    assert rel_err(oa["frame"].cpu().numpy(), gv["frame"]) < SCORE_RTOL
    assert max_abs(oa["recon"].cpu().numpy(), gv["recon"]) < ACT_ATOL
    # dense windows over one 2-channel video == every window scored as its own clip
    frames = torch.from_numpy(vad.synth.frames(77, 0, 9, 2, hw, hw)).cuda()
    tw = 4
    with torch.no_grad():
        dense = mv.score_windows(frames, sequence_length=tw, stride=1, errmap=True, recon=True)
        clips = torch.stack([frames[k:k + tw] for k in range(9 - tw + 1)])
        each = mv.score_all(clips)
    assert dense["recon"].shape == clips.shape
    for k in ("seq", "frame", "errmap", "recon"):
        assert torch.equal(dense[k], each[k]), k
    ref = torch_oracle.vid_scores({k: torch.from_numpy(np.asarray(v)) for k, v in stv.items()}, frames[:tw][None].cpu(),
                                  int(gv["hid"]), int(gv["layers"]))
    assert rel_err(dense["frame"][0].cpu().numpy(), ref["frame"][0].numpy()) < SCORE_RTOL
    # uint8 frames are offered for 3-channel models only, and say so before anything is packed
    with torch.no_grad(), pytest.raises(vad.hip.VadError, match="uint8 .* needs in_channels == 3"):
        mv.score_windows(torch.zeros(9, hw, hw, 2, dtype=torch.uint8).cuda(), sequence_length=tw)
    with torch.no_grad(), pytest.raises(vad.hip.VadError, match="uint8 .* needs in_channels == 3"):
        mi.get_reconstruction_error(torch.zeros(2, hw, hw, 1, dtype=torch.uint8).cuda())


def test_blob_launched_under_the_wrong_precision_is_rejected_on_the_device(vad):
    """ABI 2 safety net: the packed blob's header carries the mode it was packed for; a launch that names another mode
    returns NaN scores (checked by the finalising kernel on the device), never a plausible wrong number."""
    l = vad.hip.lib()
    m, _ = _img_model(vad, 64, 5)
    x = vad.scoring.synth_frames_device(1, 0, 2, 64, 64)
    packed = m._packed(x.device)                                    # packed for fp32
    need = l.vad_img_workspace_bytes(2, 64, 64, 64)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    scores = torch.zeros(2, device="cuda")
    s = vad.hip.current_stream()
    args = (2, 64, 64, 64, packed.data_ptr(), ws.data_ptr(), need, 2, scores.data_ptr(), None, None, None, s)
    assert l.vad_img_score_x(x.data_ptr(), 0, vad.hip.PREC_SPLIT, *args) == 0
    torch.cuda.synchronize()
    assert torch.isnan(scores).all()
    assert l.vad_img_score_x(x.data_ptr(), 0, vad.hip.PREC_FP32, *args) == 0
    torch.cuda.synchronize()
    with torch.no_grad():
        assert torch.equal(scores, m.get_reconstruction_error(x))
    assert l.vad_img_score_x(x.data_ptr(), 0, 5, *args) == -1 and b"precision" in l.vad_last_error()


def test_uint8_ingest_is_bit_identical(vad):
    """Row f-3: raw uint8 NHWC frames, normalised inside the kernels (reference transform utils/dataset.py:65-70),
    give bit-identical scores / maps / reconstructions to feeding the normalised fp32 NCHW tensor."""
    u8 = vad.synth.frames_u8(31, 5, 6, 3, 48, 64, anomalies=True)                    # [N,3,H,W] uint8
    xf = torch.from_numpy(vad.synth.u8_to_unit(u8)).cuda()
    xu = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).cuda()     # [N,H,W,3]
    m, _ = _img_model(vad, 64, 9)
    with torch.no_grad():
        a, b = m.score_all(xf), m.score_all(xu)
        assert torch.equal(m.get_latent(xf), m.get_latent(xu))
    for k in ("scores", "errmap", "recon"):
        assert torch.equal(a[k], b[k]), k
    v, _ = _vid_model(vad, 64, 64, 2, 10)
    with torch.no_grad():
        c, d = v.score_all(xf.view(2, 3, 3, 48, 64)), v.score_all(xu.view(2, 3, 48, 64, 3))
        e, f = v.score_windows(xf, sequence_length=4, stride=1), v.score_windows(xu, sequence_length=4, stride=1)
    for k in ("seq", "frame", "errmap", "recon"):
        assert torch.equal(c[k], d[k]), k
    assert torch.equal(e["frame"], f["frame"]) and torch.equal(e["seq"], f["seq"])


def test_video_fresh_inputs_vs_oracle(vad):
    m, st = _vid_model(vad, 128, 128, 2, 43)
    x = vad.synth.clips(321, 3, 2, 5, 3, 48, 64)
    ref = torch_oracle.vid_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, torch.from_numpy(x), 128, 2)
    with torch.no_grad():
        out = m.score_all(torch.from_numpy(x).cuda())
    assert rel_err(out["seq"].cpu().numpy(), ref["seq"].numpy()) < SCORE_RTOL
    assert rel_err(out["frame"].cpu().numpy(), ref["frame"].numpy()) < SCORE_RTOL
    assert max_abs(out["recon"].cpu().numpy(), ref["recon"].numpy()) < ACT_ATOL


def test_empty_batches_give_empty_outputs(vad):
    """Edge case: the reference's modules accept a batch of zero items (every torch layer does) and return empty tensors of the
    right shape; so do the HIP-backed ones, without launching anything."""
    mi, _ = _img_model(vad, 128, 7)
    mv, _ = _vid_model(vad, 128, 128, 2, 43)
    with torch.no_grad():
        assert mi(torch.empty(0, 3, 64, 64, device="cuda")).shape == (0, 3, 64, 64)
        assert mi.get_reconstruction_error(torch.empty(0, 3, 64, 64, device="cuda")).shape == (0,)
        out = mv.score_all(torch.empty(0, 5, 3, 64, 64, device="cuda"))
        assert out["seq"].shape == (0,) and out["frame"].shape == (0, 5) and out["recon"].shape == (0, 5, 3, 64, 64)
        # a ragged tail after full batches: 5 clips in batches of 4 equal the same 5 clips in one call, bit for bit
        x = torch.from_numpy(vad.synth.clips(99, 0, 5, 4, 3, 32, 32)).cuda()
        whole = mv.score_all(x)["seq"]
        parts = torch.cat([mv.score_all(x[:4])["seq"], mv.score_all(x[4:])["seq"]])
        assert torch.equal(whole, parts)


def test_full_size_properties(vad):
    """BASELINE configs[1]/[2] frame size (256x256), moderate batch: size-independent checks — score equals the
    mean of the error map, clip score equals the mean of frame scores, and a strided subset matches the oracle."""
    m, st = _img_model(vad, 256, 7)
    x = vad.scoring.synth_frames_device(0xC0FFEE + 1, 0, 48)
    with torch.no_grad():
        out = m.score_all(x)
    s = out["scores"].cpu().numpy()
    assert rel_err(out["errmap"].mean(dim=(1, 2, 3)).cpu().numpy(), s) < 2e-6
    assert rel_err(((x - out["recon"]) ** 2).mean(dim=(1, 2, 3)).cpu().numpy(), s) < 2e-6
    idx = [0, 17, 47]
    ref = torch_oracle.img_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, x[idx].cpu())
    assert rel_err(s[idx], ref["scores"].numpy()) < SCORE_RTOL
    v, _ = _vid_model(vad, 128, 128, 2, 8)
    xc = vad.scoring.synth_frames_device(0xC0FFEE + 2, 0, 40).view(4, 10, 3, 256, 256)
    with torch.no_grad():
        vo = v.score_all(xc)
    assert rel_err(vo["frame"].mean(dim=1).cpu().numpy(), vo["seq"].cpu().numpy()) < 2e-6
    assert rel_err(vo["errmap"].mean(dim=(2, 3, 4)).cpu().numpy(), vo["frame"].cpu().numpy()) < 2e-6


# ------------------------------------------------------------------------------------------------ criteria (row f-4)

def test_ssim_combined_losses_match_reference_golden(vad, golden):
    """SSIMLoss / CombinedLoss through vad_ssim_mse against the REFERENCE's outputs (tests/golden/losses.npz,
    utils/losses.py run by make_golden.py)."""
    g = golden("losses.npz")
    x = torch.from_numpy(vad.synth.frames(77, 0, 2, 3, 32, 32)).cuda()
    y = torch.from_numpy(vad.synth.frames(78, 0, 2, 3, 32, 32)).cuda() * 0.25 + x * 0.75
    n0 = vad.hip.calls.get("ssim", 0)
    with torch.no_grad():
        assert abs(float(vad.SSIMLoss()(y, x)) - float(g["ssim"])) < 2e-6
        assert abs(float(vad.CombinedLoss(alpha=0.5)(y, x)) - float(g["combined"])) < 2e-6
        assert abs(float(vad.CombinedLoss(alpha=0.3, window_size=7)(y, x)) - float(g["combined_03"])) < 2e-6
    assert vad.hip.calls["ssim"] == n0 + 3



@pytest.mark.parametrize("shape,window", [((1, 3, 256, 256), 11), ((3, 1, 37, 53), 11), ((2, 3, 7, 5), 11),
                                          ((2, 2, 64, 33), 3), ((1, 3, 40, 40), 15), ((2, 3, 33, 64), 1)])
def test_ssim_kernel_vs_torch_composition(vad, shape, window):
    """Ragged sizes (tiles cut by the image edge, images smaller than the window): forward against the stock torch
    composition on the CPU, gradient with respect to the prediction against its float64 autograd."""
    rng = np.random.default_rng(shape[2] * 131 + window)
    t = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    p = (t + torch.from_numpy(rng.normal(0, 0.2, shape).astype(np.float32))).clamp(-1, 1)
    crit = vad.CombinedLoss(alpha=0.4, window_size=window)
    crit.ssim.channels = shape[1]
    crit.ssim.window = vad.losses._gaussian_window(window, shape[1])
    ref = float(crit(p, t))                                     # CPU tensors -> torch composition
    ref_ssim = float(crit.ssim(p, t))
    with torch.no_grad():
        got = float(crit(p.cuda(), t.cuda()))
        got_ssim = float(crit.ssim(p.cuda(), t.cuda()))
    assert abs(got - ref) < 1e-5 * max(1.0, abs(ref)) and abs(got_ssim - ref_ssim) < 1e-5
    # gradient wanted for the prediction -> HIP forward + HIP backward (vad_ssim_mse_backward), against float64 autograd of
    # the composition on the CPU
    crit64 = vad.CombinedLoss(alpha=0.4, window_size=window).double()
    crit64.ssim.window = vad.losses._gaussian_window(window, shape[1]).double()
    p64 = p.double().requires_grad_(True)
    crit64(p64, t.double()).backward()
    s64 = p.double().requires_grad_(True)
    crit64.ssim(s64, t.double()).backward()
    nf, nb = vad.hip.calls.get("ssim", 0), vad.hip.calls.get("ssim_backward", 0)
    pg = p.cuda().requires_grad_(True)
    loss = crit(pg, t.cuda())
    (loss * 3.0).backward()                                   # a non-trivial upstream gradient
    sg = p.cuda().requires_grad_(True)
    crit.ssim(sg, t.cuda()).backward()
    assert vad.hip.calls["ssim"] == nf + 2 and vad.hip.calls["ssim_backward"] == nb + 2
    assert abs(float(loss.detach()) - ref) < 1e-5
    for got_g, want_g, scale in ((pg.grad.cpu() / 3.0, p64.grad, None), (sg.grad.cpu(), s64.grad, None)):
        mx = float(want_g.abs().max())
        assert float((got_g.double() - want_g).abs().max()) < 2e-4 * mx + 1e-12, f"gradient off by {float((got_g.double() - want_g).abs().max()):.3e} of {mx:.3e}"
    # a target that needs a gradient falls back to the torch composition (autograd handles both inputs)
    tg, pg2 = t.cuda().requires_grad_(True), p.cuda().requires_grad_(True)
    nf = vad.hip.calls["ssim"]
    crit(pg2, tg).backward()
    assert vad.hip.calls["ssim"] == nf and tg.grad is not None
    assert float((pg2.grad.cpu().double() - p64.grad).abs().max()) < 2e-4 * float(p64.grad.abs().max()) + 1e-12


def test_ssim_rejects_unsupported_window(vad):
    x = torch.zeros(1, 3, 32, 32, device="cuda")
    with torch.no_grad(), pytest.raises(vad.hip.VadError, match="window_size"):
        vad.SSIMLoss(window_size=17)(x, x)


# ------------------------------------------------------------------------------------------------ error contract of the C ABI
def test_c_abi_reports_errors_before_launching(vad):
    """include/vad_hip.h: every entry point returns VAD_OK or a negative VAD_ERR_* and vad_last_error() carries the text.
    Argument, workspace and configuration errors are detected on the host, before any kernel is launched."""
    l = vad.hip.lib()
    ERR_ARG, ERR_WS = -1, -3
    m, _ = _img_model(vad, 64, 5)
    x = vad.scoring.synth_frames_device(1, 0, 2, 64, 64)
    packed = m._packed(x.device)
    need = l.vad_img_workspace_bytes(2, 64, 64, 64)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    scores = torch.full((2,), float("nan"), device="cuda")
    s = vad.hip.current_stream()
    # too small a workspace
    rc = l.vad_img_score(x.data_ptr(), 2, 64, 64, 64, packed.data_ptr(), ws.data_ptr(), need - 1, 2, scores.data_ptr(), None, None, None, s)
    assert rc == ERR_WS and b"workspace" in l.vad_last_error()
    # null input, no output requested, frame size not a multiple of 16
    assert l.vad_img_score(None, 2, 64, 64, 64, packed.data_ptr(), ws.data_ptr(), need, 2, scores.data_ptr(), None, None, None, s) == ERR_ARG
    assert l.vad_img_score(x.data_ptr(), 2, 64, 64, 64, packed.data_ptr(), ws.data_ptr(), need, 2, None, None, None, None, s) == ERR_ARG
    assert l.vad_img_workspace_bytes(2, 60, 64, 64) == 0 and l.vad_img_packed_floats(1, 64) == 0
    torch.cuda.synchronize()
    assert torch.isnan(scores).all()                       # nothing ran
    # the same call with a correct workspace works
    assert l.vad_img_score(x.data_ptr(), 2, 64, 64, 64, packed.data_ptr(), ws.data_ptr(), need, 2, scores.data_ptr(), None, None, None, s) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(scores).all()
    # training entry point: workspace and configuration errors
    cfg = (32, 32, 1)
    n = l.vad_vid_train_nparams(*cfg)
    flat, grad, loss = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
    clips = torch.zeros(1, 2, 3, 32, 32, device="cuda")
    tb = l.vad_vid_train_workspace_bytes(1, 2, 32, 32, *cfg)
    tws = torch.empty(tb, dtype=torch.uint8, device="cuda")
    assert l.vad_vid_train_fwd_bwd(clips.data_ptr(), 1, 2, 32, 32, *cfg, flat.data_ptr(), grad.data_ptr(), None, tws.data_ptr(), tb - 1,
                                   0, loss.data_ptr(), None, s) == ERR_WS
    assert l.vad_vid_train_fwd_bwd(clips.data_ptr(), 1, 2, 24, 32, *cfg, flat.data_ptr(), grad.data_ptr(), None, tws.data_ptr(), tb,
                                   0, loss.data_ptr(), None, s) == ERR_ARG and b"multiples of 16" in l.vad_last_error()
    assert l.vad_vid_train_fwd_bwd(clips.data_ptr(), 1, 2, 32, 32, *cfg, flat.data_ptr(), grad.data_ptr(), None, tws.data_ptr(), tb,
                                   9, loss.data_ptr(), None, s) == ERR_ARG and b"precision" in l.vad_last_error()
    assert l.vad_adam_step(flat.data_ptr(), grad.data_ptr(), flat.data_ptr(), flat.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, s) == ERR_ARG   # step >= 1


# ------------------------------------------------------------------------------------------------ configs[3]: frame stream
def test_frame_stream_in_chunks_matches_oracle_on_a_strided_subset(vad):
    """BASELINE configs[3] at reduced scale (the full case is 100,000 frames over 8 ranks): a 2,100-frame stream at 256x256
    generated on the device in chunks of 512 and never materialised; parity on a strided subset regenerated on the CPU
    with the same counter-based generator, and bit-equality with scoring those frames directly."""
    m, st = _img_model(vad, 256, 9)
    seed, n = 0xC0FFEE + 3, 2100
    scores = vad.scoring.score_stream(m, seed, n, chunk=512)
    assert scores.shape == (n,) and torch.isfinite(scores).all()
    idx = list(range(5, n, 419))                                        # 5, 424, ..., crosses every chunk boundary region
    x = np.concatenate([vad.synth.frames(seed, i, 1, 3, 256, 256) for i in idx])
    ref = torch_oracle.img_scores({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, torch.from_numpy(x))
    assert rel_err(scores[idx].cpu().numpy(), ref["scores"].numpy()) < SCORE_RTOL
    with torch.no_grad():
        direct = m.get_reconstruction_error(torch.from_numpy(x).cuda())
    assert torch.equal(direct, scores[idx])
    # chunking of the stream is invisible in the result
    assert torch.equal(vad.scoring.score_stream(m, seed, n, chunk=300), scores)
