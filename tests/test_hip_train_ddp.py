"""Data-parallel semantics of the native training step on a GPU (row f-1, SURVEY.md section 8e "Training"): two ranks
share the one device of the test box over gloo (RCCL needs one device per rank; the collective is the same single
all-reduce of the flat gradient buffer), each steps on its half of the batch.  Both must end with identical parameters,
equal bit for bit to ONE process that averages the two halves' gradients and takes one Adam step.  BatchNorm statistics
stay per rank (the reference has no SyncBN)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(latent=64, layers=2, b=4, t=3, hw=32, wseed=51, xseed=151, lr=1e-4, wd=1e-5)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(vad):
    from conftest import load_synthetic
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=CFG["latent"], lstm_hidden_dim=CFG["latent"], lstm_num_layers=CFG["layers"])
    load_synthetic(vad, m, CFG["wseed"])
    return m.cuda()


def _clips(vad):
    return torch.from_numpy(vad.synth.clips(CFG["xseed"], 0, CFG["b"], CFG["t"], 3, CFG["hw"], CFG["hw"]))


def _rank_unseeded(rank, world, port, out_dir):
    """Replicas constructed independently (the reference's train_video.py sets no seed): the trainer must broadcast rank
    0's parameters and BatchNorm buffers when it is built, as DistributedDataParallel does."""
    import importlib
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import torch.distributed as dist
    from conftest import load_synthetic
    vad = importlib.import_module("video-anomaly-detection_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=32, lstm_hidden_dim=32, lstm_num_layers=1)
    load_synthetic(vad, m, 900 + rank)                     # DIFFERENT weights and BatchNorm statistics on every rank
    tr = vad.VideoTrainer(m.cuda())
    np.save(os.path.join(out_dir, f"u{rank}.npy"), np.concatenate([tr.flat.cpu().numpy(), tr.running.cpu().numpy()]))
    x = torch.from_numpy(vad.synth.clips(5, rank, 1, 2, 3, 32, 32)).cuda()
    tr.step(x)
    np.save(os.path.join(out_dir, f"v{rank}.npy"), tr.flat.cpu().numpy())
    moved = m.float()                                      # re-allocation behind the trainer's back must be caught ...
    moved.encoder.encoder[0].weight.data = moved.encoder.encoder[0].weight.data.clone()
    try:
        tr.step(x)
        ok = False
    except vad.hip.VadError as e:
        ok = "no longer aliases" in str(e)
    np.save(os.path.join(out_dir, f"w{rank}.npy"), np.array([ok]))
    dist.destroy_process_group()


def test_trainer_broadcasts_rank0_state_and_checks_aliasing(vad, tmp_path):
    world = 2
    mp.spawn(_rank_unseeded, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    u0, u1 = np.load(tmp_path / "u0.npy"), np.load(tmp_path / "u1.npy")
    assert np.array_equal(u0, u1), "replicas did not start from rank 0's parameters / BatchNorm buffers"
    assert np.array_equal(np.load(tmp_path / "v0.npy"), np.load(tmp_path / "v1.npy")), "ranks diverged after one step"
    assert np.load(tmp_path / "w0.npy")[0] and np.load(tmp_path / "w1.npy")[0], "a re-allocated parameter was not detected"


def _rank(rank, world, port, out_dir):
    import importlib
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import torch.distributed as dist
    vad = importlib.import_module("video-anomaly-detection_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = vad.VideoTrainer(_build(vad), lr=CFG["lr"], weight_decay=CFG["wd"])
    per = CFG["b"] // world
    x = _clips(vad)[rank * per:(rank + 1) * per].cuda()
    losses = [float(tr.step(x)) for _ in range(2)]
    np.save(os.path.join(out_dir, f"p{rank}.npy"), tr.flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.array(losses))
    np.save(os.path.join(out_dir, f"s{rank}.npy"), tr.running.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_equal_one_process_with_averaged_gradients(vad, tmp_path, world):
    """World 2: bit-equal to one process that sums the two halves' gradients (a two-term sum has one order).  World 4 (one
    clip per rank; the most ranks a GPU box admits beside the test session): every rank ends bit-identical to every other,
    and equal to the one-process result up to the summation order of the four-term all-reduce."""
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p0 = np.load(tmp_path / "p0.npy")
    for r in range(1, world):
        assert np.array_equal(p0, np.load(tmp_path / f"p{r}.npy")), f"rank {r} diverged from rank 0"
    # one process: per-half gradients from two replicas, averaged, one optimiser step; twice
    x = _clips(vad).cuda()
    per = CFG["b"] // world
    reps = [vad.VideoTrainer(_build(vad), lr=CFG["lr"], weight_decay=CFG["wd"]) for _ in range(world)]
    for _ in range(2):
        losses = [float(tr.forward_backward(x[r * per:(r + 1) * per])[0]) for r, tr in enumerate(reps)]
        total = reps[0].grad.clone()
        for tr in reps[1:]:
            total += tr.grad
        for tr in reps:
            tr.grad.copy_(total)
            tr.optimizer_step(1.0 / world)
    one = reps[0].flat.cpu().numpy()
    if world == 2:
        assert np.array_equal(one, p0), f"max diff {np.abs(one - p0).max():.3e}"
    else:
        # Adam normalises every entry by its own gradient history (update = lr * m / (sqrt(v) + eps), |update| <= ~lr): where a
        # gradient entry is ~0 the last bits of the four-term sum decide the update's size, so the comparison is "almost every
        # entry to 1e-6, no entry further than the two steps could move it" rather than a uniform bound
        diff = np.abs(one - p0)
        assert diff.max() <= 2 * 2 * CFG["lr"], f"max diff {diff.max():.3e}"
        assert np.mean(diff <= 1e-6 * np.abs(one).max()) > 0.99, f"{np.mean(diff <= 1e-6 * np.abs(one).max()):.4f} of the entries agree"
    for r in range(world):
        srun, want = np.load(tmp_path / f"s{r}.npy"), reps[r].running.cpu().numpy()                 # per-rank BatchNorm statistics
        if world == 2:
            assert abs(np.load(tmp_path / f"l{r}.npy")[-1] - losses[r]) < 1e-7 * losses[r]
            assert np.array_equal(srun, want)
        else:     # the second step starts from parameters that differ in the last bits (summation order of the first all-reduce)
            assert abs(np.load(tmp_path / f"l{r}.npy")[-1] - losses[r]) < 1e-5 * losses[r]
            assert np.allclose(srun, want, rtol=1e-4, atol=1e-6)
    assert not np.array_equal(np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy"))
