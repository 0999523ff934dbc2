"""Scoring loops around the models: the build's counterpart of the reference's evaluate drivers.

* `compute_auroc` mirrors `evaluate.compute_auroc(model, test_loader, device)` (reference
  evaluate.py:46-91): same arguments, same 4-tuple result.
* `score_clips` mirrors the clip loop of `evaluate_video.evaluate` (reference evaluate_video.py:138-154).
* `sharded_scores` is new (the reference is single-process): contiguous block partition of the
  frame / clip stream over ranks, no data-path collective, ONE all_gather of the score vector
  (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Tuple

import numpy as np
import torch

from . import hip


def roc_auc(labels, scores) -> float:
    """Area under the ROC curve (Mann-Whitney U with average ranks for ties): the statistic
    sklearn.metrics.roc_auc_score returns for binary labels (reference evaluate.py:74)."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("roc_auc needs both classes")
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    return float((r[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def compute_auroc(model, test_loader: Iterable[dict], device):
    """Same contract as the reference's evaluate.compute_auroc (evaluate.py:46-91): iterate batches
    {'image', 'label', 'defect_type'}, score with get_reconstruction_error(per_pixel=False), return
    (auroc, labels, scores, per-defect {count, mean_score, is_anomaly})."""
    all_labels, all_scores, all_types = [], [], []
    with torch.no_grad():
        for batch in test_loader:
            images = batch["image"].to(device)
            scores = model.get_reconstruction_error(images, per_pixel=False)
            all_scores.extend(scores.cpu().numpy())
            all_labels.extend(np.asarray(batch["label"]))
            all_types.extend(batch["defect_type"])
    labels = np.array(all_labels)
    scores = np.array(all_scores)
    auroc = roc_auc(labels, scores)
    per_defect = {}
    for name in set(all_types):
        m = np.array([d == name for d in all_types])
        per_defect[name] = {"count": int(m.sum()), "mean_score": scores[m].mean(),
                            "is_anomaly": labels[m][0] if m.any() else 0}
    return auroc, labels, scores, per_defect


def score_clips(model, loader: Iterable[dict], device, per_frame: bool = False):
    """Clip loop of the reference's evaluate_video.evaluate (evaluate_video.py:137-154): returns
    (clip scores float32[N], labels) and, when per_frame, also float32[N,T] frame scores computed in
    the SAME pass (the reference runs a second forward, evaluate_video.py:147-149).  Only the two score
    vectors are produced: no reconstruction or error map is written."""
    seq, frm, labels = [], [], []
    with torch.no_grad():
        for batch in loader:
            frames = batch["frames"].to(device)
            if per_frame:
                out = model.score_seq_and_frames(frames)
                seq.extend(out["seq"].cpu().numpy())
                frm.extend(out["frame"].cpu().numpy())
            else:
                seq.extend(model.get_reconstruction_error(frames, per_frame=False).cpu().numpy())
            labels.extend(np.asarray(batch["label"]))
    if per_frame:
        return np.array(seq), np.array(labels), np.array(frm)
    return np.array(seq), np.array(labels)


# ------------------------------------------------------------------------------ device synth
def synth_frames_device(seed: int, first_frame: int, n: int, h: int = 256, w: int = 256, c: int = 3,
                        device="cuda", anomalies: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NCHW fp32 frames generated on the GPU, bit-identical to `synth.frames` (numpy)."""
    if out is None:
        out = torch.empty(n, c, h, w, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        hip.check(hip.lib().vad_synth_frames(out.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, first_frame, n, c, h, w,
                                             int(anomalies), hip.current_stream()), "vad_synth_frames")
    return out


# ------------------------------------------------------------------------------ multi-GPU
def block_partition(n_items: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Contiguous block partition: rank r owns [r*per, min((r+1)*per, n)); returns (start, count, per).
    Clips are items, so a clip is never split across ranks (the ConvLSTM recurrence is sequential in t)."""
    per = -(-n_items // world)
    start = min(rank * per, n_items)
    return start, max(0, min(per, n_items - start)), per


def sharded_scores(score_block: Callable[[int, int], torch.Tensor], n_items: int, width: int = 1,
                   rank: int = 0, world: int = 1, device="cpu", group=None, force_collective: bool = False) -> torch.Tensor:
    """Score items [0, n_items) across `world` ranks and return float32[n_items, width] (squeezed to
    [n_items] when width == 1) in the original order on EVERY rank.

    `score_block(first, count)` returns this rank's scores for items [first, first+count) as a float32
    tensor [count] or [count, width] on `device`.  The only communication is one all_gather of
    `per * width` floats per rank; the tail of the last block is zero padding that is cut off after the
    gather, so rank-major order == original order.  `force_collective` runs the all_gather for world == 1 too (a
    one-rank process group: how the RCCL branch is exercised on a one-GPU box).
    """
    start, count, per = block_partition(n_items, world, rank)
    local = torch.zeros(per, width, dtype=torch.float32, device=device)
    if count:
        local[:count] = score_block(start, count).reshape(count, width).to(torch.float32)
    if world > 1 or force_collective:
        import torch.distributed as dist
        gathered = torch.empty(world * per, width, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(gathered, local, group=group)
    else:
        gathered = local
    out = gathered[:n_items]
    return out[:, 0] if width == 1 else out


def score_stream(model, seed: int, n_frames: int, chunk: int = 512, h: int = 256, w: int = 256, rank: int = 0,
                 world: int = 1, device="cuda", anomalies: bool = False, group=None, force_collective: bool = False) -> torch.Tensor:
    """BASELINE configs[3]: score a long synthetic frame stream without ever materialising it.  The stream is
    block-partitioned over ranks; each rank regenerates its frames on the device `chunk` at a time (the counter-based
    generator makes any sub-range reproducible, on any rank and on the CPU), scores them with
    `model.get_reconstruction_error`, and ONE all_gather returns float32[n_frames] in stream order on every rank."""
    dev = torch.device(device)
    buf = torch.empty(min(chunk, max(n_frames, 1)), 3, h, w, dtype=torch.float32, device=dev)

    def score_block(first: int, count: int) -> torch.Tensor:
        out = torch.empty(count, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for s in range(0, count, buf.shape[0]):
                k = min(buf.shape[0], count - s)
                synth_frames_device(seed, first + s, k, h, w, 3, dev, anomalies, out=buf[:k])
                out[s:s + k] = model.get_reconstruction_error(buf[:k])
        return out

    return sharded_scores(score_block, n_frames, 1, rank, world, dev, group, force_collective)
