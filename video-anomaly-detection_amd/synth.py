"""Counter-based synthetic frames and weights (numpy side).

Everything here is a pure function of (seed, global element index), so any rank,
any batch split and the device kernel `vad_synth_frames` (csrc/synth.hip) produce
bit-identical data.  The value pipeline mirrors the reference's input transform
`ToTensor -> Normalize(0.5, 0.5)` (reference utils/dataset.py:65-70,
utils/video_dataset.py:62-66): u8 -> (u8/255 - 0.5)/0.5 in fp32.

Weights: the reference initialises with Xavier-normal / zero bias / identity BN
(models/autoencoder.py:170-179, models/video_autoencoder.py:318-327).  An identity
BN hides BN bugs, so the synthetic state dict keeps the Xavier *scale* but draws
every tensor (biases and BN buffers included) from the hash stream.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

PATCH = 32  # default side of the saturated "defect" patch used for the AUROC check


def patch_side(anomalies) -> int:
    """`anomalies` argument -> patch side: False/0 = no patches, True/1 = PATCH, k >= 2 = k pixels (a small patch moves
    a frame's score by about one standard deviation of the normal scores, so the AUROC is not trivially 1)."""
    k = int(anomalies)
    return PATCH if k == 1 else k


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _stream(seed: int, start: int, count: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _GOLDEN
        idx = np.arange(start, start + count, dtype=np.uint64) + base
    return mix64(idx)


def u8_to_unit(u8: np.ndarray) -> np.ndarray:
    """(u8/255 - 0.5)/0.5 evaluated in fp32, step by step like the reference transform."""
    f = u8.astype(np.float32)
    f = f / np.float32(255.0)
    f = f - np.float32(0.5)
    return f / np.float32(0.5)


def frame_label(seed: int, frame_idx: np.ndarray) -> np.ndarray:
    """0 = normal, 1 = anomalous; top bit of a second hash stream."""
    with np.errstate(over="ignore"):
        z = np.asarray(frame_idx, dtype=np.uint64) + np.uint64((seed ^ 0x5DEECE66D) & 0xFFFFFFFFFFFFFFFF) * _GOLDEN
    return (mix64(z) >> np.uint64(63)).astype(np.int64)


def _patch_origin(seed: int, frame_idx: int, h: int, w: int, side: int = PATCH):
    with np.errstate(over="ignore"):
        z = mix64(np.uint64(frame_idx) + np.uint64((seed ^ 0xB5297A4D) & 0xFFFFFFFFFFFFFFFF) * _GOLDEN)
    z = int(z)
    py = (z & 0xFFFF) % max(h - side + 1, 1)
    px = ((z >> 16) & 0xFFFF) % max(w - side + 1, 1)
    return py, px


def frames_u8(seed: int, first_frame: int, n: int, c: int, h: int, w: int, anomalies=False) -> np.ndarray:
    """uint8 [n, c, h, w]; element (f, ch, y, x) = hash(seed, ((f*c+ch)*h+y)*w+x) >> 56.  `anomalies`: see patch_side."""
    per = c * h * w
    out = (_stream(seed, first_frame * per, n * per) >> np.uint64(56)).astype(np.uint8).reshape(n, c, h, w)
    side = patch_side(anomalies)
    if side:
        lab = frame_label(seed, np.arange(first_frame, first_frame + n))
        for i in range(n):
            if lab[i]:
                py, px = _patch_origin(seed, first_frame + i, h, w, side)
                out[i, :, py:py + side, px:px + side] = 255
    return out


def frames(seed: int, first_frame: int, n: int, c: int = 3, h: int = 256, w: int = 256, anomalies=False) -> np.ndarray:
    """fp32 NCHW frames in [-1, 1]."""
    return u8_to_unit(frames_u8(seed, first_frame, n, c, h, w, anomalies))


def clips(seed: int, first_clip: int, n: int, t: int, c: int = 3, h: int = 256, w: int = 256) -> np.ndarray:
    """fp32 [n, t, c, h, w]; clip k is frames [k*t, (k+1)*t) of the same stream."""
    return frames(seed, first_clip * t, n * t, c, h, w).reshape(n, t, c, h, w)


# --------------------------------------------------------------------------- weights

def _uniform(seed: int, salt: int, count: int) -> np.ndarray:
    """fp64 uniform [0,1) from the top 53 bits of the stream (seed, salt)."""
    z = _stream(seed * 1000003 + salt, 0, count)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synthetic_state(shapes: "dict[str, tuple]", seed: int) -> "dict[str, np.ndarray]":
    """Deterministic state dict for a {key: shape} description.

    conv / convT `.weight` (4-D): uniform with the Xavier-normal standard deviation
    sqrt(2/(fan_in+fan_out)); 1-D keys by suffix: BN `.weight` U(0.5,1.5), `.bias`
    U(-0.1,0.1), `.running_mean` U(-0.2,0.2), `.running_var` U(0.25,1.75),
    `.num_batches_tracked` = 1.
    """
    out = {}
    for salt, (key, shape) in enumerate(shapes.items()):
        n = int(np.prod(shape)) if len(shape) else 1
        if key.endswith("num_batches_tracked"):
            out[key] = np.array(1, dtype=np.int64)
            continue
        u = _uniform(seed, salt, n)
        if len(shape) == 4:
            rf = shape[2] * shape[3]
            std = np.sqrt(2.0 / ((shape[0] + shape[1]) * rf))
            v = (u * 2.0 - 1.0) * np.sqrt(3.0) * std
        elif key.endswith("running_var"):
            v = 0.25 + 1.5 * u
        elif key.endswith("running_mean"):
            v = (u - 0.5) * 0.4
        elif key.endswith(".weight"):
            v = 0.5 + u
        else:  # any bias
            v = (u - 0.5) * 0.2
        out[key] = v.astype(np.float32).reshape(shape)
    return out
