"""ctypes wrapper of oracle/libvad_oracle.so (vad_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "libvad_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = _DIR / "vad_oracle.c"
    if force or not _LIB.exists() or _LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-B", "libvad_oracle.so"], check=True, capture_output=True)
    return _LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
    return _lib


def _params(state: dict):
    arrs = [np.ascontiguousarray(np.asarray(v, dtype=np.float32)) for k, v in state.items()
            if not k.endswith("num_batches_tracked")]
    return arrs, (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def img_forward(state: dict, latent: int, x: np.ndarray, want_latent: bool = False):
    """-> recon [N,3,H,W] (and latent [N,latent,H/16,W/16])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, _, h, w = x.shape
    arrs, pp = _params(state)
    recon = np.empty_like(x)
    lat = np.empty((n, latent, h // 16, w // 16), np.float32) if want_latent else None
    rc = lib().vo_img_forward(pp, C.c_int(latent), _p(x), n, h, w, _p(recon), _p(lat))
    assert rc == 0
    return (recon, lat) if want_latent else recon


def error(x: np.ndarray, recon: np.ndarray, t: int = 1):
    """-> (errmap [N,H,W], frame_scores [N], seq_scores [N/t]) following the reference's means."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    recon = np.ascontiguousarray(recon, dtype=np.float32)
    n, c, h, w = x.shape
    emap = np.empty((n, h, w), np.float32)
    frame = np.empty(n, np.float32)
    seq = np.empty(n // t, np.float32)
    lib().vo_error(_p(x), _p(recon), n, c, h, w, t, _p(emap), _p(frame), _p(seq))
    return emap, frame, seq


def img_scores(state: dict, latent: int, x: np.ndarray):
    recon = img_forward(state, latent, x)
    emap, frame, _ = error(x, recon)
    return {"recon": recon, "errmap": emap[:, None], "scores": frame}


def vid_forward(state: dict, latent: int, hid: int, layers: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    b, t, _, h, w = x.shape
    arrs, pp = _params(state)
    recon = np.empty_like(x)
    rc = lib().vo_vid_forward(pp, C.c_int(latent), C.c_int(hid), C.c_int(layers), _p(x), b, t, h, w, _p(recon))
    assert rc == 0
    return recon


def vid_scores(state: dict, latent: int, hid: int, layers: int, x: np.ndarray):
    b, t, c, h, w = x.shape
    recon = vid_forward(state, latent, hid, layers, x)
    emap, frame, seq = error(x.reshape(b * t, c, h, w), recon.reshape(b * t, c, h, w), t)
    return {"recon": recon, "errmap": emap.reshape(b, t, 1, h, w), "frame": frame.reshape(b, t), "seq": seq}


def conv2d(x, w, b, k=3):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    n, cin, h, ww = x.shape
    cout = w.shape[0]
    y = np.empty((n, cout, h, ww), np.float32)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    lib().vo_conv2d(_p(x), _p(w), _p(bb), _p(y), n, cin, h, ww, cout, k)
    return y


def convt2x2(x, w, b):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    n, cin, h, ww = x.shape
    cout = w.shape[1]
    y = np.empty((n, cout, 2 * h, 2 * ww), np.float32)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    lib().vo_convt2x2(_p(x), _p(w), _p(bb), _p(y), n, cin, h, ww, cout)
    return y


def batchnorm_eval(x, g, b, m, v):
    y = np.array(x, dtype=np.float32, order="C")
    n, c, h, w = y.shape
    arrs = [np.ascontiguousarray(a, np.float32) for a in (g, b, m, v)]
    lib().vo_batchnorm_eval(_p(y), *[_p(a) for a in arrs], n, c, h * w)
    return y


def maxpool2(x):
    x = np.ascontiguousarray(x, np.float32)
    n, c, h, w = x.shape
    y = np.empty((n, c, h // 2, w // 2), np.float32)
    lib().vo_maxpool2(_p(x), _p(y), n, c, h, w)
    return y


def convlstm_cell(x, h, c, w, b):
    """-> (h', c') for one step; inputs are not modified."""
    x = np.ascontiguousarray(x, np.float32)
    h2 = np.array(h, dtype=np.float32, order="C"); c2 = np.array(c, dtype=np.float32, order="C")
    w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
    n, cx, hh, ww = x.shape
    lib().vo_convlstm_cell(_p(x), _p(h2), _p(c2), _p(w), _p(b), n, cx, h2.shape[1], hh, ww)
    return h2, c2
