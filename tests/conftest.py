import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vad():
    """The product package (its directory name is not a Python identifier).  Builds the native library first if a
    fresh checkout has none (hipcc cross-compiles without a GPU; a no-op when the .so is up to date)."""
    pkg = importlib.import_module("video-anomaly-detection_amd")
    if not pkg.hip.LIB_PATH.exists():
        pkg.hip.build()
    return pkg


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


def synthetic_state_for(vad_pkg, module, seed):
    """Deterministic state dict (numpy) for one of our modules; same generator make_golden.py used."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    return vad_pkg.synth.synthetic_state(shapes, seed)


def load_synthetic(vad_pkg, module, seed):
    import torch
    st = synthetic_state_for(vad_pkg, module, seed)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=True)
    return st


IMG_GOLDENS = ["img_l32_32.npz", "img_l256_64.npz", "img_l100_32.npz", "img_c1_l24_32.npz", "img_c5_l32_32.npz"]
VID_GOLDENS = ["vid_default_64.npz", "vid_proj_32.npz", "vid_l3_32.npz", "vid_l48_h96_32.npz", "vid_l100_32.npz",
               "vid_c2_l32_h40_32.npz", "vid_c4_l32_32.npz"]


def in_channels_of(g) -> int:
    """Fixtures written before round 3 carry no `in_channels` entry: they are 3-channel models."""
    return int(g["in_channels"]) if "in_channels" in g.files else 3


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def auroc_slack(labels, ref_scores, rtol):
    """Largest AUROC change that score errors of `rtol` (relative) can cause: every (anomalous, normal) pair whose
    reference scores are closer than 2*rtol may swap order or tie.  With 64 scores inside one per-cent of each other some
    pairs differ by a single float32 ulp, so an exact AUROC match is not implied by matching scores; everything beyond these
    few pairs is."""
    labels = np.asarray(labels).astype(bool)
    s = np.asarray(ref_scores, dtype=np.float64)
    pos, neg = s[labels], s[~labels]
    close = np.abs(pos[:, None] - neg[None, :]) <= 2 * rtol * np.maximum(np.abs(pos[:, None]), np.abs(neg[None, :]))
    return float(close.sum()) / (len(pos) * len(neg)), int(close.sum())
