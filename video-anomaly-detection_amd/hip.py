"""ctypes binding of libvad_hip.so (the C ABI declared in include/vad_hip.h).

This is the reference-side stub a maintainer would add: raw device pointers
from `tensor.data_ptr()`, the current HIP stream handle, plain ints.  There is NO fallback: if the
library is missing, `lib()` raises, and every scoring entry point of the package goes through it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = Path(os.environ.get("VAD_LIB", PKG_DIR / "libvad_hip.so"))
SOURCES = ["conv_mfma.hip", "conv_wino.hip", "dec4_fused.hip", "tail.hip", "wide_io.hip", "ssim.hip", "train_ops.hip", "train_step.hip", "train_step_img.hip", "vad_api.hip", "pack.cpp"]

VAD_OK = 0
ABI_VERSION = 3
MAX_IN_CHANNELS = 32    # include/vad_hip.h VAD_MAX_IN_CHANNELS
PREC_FP32, PREC_SPLIT, PREC_BF16, PREC_BF16S, PREC_WINO = 0, 1, 2, 3, 4
# bf16 modes: training entry points only.  "bf16_operands" = bf16 MFMA operands converted from fp32 tensors; "bf16_tensors" =
# the activation / gradient tensors themselves are bf16 in HBM (VAD_PREC_BF16S, video training step only)
# "bf16" is an alias of "bf16_tensors" EVERYWHERE (BASELINE.json configs[4]'s dtype); a trainer without a bf16-tensor form
# (ImageTrainer) rejects it by name instead of quietly running other arithmetic under the same string.
# "winograd" (scoring only, opt-in): the 3x3 convolutions behind the first layer as Winograd F(2x2,3x3) on the exact-fp32 MFMA
PRECISIONS = {"fp32": PREC_FP32, "split": PREC_SPLIT, "bf16": PREC_BF16S, "bf16_operands": PREC_BF16, "bf16_tensors": PREC_BF16S,
              "winograd": PREC_WINO}


def training_precision(name: str, trainer: str, tensors: bool) -> str:
    """The ONE place the trainers' `precision=` strings are resolved: returns 'fp32', 'split', 'bf16_operands' or
    'bf16_tensors' ('bf16' -> 'bf16_tensors').  `tensors`: does this trainer have the bf16-tensor form?"""
    # "winograd" (both trainers): fp32 everywhere, the 3x3 convolutions (forward + data gradients) as Winograd F(2x2,3x3)
    ok = ("fp32", "split", "bf16_operands", "winograd") + (("bf16", "bf16_tensors") if tensors else ())
    if name in ("bf16", "bf16_tensors") and not tensors:
        raise VadError(f"{trainer}: precision {name!r} means bf16 TENSORS (activations / gradients bf16 in HBM), which this "
                       f"trainer does not implement; its bf16 form is 'bf16_operands' (fp32 tensors, bf16 MFMA operands)")
    if name not in ok:
        raise VadError(f"{trainer}: precision must be one of {ok}, got {name!r}")
    return "bf16_tensors" if name == "bf16" else name
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
X_F32_NCHW, X_U8_NHWC = 0, 1
PROF_SLOTS = 32
MAX_WIDTH = 4096          # VAD_MAX_WIDTH: largest latent_dim / lstm_hidden_dim

_lock = threading.Lock()
_lib = None
calls = {"img_score": 0, "vid_score": 0}   # tests assert the native path really ran


class VadError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile libvad_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU): one object per source, compiled
    in parallel and re-used while it is newer than its source and every header, then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [CSRC / s for s in SOURCES]
    hdrs = sorted(CSRC.glob("*.h")) + [REPO_DIR / "include" / "vad_hip.h"]
    hdr_time = max(h.stat().st_mtime for h in hdrs)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", f"-I{REPO_DIR / 'include'}", f"-I{CSRC}"]

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.name + ".o")
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_time):
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, srcs))
    if force or not LIB_PATH.exists() or any(LIB_PATH.stat().st_mtime < o.stat().st_mtime for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(LIB_PATH)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


def source_digest() -> str:
    """sha256 over the library's sources (csrc/*, include/vad_hip.h), in name order: what a committed counter file
    (profiles/r*_pmc_traffic_*.json, written by tools/pmc_traffic_summary.py) was measured on.  `bench.py` prints a committed
    `roofline.traffic` only while this still matches, so the figure cannot go stale behind a kernel change."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(CSRC.iterdir()) + [REPO_DIR / "include" / "vad_hip.h"]:
        if f.suffix in (".hip", ".h", ".cpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p
_ll = C.c_longlong
_i = C.c_int
_sz = C.c_size_t
_f = C.c_float

# name -> (restype, argtypes); every symbol of include/vad_hip.h
SIGNATURES = {
    "vad_abi_version": (_i, []),
    "vad_last_error": (C.c_char_p, []),
    "vad_pack_conv3x3_floats": (_sz, [_i, _i]),
    "vad_pack_conv3x3": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "vad_pack_conv3x3_c3_floats": (_sz, [_i]),
    "vad_pack_conv3x3_c3": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "vad_pack_convt2x2_floats": (_sz, [_i, _i]),
    "vad_pack_convt2x2": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "vad_pack_conv1x1_floats": (_sz, [_i, _i]),
    "vad_pack_conv1x1": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "vad_pack_conv3x3_to3_floats": (_sz, [_i]),
    "vad_pack_conv3x3_to3": (_i, [_vp, _i, _vp]),
    "vad_conv3x3_c3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_conv3x3_c3_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_conv3x3": (_i, [_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_convt2x2": (_i, [_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_pack_conv3x3_wino_floats": (_sz, [_i, _i]),
    "vad_pack_conv3x3_wino": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "vad_conv3x3_wino": (_i, [_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_convlstm_step_wino": (_i, [_vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vad_conv1x1": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "vad_convlstm_step": (_i, [_vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_score_partials": (_i, [_i, _i, _i]),
    "vad_conv3x3_to3_score": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_convt2x2_to3_score": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_score_finalize": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "vad_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_ssim_workspace_floats": (_sz, [_ll, _i, _i]),
    "vad_ssim_mse": (_i, [_vp, _vp, _ll, _i, _i, _i, C.c_float, _vp, _vp, _vp]),
    "vad_ssim_grad_workspace_floats": (_sz, [_ll, _i, _i]),
    "vad_ssim_mse_backward": (_i, [_vp, _vp, _ll, _i, _i, _i, C.c_float, _vp, _vp, _vp, _vp]),
    "vad_chan_ws_floats": (_sz, [_ll, _i]),
    "vad_bn_stats": (_i, [_vp, _ll, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "vad_chan_sum": (_i, [_vp, _ll, _i, _vp, _vp, _vp]),
    "vad_bn_act_pool_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_bn_act_pool_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp,
                                 _i, _i, _i, _i, _i, _i, _vp]),
    "vad_lstm_gates_fwd": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _vp, _ll, _i, _i, _i, _i, _vp]),
    "vad_lstm_gates_bwd": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _vp, _ll, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vad_conv_wgrad_ws_floats": (_sz, [_i, _i, _i, _i, _i]),
    "vad_conv_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_conv_c3_wgrad_ws_floats": (_sz, [_i, _i, _i]),
    "vad_conv_c3_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_convt_to3_mse_ws_floats": (_sz, [_i, _i, _i]),
    "vad_convt_to3_mse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vad_chan_sum_t": (_i, [_vp, _i, _ll, _i, _vp, _vp, _vp]),
    "vad_bn_act_pool_fwd_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_bn_act_pool_bwd_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp,
                                   _i, _i, _i, _i, _i, _i, _vp]),
    "vad_lstm_gates_fwd_t": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _i, _vp, _ll, _i, _i, _i, _i, _vp]),
    "vad_lstm_gates_bwd_t": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _i, _vp, _ll, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vad_conv_c3_wgrad_t": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_convt_to3_mse_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "vad_scale_floats": (_i, [_vp, _ll, _f, _vp]),
    "vad_train_pack_conv1x1_p": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp]),
    "vad_conv1x1_p": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "vad_conv3x3_c3_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_conv3x3_c3_bf16op": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_debug_set_wgrad_pairs": (_i, [_i]),
    "vad_debug_set_wgrad_split": (_i, [_i]),
    "vad_debug_set_c3_routed": (_i, [_i]),
    "vad_c3_routed_enabled": (_i, []),
    "vad_conv_c3_wgrad_routed_ok": (_i, [_i, _i, _i]),
    "vad_conv_c3_wgrad_routed_ws_floats": (_sz, [_i, _i]),
    "vad_bn_act_pool_bwd_codes_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vad_conv_c3_wgrad_routed": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vad_debug_set_wgrad_ring_f32": (_i, [_i]),
    "vad_debug_set_bn_wide": (_i, [_i]),
    "vad_adam_step": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "vad_train_pack_conv3x3": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp]),
    "vad_train_pack_convt2x2": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp]),
    "vad_train_pack_conv3x3_c3": (_i, [_vp, _i, _vp, _vp]),
    "vad_train_pack_conv1x1": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "vad_train_pack_conv3x3_to3": (_i, [_vp, _i, _vp, _vp, _vp]),
    "vad_conv3x3_to3_bwd_ws_floats": (_sz, [_i, _i, _i, _i]),
    "vad_conv3x3_to3_tanh_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "vad_vid_train_nparams": (_sz, [_i, _i, _i]),
    "vad_vid_train_nstats": (_sz, [_i, _i, _i]),
    "vad_vid_train_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "vad_debug_set_train_decisions": (_i, [_vp, _sz]),
    "vad_debug_train_decisions_used": (_sz, []),
    "vad_debug_set_train_stop": (_i, [_i]),
    "vad_debug_set_split_grad_scale": (_i, [_i]),
    "vad_split_grad_scale_enabled": (_i, []),
    "vad_vid_train_debug_layout": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "vad_vid_train_fwd_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    "vad_img_train_nparams": (_sz, [_i]),
    "vad_img_train_nstats": (_sz, [_i]),
    "vad_img_train_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "vad_img_train_fwd_bwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _i, _f, _i, _i, _vp, _vp, _vp]),
    "vad_synth_frames": (_i, [_vp, C.c_ulonglong, _ll, _ll, _i, _i, _i, _i, _vp]),
    "vad_img_packed_floats": (_sz, [_i, _i]),
    "vad_img_pack": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "vad_blob_precision": (_i, [_vp]),
    "vad_img_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "vad_img_score": (_i, [_vp, _ll, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_nparams": (_i, [_i, _i]),
    "vad_vid_packed_floats": (_sz, [_i, _i, _i]),
    "vad_vid_pack": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "vad_vid_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "vad_vid_score": (_i, [_vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_debug_set_conv_variant": (_i, [_i]),
    "vad_debug_set_tail_group": (_i, [_i]),
    "vad_debug_set_dec4_fused": (_i, [_i]),
    "vad_debug_set_dec4_band": (_i, [_i]),
    "vad_dec4_score_partials": (_i, [_i, _i]),
    "vad_dec4_score": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vad_debug_set_lstm_wavefront": (_i, [_i]),
    "vad_img_score_x": (_i, [_vp, _i, _i, _ll, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_score_x": (_i, [_vp, _i, _i, _ll, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_score_windows_x": (_i, [_vp, _i, _i, _ll, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_num_windows": (_ll, [_ll, _i, _i]),
    "vad_vid_windows_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    # in_channels > 3 (the `_c` forms take in_ch; in_ch == 3 is the form above)
    "vad_img_workspace_bytes_c": (_sz, [_i, _i, _i, _i, _i]),
    "vad_img_score_c": (_i, [_vp, _i, _i, _i, _ll, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_packed_floats_c": (_sz, [_i, _i, _i, _i]),
    "vad_vid_pack_c": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vad_vid_workspace_bytes_c": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "vad_vid_score_c": (_i, [_vp, _i, _i, _i, _ll, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_windows_workspace_bytes_c": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "vad_vid_score_windows_c": (_i, [_vp, _i, _i, _i, _ll, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_vid_score_windows": (_i, [_vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "vad_graph_begin": (_i, [_vp]),
    "vad_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "vad_graph_launch": (_i, [_vp, _vp]),
    "vad_graph_destroy": (_i, [_vp]),
    "vad_prof_enable": (_i, [_i]),
    "vad_prof_reset": (_i, []),
    "vad_prof_read": (_i, [_vp, _vp]),
    "vad_prof_slot_name": (C.c_char_p, [_i, _i]),
}


def lib() -> C.CDLL:
    """Load the library (once).  Raises VadError if it has not been built: no silent fallback."""
    global _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise VadError(
                    f"{LIB_PATH} is missing: the HIP scoring path has no fallback. "
                    "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
            l = C.CDLL(str(LIB_PATH))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(l, name)
                fn.restype, fn.argtypes = res, args
            if l.vad_abi_version() != ABI_VERSION:
                raise VadError("libvad_hip.so ABI version mismatch")
            _lib = l
        return _lib


def precision_mode(name: str) -> int:
    """'fp32' / 'split' -> VAD_PREC_*: the arithmetic mode is an argument of every packer and launcher (ABI 2)."""
    if name not in PRECISIONS:
        raise VadError(f"precision must be one of {sorted(PRECISIONS)}, got {name!r}")
    return PRECISIONS[name]


def check(rc: int, what: str = "") -> None:
    if rc != VAD_OK:
        msg = lib().vad_last_error().decode("utf-8", "replace")
        raise VadError(f"{what or 'libvad_hip'} failed (code {rc}): {msg}")


class CapturedCall:
    """One scoring call captured into a hipGraph (vad_graph_*): `replay()` relaunches it on the current stream with the
    same buffers; `outputs` are the tensors it writes.  Built by `model.capture(...)`."""

    def __init__(self, run, inputs, outputs, keep, post=None):
        """`inputs` = the tensor `replay(x)` copies into (a view of the captured input buffer shaped like the caller's
        batch); `outputs` = the tensors the captured kernels write; `post` (optional) maps them to what `replay` returns
        (1- / 2-channel models: the kernels work on 3 planes, the caller sees `in_channels`)."""
        import torch
        self.inputs, self.outputs, self._keep, self._post = inputs, outputs, keep, post
        self._exec = _vp()
        l = lib()
        side = torch.cuda.Stream()                       # capture on a side stream, never on the legacy default stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            check(l.vad_graph_begin(side.cuda_stream), "vad_graph_begin")
            try:
                run()
            finally:
                rc = l.vad_graph_end(side.cuda_stream, C.byref(self._exec))
            check(rc, "vad_graph_end")
        torch.cuda.current_stream().wait_stream(side)
        calls["graph_capture"] = calls.get("graph_capture", 0) + 1

    def replay(self, x=None):
        """Relaunch; with `x` given its contents are first copied into the captured input buffer."""
        if x is not None:
            self.inputs.copy_(x)
        check(lib().vad_graph_launch(self._exec, current_stream()), "vad_graph_launch")
        calls["graph_replay"] = calls.get("graph_replay", 0) + 1
        return self._post(self.outputs) if self._post else self.outputs

    __call__ = replay

    def __del__(self):
        try:
            if self._exec:
                lib().vad_graph_destroy(self._exec)
        except Exception:                                 # noqa: BLE001 - interpreter shutdown
            pass


def ptr(t) -> int:
    """Raw address of a torch tensor / numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def pointer_array(arrays) -> "C.Array":
    return (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
