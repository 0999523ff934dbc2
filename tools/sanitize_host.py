"""AddressSanitizer + UndefinedBehaviorSanitizer run of the host-side native code (CPU only; GPU sanitizers are not
available on the pool): `csrc/pack.cpp` (BatchNorm folding, operand re-ordering, zero-padded widths) and the C oracle
`oracle/vad_oracle.c`.

    python tools/sanitize_host.py            # build both with -fsanitize=address,undefined, run the driver below
    python tools/sanitize_host.py --driver   # (internal) the sanitized run itself; needs the ASan runtime preloaded

ROCm's clang is the compiler (g++ 11 has no `_Float16`, which the split-fp16 packer uses); its shared ASan runtime is
preloaded into a child Python that loads the two libraries with ctypes and drives every packer (exact / split, padded and
unpadded widths, error returns) and the oracle's image and video forward against the reference's golden vectors.  numpy
only: torch is not imported under the sanitizer.  Any report aborts the child (`-fno-sanitize-recover`, `abort_on_error`).
"""
from __future__ import annotations

import ctypes as C
import glob
import importlib.util
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "video-anomaly-detection_amd" / "csrc"
CLANG = Path(os.environ.get("VAD_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang"))
SAN = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
       "-shared-libsan", "-fPIC", "-shared"]


def asan_runtime() -> Path:
    hits = glob.glob(str(CLANG.parent.parent / "lib" / "clang" / "*" / "lib" / "linux" / "libclang_rt.asan-x86_64.so"))
    if not hits:
        raise FileNotFoundError("libclang_rt.asan-x86_64.so not found next to " + str(CLANG))
    return Path(hits[0])


def build(outdir: Path) -> tuple[Path, Path]:
    pack = outdir / "libvad_pack_san.so"
    orc = outdir / "libvad_oracle_san.so"
    subprocess.run([str(CLANG) + "++", "-std=c++17", *SAN, f"-I{REPO / 'include'}", f"-I{CSRC}", str(CSRC / "pack.cpp"),
                    "-o", str(pack)], check=True)
    subprocess.run([str(CLANG), "-std=c11", "-ffp-contract=off", "-Wno-comment", *SAN, str(REPO / "oracle" / "vad_oracle.c"),
                    "-o", str(orc), "-lm"], check=True)
    return pack, orc


# ----------------------------------------------------------------------------------------------- the sanitized run
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _vp(a):
    return C.c_void_p(a.ctypes.data)          # (a bare int would be truncated to a C int)


def _ptrs(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _golden_state(synth, g):
    shapes = {k: tuple(int(d) for d in s.split(",")) if s else () for k, s in zip(g["keys"], g["shapes"])}
    return synth.synthetic_state(shapes, int(g["wseed"]))


def _float_params(state):
    return [np.ascontiguousarray(np.asarray(v, dtype=np.float32)) for k, v in state.items()
            if not k.endswith("num_batches_tracked")]


def driver(pack_path: str, oracle_path: str) -> None:
    synth = _load("_vad_synth", REPO / "video-anomaly-detection_amd" / "synth.py")
    gold = REPO / "tests" / "golden"
    L = C.CDLL(pack_path)
    for f in ("vad_img_packed_floats", "vad_vid_packed_floats", "vad_pack_conv3x3_floats", "vad_pack_convt2x2_floats",
              "vad_pack_conv1x1_floats", "vad_pack_conv3x3_to3_floats", "vad_pack_conv3x3_c3_floats", "vad_pack_conv3x3_wino_floats"):
        getattr(L, f).restype = C.c_size_t
    L.vad_last_error.restype = C.c_char_p
    checked = 0
    # whole-model packers on the reference's own constructor shapes, padded and unpadded widths, both operand forms
    for name in ("img_l32_32.npz", "img_l256_64.npz", "img_l100_32.npz"):
        g = np.load(gold / name)
        params = _float_params(_golden_state(synth, g))
        latent = int(g["latent_dim"])
        for prec in (0, 1, 4):                 # exact fp32, split fp16, Winograd (VAD_PREC_WINO)
            blob = np.full(L.vad_img_packed_floats(3, latent), np.nan, np.float32)
            assert L.vad_img_pack(_ptrs(params), len(params), 3, latent, prec, _vp(blob)) == 0, L.vad_last_error()
            assert np.isfinite(blob[4:]).all() if prec == 0 else True
            assert L.vad_blob_precision(_vp(blob)) == prec
            checked += 1
        assert L.vad_img_pack(_ptrs(params), len(params) - 1, 3, latent, 0, _vp(blob)) < 0     # wrong tensor count
        assert L.vad_img_pack(_ptrs(params), len(params), 3, latent, 7, _vp(blob)) < 0           # unknown precision
    for name in ("vid_default_64.npz", "vid_proj_32.npz", "vid_l3_32.npz", "vid_l48_h96_32.npz", "vid_l100_32.npz"):
        g = np.load(gold / name)
        params = _float_params(_golden_state(synth, g))
        lat, hid, layers = int(g["latent_dim"]), int(g["hid"]), int(g["layers"])
        for prec in (0, 1, 4):
            n = L.vad_vid_packed_floats(lat, hid, layers)
            assert n > 0, name
            blob = np.full(n, np.nan, np.float32)
            assert L.vad_vid_pack(_ptrs(params), len(params), lat, hid, layers, prec, _vp(blob)) == 0, L.vad_last_error()
            checked += 1
        assert L.vad_vid_pack(_ptrs(params), len(params), lat, hid, layers + 1, 0, _vp(blob)) < 0
    assert L.vad_vid_packed_floats(128, 128, 9) == 0 and L.vad_img_packed_floats(3, 0) == 0
    # models with more than 3 input planes (round 4: generic first / last layer slots, planes padded to 32)
    L.vad_vid_packed_floats_c.restype = C.c_size_t
    g = np.load(gold / "img_c5_l32_32.npz")
    params, latent, cin = _float_params(_golden_state(synth, g)), int(g["latent_dim"]), int(g["in_channels"])
    for prec in (0, 1, 4):
        blob = np.full(L.vad_img_packed_floats(cin, latent), np.nan, np.float32)
        assert L.vad_img_pack(_ptrs(params), len(params), cin, latent, prec, _vp(blob)) == 0, L.vad_last_error()
        checked += 1
    assert L.vad_img_packed_floats(33, latent) == 0 and L.vad_img_pack(_ptrs(params), len(params), 2, latent, 0, _vp(blob)) < 0
    g = np.load(gold / "vid_c4_l32_32.npz")
    params, cin = _float_params(_golden_state(synth, g)), int(g["in_channels"])
    lat, hid, layers = int(g["latent_dim"]), int(g["hid"]), int(g["layers"])
    for prec in (0, 1, 4):
        blob = np.full(L.vad_vid_packed_floats_c(cin, lat, hid, layers), np.nan, np.float32)
        assert L.vad_vid_pack_c(_ptrs(params), len(params), cin, lat, hid, layers, prec, _vp(blob)) == 0, L.vad_last_error()
        checked += 1
    assert L.vad_vid_packed_floats_c(33, lat, hid, layers) == 0
    # single-layer packers, odd-but-legal shapes
    rng = np.random.default_rng(0)
    for cout, cin in ((32, 32), (64, 48), (96, 8)):
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        bn = [rng.uniform(0.5, 1.5, cout).astype(np.float32) for _ in range(4)]
        out = np.empty(L.vad_pack_conv3x3_floats(cout, cin), np.float32)
        bo = np.empty(cout, np.float32)
        for prec in ((0, 1) if cin % 16 == 0 else (0,)):
            assert L.vad_pack_conv3x3(_vp(w), _vp(b), _ptrs(bn), cout, cin, prec, _vp(out), _vp(bo)) == 0
        ow = np.empty(L.vad_pack_conv3x3_wino_floats(cout, cin), np.float32)
        assert L.vad_pack_conv3x3_wino(_vp(w), _vp(b), _ptrs(bn), cout, cin, _vp(ow), _vp(bo)) == 0 and np.isfinite(ow).all()
        assert L.vad_pack_conv3x3(_vp(w), _vp(b), _ptrs(bn), cout, cin, 4, _vp(out), _vp(bo)) < 0      # the direct packer refuses the mode
        wt = rng.standard_normal((cin, cout, 2, 2)).astype(np.float32)
        ot = np.empty(L.vad_pack_convt2x2_floats(cin, cout), np.float32)
        assert L.vad_pack_convt2x2(_vp(wt), _vp(b), None, cin, cout, 0, _vp(ot), _vp(bo)) == 0
        w1 = rng.standard_normal((cout, cin)).astype(np.float32)
        o1 = np.empty(L.vad_pack_conv1x1_floats(cout, cin), np.float32)
        assert L.vad_pack_conv1x1(_vp(w1), _vp(b), cout, cin, _vp(o1), _vp(bo)) == 0
        checked += 1
    w3 = rng.standard_normal((3, 32, 3, 3)).astype(np.float32)
    o3 = np.empty(L.vad_pack_conv3x3_to3_floats(32), np.float32)
    assert L.vad_pack_conv3x3_to3(_vp(w3), 32, _vp(o3)) == 0
    wc = rng.standard_normal((32, 3, 3, 3)).astype(np.float32)
    oc = np.empty(L.vad_pack_conv3x3_c3_floats(32), np.float32)
    bc = np.empty(32, np.float32)
    assert L.vad_pack_conv3x3_c3(_vp(wc), None, None, 32, _vp(oc), _vp(bc)) == 0

    # the C oracle under the sanitizers, against the reference's golden vectors
    co = _load("_vad_c_oracle", REPO / "oracle" / "c_oracle.py")
    co._lib = C.CDLL(oracle_path)
    for name in ("img_l32_32.npz", "img_l100_32.npz"):
        g = np.load(gold / name)
        x = synth.frames(int(g["xseed"]), 0, int(g["n"]), 3, int(g["hw"]), int(g["hw"]))
        out = co.img_scores(_golden_state(synth, g), int(g["latent_dim"]), x)
        assert np.max(np.abs(out["scores"] - g["scores"]) / np.abs(g["scores"])) < 2e-6, name
        checked += 1
    for name in ("vid_proj_32.npz", "vid_l48_h96_32.npz", "vid_l100_32.npz"):
        g = np.load(gold / name)
        x = synth.clips(int(g["xseed"]), 0, int(g["b"]), int(g["t"]), 3, int(g["hw"]), int(g["hw"]))
        out = co.vid_scores(_golden_state(synth, g), int(g["latent_dim"]), int(g["hid"]), int(g["layers"]), x)
        assert np.max(np.abs(out["frame"] - g["frame"]) / np.abs(g["frame"])) < 2e-6, name
        checked += 1
    print(f"sanitized host run ok: {checked} cases, no ASan / UBSan report")


def run(outdir: Path | None = None) -> subprocess.CompletedProcess:
    with tempfile.TemporaryDirectory() as tmp:
        pack, orc = build(Path(outdir or tmp))
        env = dict(os.environ, LD_PRELOAD=str(asan_runtime()),
                   ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
        return subprocess.run([sys.executable, str(Path(__file__).resolve()), "--driver", str(pack), str(orc)], env=env,
                              capture_output=True, text=True)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--driver":
        driver(sys.argv[2], sys.argv[3])
    else:
        r = run()
        sys.stdout.write(r.stdout)
        sys.stderr.write(r.stderr[-4000:])
        sys.exit(r.returncode)
