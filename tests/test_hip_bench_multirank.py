"""The N > 1 form of bench.py, exactly as the driver launches it (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`), rehearsed on the one GPU of the test box: two ranks share device 0 over gloo (RCCL needs one device
per rank; the code path - block partition, no data-path collective, ONE all_gather of the score vector per step, max-over-
ranks timing - is the same).  The JSON line must verify itself: both ranks' devices listed, the collective timed, and rank
0's comparison of the GATHERED vector with the CPU oracle on one frame from every rank's block within 1e-4."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload,extra", [("image", ["--batch", "24"]), ("video", ["--batch", "3", "--clip-len", "4", "--size", "64"])])
def test_two_rank_bench_line_verifies_itself(workload, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--share-gpu", "--workload", workload, "--no-split", "--no-train", *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 2
    mg = d["multi_gpu"]
    assert mg["world_size"] == 2 and mg["backend"] == "gloo" and [r["rank"] for r in mg["ranks"]] == [0, 1]
    assert mg["allgather_ms"] > 0 and len(mg["parity"]["checked_items"]) == 2
    assert mg["parity"]["within_1e-4"] and mg["parity"]["max_rel_score_err_vs_cpu"] < 1e-5
    assert "cpu_baseline" not in d                                 # the CPU baseline is an N = 1 measurement


def test_default_bench_line_carries_every_single_gpu_object():
    """`python bench.py --gpus 1 --steps K --warmup W` as the driver runs it: ONE JSON line with the contract's fields, the
    roofline and cpu_baseline objects of configs[1], and - as secondary objects that never enter `value` - configs[2] with its
    own roofline / cpu_baseline, the configs[3] stream on one rank, the training step and the reference's own call sizes."""
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=REPO, env=dict(os.environ), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"]
    assert d["value"] > 100 * c["value"]                                     # frames/s on the GPU vs the CPU restatement
    v = d["video"]
    assert v["value"] > 0 and 0 < v["roofline"]["frac"] < 1 and v["cpu_baseline"]["value"] > 0
    assert d["stream"]["frames"] == 100000 and d["stream"]["value"] > 0
    assert d["training_step"]["bf16_precision"]["value"] > 0 and d["split_precision"]["value"] > 0
    rc = d["reference_call_sizes"]
    assert set(rc) == {"image_batch_1", "image_batch_16", "video_4x16", "video_1x16"} and all(x["ms"] > 0 for x in rc.values())
