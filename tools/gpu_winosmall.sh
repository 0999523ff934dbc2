#!/bin/bash
# Winograd mode at the reference's call sizes + the batch-independence tests
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_models.py tests/test_hip_layers.py -m gpu -x -q -k "winograd" 2>&1 | tail -2
python - <<'PY'
import importlib, json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
vad = importlib.import_module("video-anomaly-detection_amd")
lib = vad.hip.lib()
dev = torch.device("cuda", 0)
m = vad.ConvAutoencoder(in_channels=3, latent_dim=256)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, 7).items()}, strict=True)
m = m.to(dev).eval()
for prec in ("fp32", "winograd"):
    print(prec, json.dumps(bench.reference_call_sizes(vad, lib, m, dev, 256, 0xC0FFEE + 1, precision=prec)))
PY
