#!/bin/bash
# diagnostic library: the product objects, with dec4_fused.hip recompiled with cycle stamps
set -e
cd "$(dirname "$0")/.."
P=video-anomaly-detection_amd
python -c "import __graft_entry__ as g; g.build()" > /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -I$P/csrc -DVAD_D4_STAMPS -c $P/csrc/dec4_fused.hip -o $P/build/dec4_fused_stamps.o
OBJS=$(ls $P/build/*.o | grep -v dec4_fused)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $P/libvad_hip_stamps.so $OBJS $P/build/dec4_fused_stamps.o
ls -la $P/libvad_hip_stamps.so
