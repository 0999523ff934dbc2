#!/bin/bash
# Evidence session on the GPU box: kernel-trace stats + MFMA-utilisation counters + HBM traffic counters for the image and
# video bench commands.  Every rocprofv3 --pmc pass is kernel-trace only; the program follows `--` directly.
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r02}
cd /tmp && export TMPDIR=/tmp
for wl in image video; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_trace_$wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-train --no-video --stream-frames 0 > $R/gpurun_out/${T}_trace_$wl.json 2> $R/gpurun_out/${T}_trace_$wl.err
  echo "trace $wl done"
done
cd $R
bash tools/pmc_mfma.sh ${T}_image > /dev/null && echo "mfma image done"
bash tools/pmc_mfma.sh ${T}_video --workload video > /dev/null && echo "mfma video done"
bash tools/pmc_traffic.sh > /dev/null && cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic_image.json && echo "traffic image done"
bash tools/pmc_traffic.sh --workload video > /dev/null && cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic_video.json && echo "traffic video done"
for wl in image video; do f=$(find gpurun_out/${T}_trace_$wl -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${T}_kernel_stats_$wl.csv; done
ls gpurun_out | grep ${T}_ | head -40
