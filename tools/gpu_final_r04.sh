#!/bin/bash
# Round-4 closing evidence: full -m gpu suite, the default bench line, kernel-trace stats (image direct / image winograd / video /
# bf16 training step), MFMA-utilisation counters and HBM-traffic counters of the image and video bench commands.
# Every rocprofv3 --pmc pass is kernel-trace only; the program follows `--` directly.
set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r04final}
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
timeout -k 10 600 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err || { tail -20 $O/${T}_bench_default.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60])
print("split", d["split_precision"]["value"], "winograd", d["winograd_precision"]["value"], d["winograd_precision"]["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_vs_cpu_max_rel_score_err"])
v = d["video"]; print("video", v["value"], v["roofline"]["frac"], "winograd", v["winograd_precision"]["value"]); print("stream", d["stream"]["seconds"], d["stream"]["value"], d["stream"]["parity"])
ts = d["training_step"]
for k in ("fp32", "split_precision", "bf16_precision", "winograd_precision"):
    r = ts if k == "fp32" else ts[k]
    print(k, r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["batchnorm_passes"]["share_of_step"])
print(d["reference_call_sizes"]); print(d["winograd_precision"].get("reference_call_sizes"))
PY
cd /tmp && export TMPDIR=/tmp
for wl in "image:" "winograd:--precision winograd" "video:--workload video"; do
  name=${wl%%:*}; args=${wl#*:}
  rm -rf $R/$O/${T}_trace_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${T}_trace_$name -- python3 $R/bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-train --no-video --no-small --stream-frames 0 > $R/$O/${T}_trace_$name.json 2> $R/$O/${T}_trace_$name.err
  f=$(find $R/$O/${T}_trace_$name -name "*kernel_stats.csv" | head -1); cp $f $R/$O/${T}_kernel_stats_$name.csv && echo "trace $name done"
done
rm -rf $R/$O/${T}_trace_train
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${T}_trace_train -- python3 $R/tools/train_bench.py --clips 32 --steps 5 --warmup 2 --precision bf16 > $R/$O/${T}_trace_train.json 2> $R/$O/${T}_trace_train.err
f=$(find $R/$O/${T}_trace_train -name "*kernel_stats.csv" | head -1); cp $f $R/$O/${T}_kernel_stats_train_bf16.csv && echo "trace train done"
for prec in fp32 split; do
  rm -rf $R/$O/${T}_trace_train_$prec
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${T}_trace_train_$prec -- python3 $R/tools/train_bench.py --clips 32 --steps 5 --warmup 2 --precision $prec > $R/$O/${T}_trace_train_$prec.json 2> $R/$O/${T}_trace_train_$prec.err
  f=$(find $R/$O/${T}_trace_train_$prec -name "*kernel_stats.csv" | head -1); cp $f $R/$O/${T}_kernel_stats_train_$prec.csv && echo "trace train $prec done"
done
cd $R
timeout -k 10 300 python tools/wgrad_bench.py > $O/${T}_wgrad_kernel_forms.txt 2>/dev/null && echo "wgrad forms done"
bash tools/pmc_mfma.sh ${T}_image > /dev/null 2>&1 && echo "mfma image done"
bash tools/pmc_mfma.sh ${T}_winograd --precision winograd > /dev/null 2>&1 && echo "mfma winograd done"
bash tools/pmc_mfma.sh ${T}_video --workload video > /dev/null 2>&1 && echo "mfma video done"
( cd /tmp && rm -rf $R/$O/pmc_mfma_${T}_train_bf16 && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/$O/pmc_mfma_${T}_train_bf16 -- python3 $R/tools/train_bench.py --clips 32 --steps 2 --warmup 1 --precision bf16 > /dev/null 2>&1 ) \
  && python3 tools/pmc_mfma_summary.py $O/pmc_mfma_${T}_train_bf16 > $O/pmc_mfma_${T}_train_bf16.json && echo "mfma train bf16 done"
bash tools/pmc_traffic.sh > /dev/null 2>&1 && cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic_image.json && echo "traffic image done"
bash tools/pmc_traffic.sh --workload video > /dev/null 2>&1 && cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic_video.json && echo "traffic video done"
ls gpurun_out | grep "${T}_" | grep -v trace_ | head -30
