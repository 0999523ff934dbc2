#!/bin/bash
# Where do the training kernels wait?  Two rocprofv3 --pmc passes (kernel-trace only) of tools/train_bench.py: wave-time /
# memory-latency counters, then L2 hit / miss counters.  usage: tools/pmc_train_mem.sh <tag> [precision]
set -e
tag=$1; P=${2:-bf16}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mem_${tag}_a $R/gpurun_out/pmc_mem_${tag}_b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mem_${tag}_a -- python3 $R/tools/train_bench.py --clips 32 --steps 2 --warmup 1 --precision $P > /dev/null 2> $R/gpurun_out/pmc_mem_${tag}_a.err || { tail -5 $R/gpurun_out/pmc_mem_${tag}_a.err; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mem_${tag}_b -- python3 $R/tools/train_bench.py --clips 32 --steps 2 --warmup 1 --precision $P > /dev/null 2> $R/gpurun_out/pmc_mem_${tag}_b.err || { tail -5 $R/gpurun_out/pmc_mem_${tag}_b.err; }
cd $R
python3 - <<PY
import csv, glob, json
from collections import defaultdict
csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(dict)
for d in ("gpurun_out/pmc_mem_${tag}_a", "gpurun_out/pmc_mem_${tag}_b"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][(d, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
out = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    us = sum(dur[k].values()) / len(dur[k])
    if us < 50: continue
    e = {"avg_us": round(us, 1)}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        e["parked(waitcnt/barrier)"] = round(m.get("SQ_WAIT_ANY", 0) / wc, 3)
        e["issue_stall"] = round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    if m.get("SQ_INSTS_VMEM_RD"):
        e["avg_vmem_latency_cycles"] = round(m.get("SQ_INST_LEVEL_VMEM", 0) / max(m.get("SQ_INSTS_VMEM_RD", 1), 1), 1)
    if m.get("TCC_REQ_sum"):
        e["L2_hit_rate"] = round(m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1), 3)
        e["L2_req_per_us"] = round(m.get("TCC_REQ_sum", 0) / us, 1)
        e["EA_rdreq"] = round(m.get("TCC_EA0_RDREQ_sum", 0))
    e["counters"] = {c: round(v) for c, v in sorted(m.items())}
    out[k] = e
json.dump(out, open("gpurun_out/pmc_mem_${tag}.json", "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"]):
    print(k[:60], {a: b for a, b in e.items() if a != "counters"})
PY
