#!/bin/bash
# SQ counters per kernel of the pipeline (depth 1, 3 batches): tools/gpu_pmc_pipe.sh tag
tag=${1:-pmc_pipe}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1)); (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o g$i -- python $R/tools/pipeline_bench.py --depth 1 --batches 2 > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        if float(r.get("Grid_Size", 0) or 0) < 100000 and "fused" in k: k += " (small)"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print("    %-24s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
