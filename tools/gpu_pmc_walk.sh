#!/bin/bash
# what do the pre-scan walkers wait for?  tools/gpu_pmc_walk.sh tag
tag=${1:-pmc_walk}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $R/$out/sq_counters_available.txt)
i=0
for grp in "SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_IFETCH SQ_WAVES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH" "SQ_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM"; do
  i=$((i+1)); (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o g$i -- python $R/tools/pipeline_bench.py --depth 1 --batches 2 > $R/$out/g$i.log 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-30:]
        if float(r.get("Grid_Size", 0) or 0) < 600000 and "fused" in k: k += " (list)"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "rocclr" in k: continue
    print(k)
    for c, v in sorted(d.items()): print("    %-26s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
