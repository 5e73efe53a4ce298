#!/bin/bash
# VALU/SALU/LDS instructions per tile by phase: the one-tile-per-wave kernel with phases switched off
# (JDA_DEBUG_SKIP bits: 1 = no P4, 2 = no P2/P3, 4 = no P1)
export TMPDIR=/tmp JDA_KERNEL=s
[ -n "$1" ] && export JDA_LIBRARY=$(readlink -f $1)
mkdir -p gpurun_out/phase
for skip in 0 1 3 7; do
  JDA_DEBUG_SKIP=$skip timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/phase -o p$skip -- python bench.py --steps 2 --warmup 1 --batch 16 --no-parity --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv, collections
for skip in (0, 1, 3, 7):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/phase/p%d_counter_collection.csv" % skip)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("skip", skip, {k: round(sum(v) / len(v) / 16 / 6554, 1) for k, v in acc.items()})
PY
