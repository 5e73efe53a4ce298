#!/bin/bash
# VALU / SALU / LDS instructions per 4:2:0 tile of the default workload for each given library build
export TMPDIR=/tmp
for lib in "$@"; do
  d=gpurun_out/counts_$(basename $lib .so); mkdir -p $d
  JDA_LIBRARY=$(readlink -f $lib) timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $d -o q -- python bench.py --steps 2 --warmup 1 --batch 16 --no-parity --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("$d/q_counter_collection.csv")):
    if "jda_decode" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$lib", {k: round(sum(v) / len(v) / 16 / 6554, 1) for k, v in acc.items()}, "per 4:2:0 tile")
PY
done
