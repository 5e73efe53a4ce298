#!/bin/bash
# instruction counts per 4:2:0 tile for library builds: tools/gpu_counts2.sh libA.so libB.so ...
export TMPDIR=/tmp
for lib in "$@"; do
  d=gpurun_out/counts_$(basename $lib .so); rm -rf $d; mkdir -p $d
  JDA_LIBRARY=$(readlink -f $lib) timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $d -o q -- python bench.py --steps 2 --warmup 1 --batch 16 --no-parity --no-cpu-baseline > /dev/null 2>&1
  python - "$lib" "$d" <<PY
import csv, collections, sys, glob
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v) / len(v) / 16 / 6554, 1) for k, v in acc.items()}, "per 4:2:0 tile")
PY
done
