#!/bin/bash
# instruction counts per 4:4:4 tile (21 MCUs = 63 blocks; 4096x4096 = 512 rows x 25 tiles = 12,800 tiles per image): tools/gpu_counts444.sh libA.so ...
export TMPDIR=/tmp
for lib in "$@"; do
  d=gpurun_out/counts444_$(basename $lib .so); rm -rf $d; mkdir -p $d
  JDA_LIBRARY=$(readlink -f $lib) timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $d -o q -- python bench.py --subsampling 4:4:4 --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline > /dev/null 2>&1
  python - "$lib" "$d" <<PY
import csv, collections, sys, glob
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v) / len(v) / 16 / 12800, 1) for k, v in acc.items()}, "per 4:4:4 tile")
PY
done
