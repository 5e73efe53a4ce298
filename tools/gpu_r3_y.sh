#!/bin/bash
# every kernel's own time for batches of 256 x 1920x1080 and 256 x 1280x720 (pipeline at depth 1)
out=gpurun_out/r3_y; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wh in "1920 1080" "1280 720"; do
  set -- $wh
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$out -o d1_$1 -- python $R/tools/pipeline_bench.py --depth 1 --batch 256 --batches 6 --distinct 16 --width $1 --height $2 > $R/$out/run_$1.txt 2>&1)
  tail -1 $out/run_$1.txt | cut -c1-200
done
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/d1_*kernel_stats.csv")):
    print(f)
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 5000: print("  %-70s calls %5s avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in sorted(glob.glob("$out/d1_*memory_copy_stats.csv")):
    print(f); print(open(f).read()[:600])
PY
