#!/bin/bash
# every kernel's own time at depth 1: ab/lib_before.so against the in-tree library (no tests: for quick looks at one kernel)
out=gpurun_out/r3_k1; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o d1_$tag -- python $R/tools/pipeline_bench.py --depth 1 --batches 8 --distinct 16 > /dev/null 2>&1)
done
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/d1_*kernel_stats.csv")):
    print(f)
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 20000: print("  %-70s calls %5s avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
