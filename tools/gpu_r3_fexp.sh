#!/bin/bash
# what the filter's write kernel spends its time on: builds that leave parts out (results wrong, times informative)
out=gpurun_out/r3_fexp; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in jpegdec_amd/libjpegdec_amd.so ab/lib_fexp*.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o $tag -- python $R/tools/pipeline_bench.py --depth 1 --batches 4 --distinct 16 > $R/$out/$tag.txt 2>&1)
done
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/*kernel_stats.csv")):
    d = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
    print("%-30s filter_count %6.1f write %6.1f" % (f.split("/")[-1].replace("_kernel_stats.csv", ""), d.get("jda_filter_count", 0), d.get("jda_filter_write", 0)))
PY
