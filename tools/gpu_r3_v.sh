#!/bin/bash
# the walks' time against the steps between two reloads of a walker's stream registers (JDA_SEG_REFILL_STEPS; ab/lib_refill<N>.so)
out=gpurun_out/r3_v; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in jpegdec_amd/libjpegdec_amd.so ab/lib_refill*.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o $tag -- python $R/tools/pipeline_bench.py --depth 1 --batches 6 --distinct 16 > $R/$out/$tag.txt 2>&1)
  tail -1 $out/$tag.txt | cut -c1-80
done
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/*kernel_stats.csv")):
    d = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
    print("%-40s r0 %6.1f  counting rounds (avg of 3) %6.1f  tail %6.1f" % (f.split("/")[-1].replace("_kernel_stats.csv", ""), d.get("jda_segscan_fused<0>", 0), d.get("jda_segscan_fused<3>", 0), d.get("jda_segscan_tail", 0)))
PY
