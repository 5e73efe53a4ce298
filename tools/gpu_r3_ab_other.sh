#!/bin/bash
# ab/lib_before.so against the in-tree library on the inputs a walk change could hurt: q98 (long symbols), restart rows, the reference's photographs
out=gpurun_out/r3_ab_other; rm -rf $out; mkdir -p $out
R=$GRAFT_REPO_ROOT
run() { for rep in 1 2; do for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do echo -n "$* : $lib " >> $out/other.txt; JDA_LIBRARY=$R/$lib timeout 300 python tools/pipeline_bench.py --depth 4 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), 'rounds', d['stats']['spec_rounds_max'], 'host path', d['stats']['host_path_images'])" >> $out/other.txt; done; done; }
run --quality 98 --batches 12 --distinct 8
run --batches 24 --distinct 16 --restart-rows 1
run --width 1920 --height 1080 --batch 256 --batches 12 --distinct 16
cat $out/other.txt
for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
  (cd /tmp && export TMPDIR=/tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o q98_$(basename $lib .so) -- python $R/tools/pipeline_bench.py --depth 1 --batches 4 --distinct 8 --quality 98 > /dev/null 2>&1)
done
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/q98_*kernel_stats.csv")):
    d = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
    print("%-40s r0 %6.1f  counting rounds (avg of 3) %6.1f  tail %6.1f" % (f.split("/")[-1].replace("_kernel_stats.csv", ""), d.get("jda_segscan_fused<0>", 0), d.get("jda_segscan_fused<3>", 0), d.get("jda_segscan_tail", 0)))
PY
