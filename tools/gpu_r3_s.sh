#!/bin/bash
# A/B of two symbols a step in the walks (the tables' pair halves): ab/lib_before.so against the in-tree library on one box.
# Parity first, then every kernel's own time at depth 1 under rocprofv3, round 0's LDS padding, the end-to-end figure at depth 4.
out=gpurun_out/r3_s; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_ref_fixtures.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o d1_$tag -- python $R/tools/pipeline_bench.py --depth 1 --batches 8 --distinct 16 > /dev/null 2>&1)
done
for x in; do
  (cd /tmp && JDA_WALK_LDS_R0=$x timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o d1_r0lds$x -- python $R/tools/pipeline_bench.py --depth 1 --batches 8 --distinct 16 > /dev/null 2>&1)
done
for rep in 1 2 3; do
  for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
    echo -n "$lib " >> $out/e2e.txt
    JDA_LIBRARY=$R/$lib timeout 300 python tools/pipeline_bench.py --depth 4 --batches 24 --distinct 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), d['stats']['spec_rounds_max'])" >> $out/e2e.txt
  done
done
cat $out/e2e.txt
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/d1_*kernel_stats.csv")):
    print(f)
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 20000: print("  %-70s calls %5s avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
