#!/bin/bash
# A/B of the in-tree library against ab/lib_before.so on one box: the GPU parity tests first, then every kernel's own time at
# depth 1 under rocprofv3, then the end-to-end figure at depth 4 (64 x 4096x4096 per batch, 16 distinct files), interleaved.
out=gpurun_out/r3_s; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_ref_fixtures.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 600 python tools/gpu_fuzz_pipeline.py > $out/fuzz.txt 2>&1; tail -1 $out/fuzz.txt
for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o d1_$tag -- python $R/tools/pipeline_bench.py --depth 1 --batches 8 --distinct 16 > /dev/null 2>&1)
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
