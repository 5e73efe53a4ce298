#!/bin/bash
# A/B on streams with restart intervals (a marker per MCU row): ab/lib_before.so against the in-tree library; the GPU tests first
out=gpurun_out/r3_z; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
for rep in 1 2 3; do
  for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
    echo -n "$lib " >> $out/e2e.txt
    JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/pipeline_bench.py --depth 4 --batches 24 --distinct 16 --restart-rows 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), d['stats']['spec_rounds_max'])" >> $out/e2e.txt
  done
done
cat $out/e2e.txt
