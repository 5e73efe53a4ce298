#!/bin/bash
# JDA_PIPE_SPEC_ROUNDS = 2 (ab/lib_spec2.so) against 4 (in-tree) on the other inputs
out=gpurun_out/r3_spec2; rm -rf $out; mkdir -p $out
R=$GRAFT_REPO_ROOT
run() { for lib in ab/lib_spec2.so jpegdec_amd/libjpegdec_amd.so; do echo -n "$* : $lib " >> $out/other.txt; JDA_LIBRARY=$R/$lib timeout 300 python tools/pipeline_bench.py --depth 4 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), 'rounds', d['stats']['spec_rounds_max'], 'host path', d['stats']['host_path_images'])" >> $out/other.txt; done; }
run --batches 24 --distinct 2
run --batches 24 --distinct 64
run --batches 24 --distinct 16 --restart-rows 1
run --width 1920 --height 1080 --batch 256 --batches 12 --distinct 16
run --width 1280 --height 720 --batch 256 --batches 16 --distinct 16
run --subsampling 4:4:4 --batches 16 --distinct 8
run --subsampling gray --width 8192 --height 8192 --batch 16 --batches 16 --distinct 4
run --quality 98 --batches 12 --distinct 8
cat $out/other.txt
JDA_LIBRARY=$R/ab/lib_spec2.so timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
JDA_LIBRARY=$R/ab/lib_spec2.so timeout 300 python tools/gpu_fuzz_pipeline.py 30 5 | tail -1
