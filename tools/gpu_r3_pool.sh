#!/bin/bash
# the worker pool handing out a few items per visit to its mutex: ab/lib_before.so against the in-tree library on batches of many small files
out=gpurun_out/r3_pool; rm -rf $out; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_c_api.py -x -q -m gpu 2>&1 | tail -1
export JDA_PIPE_TIME=1
for a in "--width 1280 --height 720 --batch 256 --batches 24" "--width 1920 --height 1080 --batch 256 --batches 16" "--batches 24"; do
  for rep in 1 2; do for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
    echo "$a : $lib" >> $out/pool.txt
    JDA_LIBRARY=$R/$lib timeout 300 python tools/pipeline_bench.py --depth 4 --distinct 16 $a 2>&1 | grep "jda_pipeline_submit\|mpix_s" | sed "s/^{\"mpix_s\": \([0-9.]*\).*host_submit_ms_per_image\": \([0-9.]*\).*/  mpix_s \1 host_submit_ms_per_image \2/" >> $out/pool.txt
  done; done
done
cat $out/pool.txt
