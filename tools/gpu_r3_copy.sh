#!/bin/bash
# streaming stores for the copy into the page-locked mirror (default) against plain memcpy (JDA_PIPE_PLAIN_COPY=1)
out=gpurun_out/r3_copy; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
export JDA_PIPE_TIME=1
for a in "--width 1280 --height 720 --batch 256 --batches 24" "--width 1920 --height 1080 --batch 256 --batches 16" "--batches 24"; do
  for rep in 1 2; do for plain in 1 0; do
    echo "$a : plain memcpy $plain" >> $out/copy.txt
    if [ $plain = 1 ]; then export JDA_PIPE_PLAIN_COPY=1; else unset JDA_PIPE_PLAIN_COPY; fi
    timeout 300 python tools/pipeline_bench.py --depth 4 --distinct 16 $a 2>&1 | grep "jda_pipeline_submit\|mpix_s" | sed "s/^{\"mpix_s\": \([0-9.]*\).*host_submit_ms_per_image\": \([0-9.]*\).*/  mpix_s \1 host_submit_ms_per_image \2/; s/.*strips + copy into the page-locked mirror \([0-9.]*\), enqueue.*/  copy phase \1 us/" >> $out/copy.txt
  done; done
done
cat $out/copy.txt
