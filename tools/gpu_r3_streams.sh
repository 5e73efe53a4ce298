#!/bin/bash
# pre-scan streams (JDA_PIPE_UP_STREAMS) and batches in flight for batches of 256 small files
out=gpurun_out/r3_streams; rm -rf $out; mkdir -p $out
for a in "--width 1280 --height 720 --batch 256 --batches 24" "--width 1920 --height 1080 --batch 256 --batches 16"; do
  for s in 2 3; do for d in 3 4; do
    echo -n "$a streams $s depth $d : " >> $out/s.txt
    JDA_PIPE_UP_STREAMS=$s timeout 300 python tools/pipeline_bench.py --depth $d --distinct 16 $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']))" >> $out/s.txt
  done; done
done
cat $out/s.txt
