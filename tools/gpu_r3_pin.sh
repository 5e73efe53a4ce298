#!/bin/bash
# pipeline_bench with the process on the GPU's NUMA node (default) against --no-pin: how much of the run-to-run spread is placement?
out=gpurun_out/r3_pin; rm -rf $out; mkdir -p $out
for a in "--width 1280 --height 720 --batch 256 --batches 24" "--width 1920 --height 1080 --batch 256 --batches 16" "--batches 24"; do
  for rep in 1 2 3; do for pin in "" "--no-pin"; do
    echo -n "$a $pin : " >> $out/pin.txt
    timeout 300 python tools/pipeline_bench.py --depth 4 --distinct 16 $a $pin 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), d.get('pinned'))" >> $out/pin.txt
  done; done
done
cat $out/pin.txt
