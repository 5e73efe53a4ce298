#!/bin/bash
# every file of a batch different (64 distinct), and the rounds they need
out=gpurun_out/r3_d64; mkdir -p $out
for d in 16 64; do
  echo -n "distinct $d: " >> $out/e2e.txt
  timeout 600 python tools/pipeline_bench.py --depth 4 --batches 24 --distinct $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), 'rounds', d['stats']['spec_rounds_max'], 'host path', d['stats']['host_path_images'])" >> $out/e2e.txt
done
cat $out/e2e.txt
