#!/bin/bash
export TMPDIR=/tmp
for round in 1 2; do for m in 1 2 3 4 8; do
  for cfg in "" "--subsampling 4:4:4" "--batch 1024 --width 1280 --height 720 --distinct 4"; do
  JDA_GRID_MULT=$m python bench.py --no-cpu-baseline --no-parity --e2e-batches 0 --steps 60 $cfg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mult $m  %-50s %9.0f Mpix/s kernel %.4f ms' % ('$cfg', d['value'], d['roofline']['kernel_ms_per_launch']))"
done; done; done
