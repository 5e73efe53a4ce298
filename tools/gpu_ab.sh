#!/bin/bash
# A/B of library builds on ONE GPU box (box-to-box variation is ~1 %): tools/gpu_ab.sh libA.so libB.so ...
# Each library is benched 3 times, interleaved; prints us per 4096x4096 image.
export TMPDIR=/tmp
for round in 1 2 3; do
  for lib in "$@"; do
    JDA_LIBRARY=$(readlink -f $lib) python bench.py --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib  us/image %.2f  frac %.3f' % (d['roofline']['kernel_ms_per_launch']*1000/64, d['roofline']['frac']))"
  done
done
