#!/bin/bash
# A/B of library builds on ONE GPU box, ROUNDS (default 2) interleaved rounds: tools/gpu_ab2.sh libA.so libB.so ...
export TMPDIR=/tmp
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    JDA_LIBRARY=$(readlink -f $lib) python bench.py --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib  us/image %.2f  frac %.4f' % (d['roofline']['kernel_ms_per_launch']*1000/64, d['roofline']['frac']))"
  done
done
