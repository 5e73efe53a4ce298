#!/bin/bash
# A/B of library builds on other workloads: tools/gpu_ab_cfg.sh libA.so libB.so -- "<bench args>" ...
libs=(); while [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
export TMPDIR=/tmp
for cfg in "$@"; do
  for lib in "${libs[@]}"; do
    JDA_LIBRARY=$(readlink -f $lib) python bench.py $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %-60s %9.0f Mpix/s kernel-only' % ('$lib', '$cfg', d['kernel_only_mpix_s']))"
  done
done
