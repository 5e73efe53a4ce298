#!/bin/bash
# A/B of library builds over the kernel-only workloads: tools/gpu_ab2.sh tag libA libB ...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
: > $out/ab.txt
one() { label=$1; lib=$2; shift 2
  JDA_LIBRARY=$(readlink -f $lib) python bench.py --no-cpu-baseline --e2e-batches 0 --steps 60 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-12s %-26s %9.0f Mpix/s  kernel %.4f ms  frac %.3f  exact %s' % (sys.argv[1], sys.argv[2], d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'], d['parity']['bit_exact'] if d.get('parity') else None))" "$label" "$(basename $lib)" >> $out/ab.txt
}
for round in 1 2; do
for lib in "$@"; do
  one "420 q85" $lib
  one "420 q95" $lib --quality 95
  one "420 q98" $lib --quality 98
  one "444 q85" $lib --subsampling 4:4:4
  one "gray 8192" $lib --subsampling gray --pixel-type gray8 --width 8192 --height 8192 --batch 16
done
done
sort -s -k1,2 $out/ab.txt
