#!/bin/bash
# A/B on one box: library builds x JDA_BIG_WINDOW settings x workloads.  usage: tools/gpu_ab_win.sh out_tag libA libB ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
: > $out/ab.txt
one() { # label env lib args...
  label=$1; envs=$2; lib=$3; shift 3
  env $envs JDA_LIBRARY=$(readlink -f $lib) python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s %-28s %9.0f Mpix/s  kernel %.4f ms  frac %.3f  exact %s' % (sys.argv[1], sys.argv[2], d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'], d['parity']['bit_exact'] if d.get('parity') else None))" "$label" "$(basename $lib) $envs" >> $out/ab.txt
}
for round in 1 2; do
for lib in "$@"; do
  one "420 q85" "X=1" $lib
  one "420 q95" "X=1" $lib --quality 95
  one "420 q98" "X=1" $lib --quality 98
  one "444 q85" "X=1" $lib --subsampling 4:4:4
done
lib=${@: -1}
one "420 q85" "JDA_BIG_WINDOW=1" $lib
one "420 q95" "JDA_BIG_WINDOW=1" $lib --quality 95
one "420 q98" "JDA_BIG_WINDOW=0" $lib --quality 98
one "444 q85" "JDA_BIG_WINDOW=1" $lib --subsampling 4:4:4
done
sort -s -k1,2 $out/ab.txt
