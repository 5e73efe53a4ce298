#!/bin/bash
# A/B of two library builds on one box, interleaved: tools/gpu_ab_lib.sh ab/lib_before.so [bench args]  (B = the in-tree library)
A=$1; shift
out=gpurun_out/ab_lib; mkdir -p $out; echo "# $A vs in-tree, bench args: $*" >> $out/ab.txt
for rep in $(seq 1 ${REPS:-3}); do
  for lib in $A jpegdec_amd/libjpegdec_amd.so; do
    JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-configs --no-cpu-baseline --e2e-batches 0 --no-parity --steps 200 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['roofline']['kernel_ms_per_launch'],4), round(d['roofline']['frac'],4))" >> $out/ab.txt
  done
done
cat $out/ab.txt
python - <<PY
import collections
acc = collections.defaultdict(list)
for ln in open("$out/ab.txt"):
    f = ln.split()
    if len(f) == 4 and not ln.startswith("#"): acc[f[0]].append(float(f[2]))
for k, v in acc.items(): print("%-40s n=%d  mean %.4f ms  min %.4f  max %.4f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
