#!/bin/bash
# same-box A/B of library builds on chosen config legs: tools/gpu_ab_configs.sh "c5_eighth,c5_quarter" ab/lib_x.so ab/lib_y.so ...  (the in-tree library rides along)
cfg=$1; shift
out=gpurun_out/ab_cfg; mkdir -p $out
for rep in $(seq 1 ${REPS:-3}); do
  for lib in "$@" jpegdec_amd/libjpegdec_amd.so; do
    JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 200 python bench.py --no-cpu-baseline --e2e-batches 0 --steps 30 --configs $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', ' '.join('%s %.4f ms frac %.3f' % (k, v.get('kernel_ms_per_launch', v.get('kernel_ms_per_step', 0)), v.get('frac', 0)) for k, v in d['configs'].items()))" | tee -a $out/ab.txt
  done
done
