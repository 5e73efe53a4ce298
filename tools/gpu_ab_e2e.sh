#!/bin/bash
# same-box A/B of library builds END TO END (metric e2e + the c2_e2e / c4_e2e legs), interleaved: tools/gpu_ab_e2e.sh ab/lib_x.so ...  (the in-tree library rides along)
out=gpurun_out/ab_e2e; mkdir -p $out
for rep in 1 2 3; do
  for lib in "$@" jpegdec_amd/libjpegdec_amd.so; do
    JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 20 --no-e2e-sweep --configs c2_e2e,c4_e2e,vga_e2e 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'metric_e2e', round(d['end_to_end']['mpix_s']), ' '.join('%s %d' % (k, round(v['mpix_s'])) for k, v in d['configs'].items()))" | tee -a $out/ab.txt
  done
done
