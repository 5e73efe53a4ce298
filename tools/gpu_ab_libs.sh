#!/bin/bash
# same-box A/B of several library builds on the metric kernel, interleaved: tools/gpu_ab_libs.sh ab/lib_x.so ab/lib_y.so ...  (the in-tree library rides along)
out=gpurun_out/ab_libs; mkdir -p $out; LIBS="$* jpegdec_amd/libjpegdec_amd.so"
for rep in 1 2 3; do
  for lib in $LIBS; do
    JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-configs --no-cpu-baseline --e2e-batches 0 --no-parity --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['roofline']['kernel_ms_per_launch'],4), round(d['roofline']['frac'],4))" | tee -a $out/ab.txt
  done
done
