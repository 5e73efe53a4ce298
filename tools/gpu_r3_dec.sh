#!/bin/bash
# A/B of the decode kernel: ab/lib_before.so against the in-tree library -- the GPU parity tests, then the metric kernel (three
# interleaved repetitions), then the other layouts and q98 (bench.py config legs) once each
out=gpurun_out/r3_dec; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_ref_fixtures.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
bash tools/gpu_ab_lib.sh ab/lib_before.so | tail -7
for lib in ab/lib_before.so jpegdec_amd/libjpegdec_amd.so; do
  JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 600 python bench.py --no-cpu-baseline --e2e-batches 0 --no-parity --steps 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), {k:(round(v['mpix_s']), round(v['frac'],4)) for k,v in d['configs'].items()})"
done
