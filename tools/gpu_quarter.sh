#!/bin/bash
# 1/4 scale on one box: parity (the whole GPU suite), the c5 leg at 1/4 and 1/8, the metric batch at 1/4 (4:2:0 -> RGB8888) and 4:4:4 at 1/4
tag=${1:-q4}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest.txt; cat $out/pytest.txt
B="--no-cpu-baseline --e2e-batches 0 --steps 200"
timeout 200 python bench.py $B --configs c5_quarter,c5_eighth 2> $out/c5.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d['configs'].items(): print(k, round(v['kernel_ms_per_launch'], 4), 'ms', round(v['frac'], 4), v['parity_image_0']['bit_exact'])
" | tee $out/c5.txt
for sub in 4:2:0 4:4:4; do
timeout 200 python bench.py $B --no-configs --options 4 --subsampling $sub 2> $out/m.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sub at 1/4 -> RGB8888:', round(d['roofline']['kernel_ms_per_launch'], 4), 'ms', round(d['value']), 'Mpix/s', d.get('parity'))
" | tee -a $out/c5.txt
done
