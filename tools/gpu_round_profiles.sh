#!/bin/bash
# Everything the round's profiles/ files are made from, in one GPU call: tools/gpu_round_profiles.sh <tag>
tag=${1:-r01c}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/$tag/pytest_gpu.txt
tools/gpu_profile.sh $tag > /dev/null 2>&1
tools/gpu_pmc_sq.sh ${tag}_sq > gpurun_out/$tag/sq_counters.txt 2>&1
python tools/wg_balance.py > gpurun_out/$tag/wg_balance.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o prescan -- python bench.py --device-prescan --steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline > gpurun_out/$tag/prescan_bench.json 2>/dev/null
tools/gpu_config_sweep.sh > /dev/null 2>&1
cp gpurun_out/sweep.txt gpurun_out/$tag/config_sweep.txt
ls gpurun_out/$tag; cat gpurun_out/$tag/pytest_gpu.txt; tail -1 gpurun_out/$tag/bench_default.json; cat gpurun_out/$tag/config_sweep.txt
