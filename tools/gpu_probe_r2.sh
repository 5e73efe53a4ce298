#!/bin/bash
# round 2, first GPU call: all GPU tests, the box's CPU / PCIe facts, two ranks on one GPU, baseline numbers for the high-bitrate work
out=gpurun_out/r02_probe
mkdir -p $out
export TMPDIR=/tmp
( nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | head -25; numactl -H 2>/dev/null | head -12; rocm-smi --showtopo 2>/dev/null | head -30 ) > $out/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $out/pytest_gpu.txt
python tools/probe_h2d.py > $out/h2d.txt 2>&1
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/probe_two_ranks.py nccl > $out/two_ranks_nccl.txt 2>&1
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/probe_two_ranks.py gloo > $out/two_ranks_gloo.txt 2>&1
for q in 85 95 98; do
  python bench.py --quality $q --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_q$q.json
done
python bench.py --subsampling 4:4:4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_444.json
python bench.py --subsampling gray --pixel-type gray8 --width 8192 --height 8192 --batch 16 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_gray.json
cat $out/pytest_gpu.txt; tail -3 $out/two_ranks_nccl.txt; cat $out/h2d.txt
for f in $out/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value']), 'Mpix/s frac', round(d['roofline']['frac'],3), 'bits/px', d['config']['bits_per_pixel'], d['parity'])"; done
