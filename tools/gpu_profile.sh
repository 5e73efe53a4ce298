#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and the two HBM PMC passes for the default
# bench workload, written under gpurun_out/$1.  Counters are collected in their own passes with
# --kernel-trace/--stats absent, as the MI355X guide prescribes.
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline"   # same workload as the default run (counters are per launch: no clock ramp needed)
python bench.py > $out/bench_default.json 2> $out/bench_default.err
# the kernel trace of the default command itself (clock ramp + 3 warm-up + 20 timed launches): its average must agree with the
# HIP-event figure in bench_default.json; the first launches of the ramp run at low clocks and pull the average up a little
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python bench.py --no-parity --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- $B > /dev/null 2>&1
ls $out
