#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and the two HBM PMC passes for the default
# bench workload, written under gpurun_out/$1.  Counters are collected in their own passes with
# --kernel-trace/--stats absent, as the MI355X guide prescribes.
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --no-parity --no-cpu-baseline"   # same workload as the default run
python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- $B > /dev/null 2>&1
ls $out
