#!/bin/bash
# the pipeline's depth, pre-scan streams and host threads after the round's kernel work (64 x 4096x4096 per batch, 16 distinct files)
out=gpurun_out/r3_depth; mkdir -p $out
run() { echo -n "$* : " >> $out/sweep.txt; env $ENVV timeout 300 python tools/pipeline_bench.py --batches 24 --distinct 16 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']))" >> $out/sweep.txt; }
for d in 2 3 4 6 8; do ENVV="X=1" run --depth $d; done
for s in 1 2 3; do ENVV="JDA_PIPE_UP_STREAMS=$s"; echo -n "streams $s " >> $out/sweep.txt; run --depth 4; done
for t in 2 4 8 16; do ENVV="X=1" run --depth 4 --threads $t; done
cat $out/sweep.txt
