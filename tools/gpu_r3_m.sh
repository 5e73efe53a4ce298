#!/bin/bash
out=gpurun_out/r03_m
mkdir -p $out
make nodeuser >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
: > $out/bench.txt
for lv in 1 2 3 1 2 3; do
  JDA_DECODE_LEVELS=$lv timeout 300 python bench.py --no-configs --no-cpu-baseline --e2e-batches 0 --no-parity --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('levels $lv', round(d['value']), round(d['roofline']['kernel_ms_per_launch'],4), round(d['roofline']['frac'],4))" >> $out/bench.txt
done
cat $out/bench.txt
: > $out/pipe.txt
for rep in 1 2; do
for lv in 1 3 4 6; do
  JDA_PIPE_DECODE_LEVELS=$lv timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
done
done
JDA_PIPE_DECODE_LEVELS=4 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_DECODE_LEVELS=4 timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 --depth 4 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_DECODE_LEVELS=1 timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 --depth 4 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f batch %d depth %d distinct %d rounds %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d.get("distinct",0), d["stats"]["spec_rounds_max"]))
PY
