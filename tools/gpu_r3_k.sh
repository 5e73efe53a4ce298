#!/bin/bash
out=gpurun_out/r03_k
mkdir -p $out
: > $out/pipe.txt
for rep in 1 2; do
for gm in 1 2 4 8; do
  JDA_GRID_MULT=$gm timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
done
done
JDA_GRID_MULT=4 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
JDA_GRID_MULT=4 timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 --depth 4 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f batch %d depth %d distinct %d rounds %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d.get("distinct",0), d["stats"]["spec_rounds_max"]))
PY
