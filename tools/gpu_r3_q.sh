#!/bin/bash
out=gpurun_out/r03_q
mkdir -p $out
: > $out/pipe.txt
for thr in 8 12 16; do
  timeout 300 python tools/pipeline_bench.py --depth 4 --threads $thr --width 1280 --height 720 --batch 1024 --batches 10 --distinct 4 2>&1 | tail -1 >> $out/pipe.txt
  timeout 300 python tools/pipeline_bench.py --depth 4 --threads $thr --width 1920 --height 1080 --batch 256 --batches 16 2>&1 | tail -1 >> $out/pipe.txt
done
timeout 300 python tools/pipeline_bench.py --depth 4 --threads 16 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f batch %d depth %d threads %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d["threads"]))
PY
