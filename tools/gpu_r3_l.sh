#!/bin/bash
out=gpurun_out/r03_l
mkdir -p $out
: > $out/pipe.txt
run() { timeout 300 python tools/pipeline_bench.py "$@" 2>&1 | tail -1 >> $out/pipe.txt; }
run --depth 4 --batches 16 --quality 98
run --depth 4 --batches 16 --subsampling 4:4:4
run --depth 4 --batches 40 --restart-rows 1
run --depth 4 --width 1920 --height 1080 --batch 256 --batches 16
run --depth 4 --width 1920 --height 1080 --batch 1024 --batches 6
run --depth 4 --width 1280 --height 720 --batch 1024 --batches 8 --distinct 4
run --depth 4 --width 8192 --height 8192 --subsampling gray --batch 16 --batches 16
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f batch %d depth %d distinct %d rounds %d dev %d host-path %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d.get("distinct",0), d["stats"]["spec_rounds_max"], d["stats"]["device_images"], d["stats"]["host_path_images"]))
PY
