#!/bin/bash
# round 3: RECORD mode for restart streams -- parity, then throughput against round 2's passes
out=gpurun_out/r03_f
mkdir -p $out
export TMPDIR=/tmp
make nodeuser >/dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
: > $out/pipe.txt
for rep in 1 2; do
  timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
  JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
done
timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 16 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%s %8.0f Mpix/s  %.4f ms/img  host %.4f  %dx%d rst %s batch %d depth %d rounds %d devimgs %d hostimgs %d" % ("REC" if i%2==0 else "OLD", d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d.get("width",0), d.get("height",0), d.get("restart_rows"), d["batch"], d["depth"], d["stats"]["spec_rounds_max"], d["stats"]["device_images"], d["stats"]["host_path_images"]))
PY
