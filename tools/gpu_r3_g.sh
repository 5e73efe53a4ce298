#!/bin/bash
out=gpurun_out/r03_g
mkdir -p $out
: > $out/pipe.txt
for skew in 0 4352 69888 1052672; do
  JDA_PIPE_REC_SKEW=$skew timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
done
JDA_PIPE_REC_SKEW=4352 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  distinct %d rounds %d" % (d["mpix_s"], d["ms_per_image"], d.get("distinct",0), d["stats"]["spec_rounds_max"]))
PY
