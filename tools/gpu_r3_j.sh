#!/bin/bash
out=gpurun_out/r03_j
mkdir -p $out
: > $out/pipe.txt
for rep in 1 2; do
for ups in 2 3; do
  JDA_PIPE_UP_STREAMS=$ups timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
done
done
JDA_PIPE_UP_STREAMS=3 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_UP_STREAMS=1 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f batch %d depth %d distinct %d rounds %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d.get("distinct",0), d["stats"]["spec_rounds_max"]))
PY
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o rec_d16 -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 4 --threads 8 --batches 12 --distinct 16 > /dev/null 2>&1
