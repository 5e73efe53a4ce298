#!/bin/bash
out=gpurun_out/${1:-r02_pipe3}
mkdir -p $out
export TMPDIR=/tmp
JDA_PIPE_TRACE=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ref_fixtures.py -m gpu -q -x 2>&1 | grep -v "^$" | grep "jda_pipeline:\|passed\|failed\|Error\|^E " | head -30 > $out/pytest_pipe.txt
cat $out/pytest_pipe.txt
: > $out/pipe.txt
timeout 300 python tools/pipeline_bench.py --depth 2 --threads 8 2>&1 | tail -1 >> $out/pipe.txt
timeout 300 python tools/pipeline_bench.py --depth 2 --threads 8 --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
timeout 300 python tools/pipeline_bench.py --depth 2 --threads 8 --restart-rows 4 2>&1 | tail -1 >> $out/pipe.txt
timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 8 --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for l in open("$out/pipe.txt"):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f ms/img  batch %d depth %d thr %d rounds %d devimgs %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d["threads"], d["stats"]["spec_rounds_max"], d["stats"]["device_images"]))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o pipe -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 2 --threads 8 --batches 6 --restart-rows 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv
for r in csv.DictReader(open("$out/pipe_kernel_stats.csv")):
    print("%-70s calls %4s  avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
