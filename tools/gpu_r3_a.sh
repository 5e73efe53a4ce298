#!/bin/bash
# round 3, first GPU call: parity of the RECORD-mode pre-scan, pipeline throughput new vs old passes, kernel stats at depth 1
out=gpurun_out/r03_a
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
: > $out/pipe.txt
for d in 1 3 4; do
  timeout 300 python tools/pipeline_bench.py --depth $d --threads 8 2>&1 | tail -1 >> $out/pipe.txt
  JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --depth $d --threads 8 2>&1 | tail -1 >> $out/pipe.txt
done
timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for l in open("$out/pipe.txt"):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f ms/img  batch %d depth %d thr %d rounds %d devimgs %d hostimgs %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d["threads"], d["stats"]["spec_rounds_max"], d["stats"]["device_images"], d["stats"]["host_path_images"]))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o pipe_d1 -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 1 --threads 8 --batches 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv
for r in csv.DictReader(open("$out/pipe_d1_kernel_stats.csv")):
    print("%-70s calls %4s  avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
