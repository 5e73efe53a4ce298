#!/bin/bash
# round 3: A/B of the RECORD-mode pre-scan against round 2's passes, repeated and interleaved; kernel trace at depth 4
out=gpurun_out/r03_b
mkdir -p $out
export TMPDIR=/tmp
: > $out/pipe.txt
for rep in 1 2 3; do
  for dist in 2 16; do
    timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct $dist 2>&1 | tail -1 >> $out/pipe.txt
    JDA_PIPE_NO_RECORD=1 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct $dist 2>&1 | tail -1 >> $out/pipe.txt
  done
done
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%s %8.0f Mpix/s  %.4f ms/img  host %.4f  distinct %d depth %d rounds %d devimgs %d hostimgs %d" % ("REC" if i%2==0 else "OLD", d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d.get("distinct",0), d["depth"], d["stats"]["spec_rounds_max"], d["stats"]["device_images"], d["stats"]["host_path_images"]))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o pipe_d4 -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 4 --threads 8 --batches 12 --distinct 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv
for r in csv.DictReader(open("$out/pipe_d4_kernel_stats.csv")):
    print("%-70s calls %4s  avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
