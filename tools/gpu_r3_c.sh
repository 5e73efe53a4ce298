#!/bin/bash
# round 3: bit-parallel filter + finalize tuning + lead-in A/B
out=gpurun_out/r03_c
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
: > $out/pipe.txt
for rep in 1 2; do
  for L in 0 192 256; do
    JDA_PIPE_LEAD_BYTES=$L timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 16 2>&1 | tail -1 >> $out/pipe.txt
  done
done
JDA_PIPE_LEAD_BYTES=0 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
JDA_PIPE_LEAD_BYTES=256 timeout 300 python tools/pipeline_bench.py --depth 4 --threads 8 --batches 40 --distinct 2 2>&1 | tail -1 >> $out/pipe.txt
python - <<PY
import json
for i,l in enumerate(open("$out/pipe.txt")):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f  distinct %d depth %d rounds %d devimgs %d hostimgs %d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d.get("distinct",0), d["depth"], d["stats"]["spec_rounds_max"], d["stats"]["device_images"], d["stats"]["host_path_images"]))
PY
for L in 0 256; do
cd /tmp && JDA_PIPE_LEAD_BYTES=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o pipe_d1_L$L -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 1 --threads 8 --batches 8 --distinct 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv
print("lead bytes $L")
for r in csv.DictReader(open("$out/pipe_d1_L${L}_kernel_stats.csv")):
    print("%-70s calls %4s  avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
