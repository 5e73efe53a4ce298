#!/bin/bash
# the filter kernels without the scan of functions (v2, in-tree default) against the scan kernels (JDA_FILTER_V1=1): the filter's own
# test under both, the pipeline tests + fuzz under v2, every kernel's time at depth 1, end to end at depth 4
out=gpurun_out/r3_filter; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
JDA_FILTER_V1=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k marker_filter 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k marker_filter 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_ref_fixtures.py -x -q -m gpu 2>&1 | tail -1
timeout 600 python tools/gpu_fuzz_pipeline.py 30 21 | tail -1
for v in 1 0; do
  (cd /tmp && JDA_FILTER_V1=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o d1_v1is$v -- python $R/tools/pipeline_bench.py --depth 1 --batches 8 --distinct 16 > /dev/null 2>&1)
done
for rep in 1 2 3; do for v in 1 0; do echo -n "JDA_FILTER_V1=$v " >> $out/e2e.txt; JDA_FILTER_V1=$v timeout 300 python tools/pipeline_bench.py --depth 4 --batches 24 --distinct 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']))" >> $out/e2e.txt; done; done
cat $out/e2e.txt
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/d1_*kernel_stats.csv")):
    print(f)
    for r in csv.DictReader(open(f)):
        if "filter" in r["Name"]: print("  %-60s calls %5s avg_us %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
