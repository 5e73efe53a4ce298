#!/bin/bash
# the default bench line and the kernel trace of the same command: tools/gpu_bench_default.sh [tag]
tag=${1:-r02_bench_default}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o kt -- python $R/bench.py --no-parity --no-cpu-baseline --e2e-batches 0 > /dev/null 2>&1)
rm -f $out/*_agent_info.csv $out/*_domain_stats.csv $out/*_kernel_trace.csv
python - <<PY
import json, csv
d = json.load(open("$out/bench_default.json"))
print("value %.0f frac %.4f kernel ms %.4f e2e %.0f" % (d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["end_to_end_mpix_s"]))
for r in csv.DictReader(open("$out/kt_kernel_stats.csv")):
    if "decode" in r["Name"]: print("rocprof: calls %s avg %.1f us min %.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
