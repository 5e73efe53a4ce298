#!/bin/bash
# quick A/B on the GPU box: three default bench runs (kernel time) + VALU/SALU/LDS instruction counts
export TMPDIR=/tmp
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('us/image %.2f  frac %.3f' % (d['roofline']['kernel_ms_per_launch']*1000/64, d['roofline']['frac']))"; done
mkdir -p gpurun_out/quick
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/quick -o q -- python bench.py --steps 2 --warmup 1 --batch 16 --no-parity --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/quick/q_counter_collection.csv")):
    if "jda_decode" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v) / 16 / 6554, 1) for k, v in acc.items()}, "per 4:2:0 tile")
PY
