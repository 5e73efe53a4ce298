#!/bin/bash
# Round-3 evidence for the metric kernel in one GPU call: bench line, kernel trace + stats of the same command, HBM traffic counters
# (separate --pmc passes), SQ counters.  tools/gpu_r3_profiles.sh [tag] -> gpurun_out/<tag>/ (tools/pmc_summary.py copies into profiles/)
tag=${1:-r03_prof}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --no-configs > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o kt -- python $R/bench.py --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
B="--steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out -o fetch -- python $R/bench.py $B > /dev/null 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out -o write -- python $R/bench.py $B > /dev/null 2>&1)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1)); (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o sq_$i -- python $R/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
done
# the pipeline at depth 1 (nothing overlaps: every kernel's own time) and at depth 4, 16 distinct files per batch
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o pipe_d1 -- python $R/tools/pipeline_bench.py --depth 1 --batches 8 --distinct 16 > /dev/null 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o pipe_d4 -- python $R/tools/pipeline_bench.py --depth 4 --batches 16 --distinct 16 > /dev/null 2>&1)
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/sq_*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/sq_counters.txt", "w") as o:
    o.write("rocprofv3 --pmc (two passes) -- python bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0: jda_decode_tiles_persistent<2,true,1,0>, mean per launch of 16 x 4096x4096\n")
    for k, v in sorted(acc.items()):
        o.write("%-26s %.4g\n" % (k, sum(v) / len(v)))
    if "SQ_ACTIVE_INST_VALU" in acc and "SQ_BUSY_CYCLES" in acc:
        pass
print(open("$out/sq_counters.txt").read())
PY
ls $out | head -40
