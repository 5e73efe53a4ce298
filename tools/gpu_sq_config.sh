#!/bin/bash
# SQ counters of the decode kernel on another workload: tools/gpu_sq_config.sh "<bench.py workload flags>" TILES_PER_IMAGE BATCH
# e.g. tools/gpu_sq_config.sh "--subsampling gray --pixel-type gray8 --width 8192 --height 8192" 16384 4
flags=$1; tiles=$2; batch=${3:-4}
out=gpurun_out/sq_cfg; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); (cd /tmp && timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o sq_$i -- python $R/bench.py --steps 2 --warmup 1 --batch $batch --distinct 2 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs $flags > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/sq_*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "jda_decode_tiles_persistent" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"][:60]
tiles = $tiles * $batch
print("$flags:", name, "tiles per launch", tiles)
for k, v in sorted(acc.items()):
    print("%-26s %.6g  per tile %.1f" % (k, sum(v) / len(v), sum(v) / len(v) / tiles))
if "SQ_THREAD_CYCLES_VALU" in acc: print("lanes active per VALU instruction %.3f" % (sum(acc["SQ_THREAD_CYCLES_VALU"]) / len(acc["SQ_THREAD_CYCLES_VALU"]) / (64.0 * sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"]))))
PY
