#!/bin/bash
# The round's evidence for the metric kernel in ONE GPU call (gpurun -- 'bash tools/gpu_profile.sh [tag] [prefix]'):
#   bench line of the default run, kernel trace + stats of the same command, HBM traffic counters (separate --pmc passes, no trace
#   domains: MI355X_MICROARCH.md), SQ counters -> gpurun_out/<tag>/, and the summaries the judge reads -> profiles/<prefix>_*:
#   <prefix>_bench_default.json, <prefix>_kernel_stats.csv, <prefix>_pmc_traffic.json (names the hash of the kernel sources it was
#   measured at: bench.py reports roofline.traffic from it only while the tree still has those sources), <prefix>_sq_counters.txt
tag=${1:-r06_prof}; prefix=${2:-r06}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o kt -- python $R/bench.py --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
B="--steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs"
(cd /tmp && timeout -k 5 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out -o fetch -- python $R/bench.py $B > /dev/null 2>&1)
(cd /tmp && timeout -k 5 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out -o write -- python $R/bench.py $B > /dev/null 2>&1)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1)); (cd /tmp && timeout -k 5 90 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o sq_$i -- python $R/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
import bench
acc = collections.defaultdict(list)
for f in glob.glob("$out/sq_*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "jda_decode_tiles_persistent" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = 16 * 256 * 26          # 16 x 4096x4096 4:2:0: 256 MCU rows x 26 tiles (25 of 10 MCUs + one of 6)
with open("$out/sq_counters.txt", "w") as o:
    o.write("rocprofv3 --pmc (two passes) -- python bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0: jda_decode_tiles_persistent<2, 1, 1, 0>, mean per launch of 16 x 4096x4096 (%d tiles)\n" % tiles)
    for k, v in sorted(acc.items()):
        o.write("%-26s %.6g\n" % (k, sum(v) / len(v)))
    if "SQ_INSTS_VALU" in acc:
        # SQ_INSTS_VALU counts per SE-sampled wave on this part: reported as is; per tile = / tiles when the counter covers every wave
        o.write("SQ_INSTS_VALU / tile       %.1f\n" % (sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"]) / tiles))
    if "SQ_THREAD_CYCLES_VALU" in acc and "SQ_INSTS_VALU" in acc:
        o.write("lanes active per VALU instruction  %.3f\n" % (sum(acc["SQ_THREAD_CYCLES_VALU"]) / len(acc["SQ_THREAD_CYCLES_VALU"]) / (64.0 * sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"]))))
print(open("$out/sq_counters.txt").read())
b = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
def per_launch(f, c):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "jda_decode_tiles" in r["Kernel_Name"] and r["Counter_Name"] == c]
    return sum(v) / len(v), len(v)
fk, n = per_launch("$out/fetch_counter_collection.csv", "FETCH_SIZE")
wk, _ = per_launch("$out/write_counter_collection.csv", "WRITE_SIZE")
batch = b["config"]["images_per_gpu_per_step"]
o = {"command": "tools/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, no trace domains) --output-format csv -- python bench.py --steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs",
     "kernel": "jda_decode_tiles_persistent<2, 1, 1, 0>", "kernel_sources_sha16": bench.kernel_sources_sha(),
     "launches_sampled": n, "images_per_launch": batch, "workload": b["config"]["workload"],
     "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "write_bytes_per_image": wk * 1024 / batch,
     "fetch_bytes_per_image_raw": fk * 1024 / batch, "fetch_bytes_per_image_corrected_x2": 2 * fk * 1024 / batch,
     "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled",
     "hbm_bytes_per_image": (2 * fk + wk) * 1024 / batch, "algorithmic_bytes_per_image": b["roofline"]["algorithmic_bytes_per_launch"] / batch}
o["ratio_traffic_over_algorithmic"] = o["hbm_bytes_per_image"] / o["algorithmic_bytes_per_image"]
json.dump(o, open("$out/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in o.items() if k not in ("command", "note")}, indent=1))
PY
mkdir -p profiles
cp $out/pmc_traffic.json profiles/${prefix}_pmc_traffic.json
cp $out/kt_kernel_stats.csv profiles/${prefix}_kernel_stats.csv
cp $out/sq_counters.txt profiles/${prefix}_sq_counters.txt
python -c "import json; json.dump(json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]), open('profiles/${prefix}_bench_default.json', 'w'), indent=1)"
# (profiles/ written on the GPU box does not travel back: the same files are under gpurun_out/$tag -- copy them into profiles/ there)
cp profiles/${prefix}_pmc_traffic.json profiles/${prefix}_kernel_stats.csv profiles/${prefix}_sq_counters.txt profiles/${prefix}_bench_default.json $out/ 2>/dev/null
ls $out | head -40
