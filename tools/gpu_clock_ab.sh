#!/bin/bash
# cycles, instructions and shader clock of the metric kernel for several library builds: tools/gpu_clock_ab.sh ab/lib_x.so ...  (the in-tree library rides along)
out=gpurun_out/clock_ab; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lib in "$@" jpegdec_amd/libjpegdec_amd.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout -k 5 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out -o $tag -- python $R/bench.py --steps 12 --warmup 4 --batch 64 --ramp-ms 300 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
  python - <<PY
import csv, collections
acc = collections.defaultdict(list); dur = []
for r in csv.DictReader(open("$out/${tag}_counter_collection.csv")):
    if "jda_decode_tiles_persistent" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES": dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v[2:]) / len(v[2:]) for k, v in acc.items()}
d = sum(dur[2:]) / len(dur[2:])
print("%-28s launches %d  duration %.1f us  CU cycles %.0f  clock %.3f GHz  VALU insts/tile %.1f  VALU busy %.3f" % ("$tag", len(dur), d / 1e3, m["SQ_BUSY_CU_CYCLES"] / 256, m["SQ_BUSY_CU_CYCLES"] / 256 / d, m["SQ_INSTS_VALU"] / (64 * 6656), 4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]))
PY
done
