#!/bin/bash
# LDS-array cycles and bank conflicts of the metric kernel per phase: libraries built with -DJDA_EXP_SKIP=1 / 3 / 7 / 15 (no P4 / no P3, P4 / no P2-P4 /
# P1 alone; wrong pixels) against the product: tools/gpu_lds_by_phase.sh ab/lib_skip1.so ab/lib_skip3.so ...  -> gpurun_out/lds_phase/summary.txt
out=gpurun_out/lds_phase; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lib in jpegdec_amd/libjpegdec_amd.so "$@"; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout -k 5 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $R/$out -o $tag -- python $R/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections, os
with open("$out/summary.txt", "w") as o:
    for f in sorted(glob.glob("$out/*counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "jda_decode_tiles_persistent" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        o.write("%-28s " % os.path.basename(f).replace("_counter_collection.csv", "") + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())) + "\n")
print(open("$out/summary.txt").read())
PY
