#!/bin/bash
# VALU instructions and active lanes per phase of the decode kernel, MEASURED: the library built without P4 / + P3 / + P2 / + lists / + P1
# (tools/phase_count_libs.sh's JDA_EXP_SKIP builds, here ab/lib_skip{1,3,7,15,31}.so) under rocprofv3 --pmc; differences of consecutive
# builds are the phases.  Metric batch (16 images) and the photographs leg.  -> gpurun_out/phase_lanes/table.txt
out=gpurun_out/phase_lanes; mkdir -p $out; export TMPDIR=/tmp
for lib in jpegdec_amd/libjpegdec_amd.so ab/lib_skip1.so ab/lib_skip3.so ab/lib_skip7.so ab/lib_skip15.so ab/lib_skip31.so; do
  tag=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout -k 5 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $GRAFT_REPO_ROOT/$out -o m_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs > /dev/null 2>&1)
  (cd /tmp && JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout -k 5 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $GRAFT_REPO_ROOT/$out -o p_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --configs photos > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections, os
out = "$out"
def load(prefix, tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    f = os.path.join(out, "%s_%s_counter_collection.csv" % (prefix, tag))
    if not os.path.exists(f): return {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "jda_decode_tiles_persistent" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
tags = ["libjpegdec_amd", "lib_skip1", "lib_skip3", "lib_skip7", "lib_skip15", "lib_skip31"]
names = ["all", "- P4", "- P3", "- P2", "- lists", "- P1"]
with open(os.path.join(out, "table.txt"), "w") as o:
    for prefix, what in (("m", "metric batch (16 x 4096x4096 4:2:0 -> RGB8888)"), ("p", "photographs leg (tulips, zebra, st_peters, perf tiled)")):
        o.write("== %s: per kernel, per launch: SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU) = lanes active\n" % what)
        rows = {}
        for t, n in zip(tags, names):
            for k, c in load(prefix, t).items():
                if "SQ_INSTS_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c:
                    # the photographs leg launches the same kernel for several files: per-launch values kept in order of appearance
                    rows.setdefault(k, []).append((n, sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]), sum(c["SQ_THREAD_CYCLES_VALU"]) / len(c["SQ_THREAD_CYCLES_VALU"])))
        for k, lst in rows.items():
            o.write("  %s\n" % k)
            prev = None
            for n, iv, tc in lst:
                line = "    %-8s insts %.4g  lanes %.3f" % (n, iv, tc / (64 * iv) if iv else 0)
                if prev:
                    di, dt = prev[0] - iv, prev[1] - tc
                    line += "   | the phase taken out: insts %.4g (%.1f %% of all)  lanes %.3f" % (di, 100 * di / lst[0][1], dt / (64 * di) if di else 0)
                o.write(line + "\n")
                prev = (iv, tc)
print(open(os.path.join(out, "table.txt")).read())
PY
