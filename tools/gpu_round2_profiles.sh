#!/bin/bash
# Round-2 evidence in one GPU call: tools/gpu_round2_profiles.sh [tag]  -> gpurun_out/<tag>/ (summaries are copied to profiles/ by hand)
tag=${1:-r02_final}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $out/pytest_gpu.txt
# --- the metric workload: default bench line, kernel trace of the same command, HBM traffic counters (separate passes)
python bench.py > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o kt -- python $R/bench.py --no-parity --no-cpu-baseline --e2e-batches 0 > /dev/null 2>&1)
B="--steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out -o fetch -- python $R/bench.py $B > /dev/null 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out -o write -- python $R/bench.py $B > /dev/null 2>&1)
# --- the other BASELINE configs and the high-bitrate case: bench line + kernel trace (+ traffic for C3 and C5)
cfg() { name=$1; shift
  python bench.py --no-cpu-baseline --e2e-batches 0 "$@" > $out/bench_$name.json 2> $out/bench_$name.err
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o kt_$name -- python $R/bench.py --no-parity --no-cpu-baseline --e2e-batches 0 "$@" > /dev/null 2>&1)
}
pmc() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out -o fetch_$name -- python $R/bench.py $B "$@" > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out -o write_$name -- python $R/bench.py $B "$@" > /dev/null 2>&1)
}
cfg c2 --batch 1024 --width 1280 --height 720 --distinct 4
cfg c3 --batch 256 --subsampling 4:4:4
cfg c5 --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8
cfg q98 --quality 98
cfg q95 --quality 95
cfg c422 --subsampling 4:2:2
pmc c3 --batch 64 --subsampling 4:4:4
pmc c5 --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8
# --- SQ counters of the metric kernel
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1)); (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o sq_$i -- python $R/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 > /dev/null 2>&1)
done
# --- the pipeline: throughput lines and kernel table
: > $out/pipe.txt
for c in "--depth 1" "--depth 2" "--depth 3" "--depth 2 --restart-rows 1" "--depth 2 --width 1920 --height 1080 --batch 256 --batches 8" "--depth 2 --width 1280 --height 720 --batch 512 --batches 8"; do timeout 300 python tools/pipeline_bench.py $c 2>/dev/null | tail -1 >> $out/pipe.txt; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o pipe -- python $R/tools/pipeline_bench.py --depth 2 --batches 6 > /dev/null 2>&1)
# --- per-phase instruction counts (JDA_EXP_SKIP builds) for 4:2:0, 4:4:4 and gray
: > $out/phase_counts.txt
cnt() { mode=$1; tiles=$2; shift 2
  for lib in jpegdec_amd/libjpegdec_amd.so ab/lib_skip1.so ab/lib_skip3.so ab/lib_skip7.so ab/lib_skip15.so ab/lib_skip31.so; do
    d=$out/cnt_${mode}_$(basename $lib .so); rm -rf $d; mkdir -p $d
    (cd /tmp && JDA_LIBRARY=$R/$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/$d -o q -- python $R/bench.py --steps 2 --warmup 1 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 "$@" > /dev/null 2>&1)
    python - "$lib" "$d" "$mode" "$tiles" >> $out/phase_counts.txt <<PY
import csv, collections, sys, glob
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[3], sys.argv[1], {k: round(sum(v) / len(v) / float(sys.argv[4]), 1) for k, v in acc.items()}, "per tile")
PY
  done
}
if [ -f ab/lib_skip1.so ]; then
  cnt 420 $((16*6554)) --batch 16
  cnt 444 $((16*12800)) --batch 16 --subsampling 4:4:4
  cnt gray $((4*16384)) --batch 4 --width 8192 --height 8192 --subsampling gray --pixel-type gray8
fi
rm -rf $out/cnt_*
ls $out | head -80; cat $out/pytest_gpu.txt; cat $out/phase_counts.txt
