#!/bin/bash
# HBM traffic of the metric kernel on the other layouts (separate --pmc passes, no trace domains): 4:4:4 (c3), 8192x8192 gray (c5), q98
out=gpurun_out/r3_pmc_cfg; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="--steps 6 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs"
pmc() { name=$1; shift
  python bench.py --no-configs --no-cpu-baseline --e2e-batches 0 --steps 30 "$@" > $out/bench_$name.json 2>/dev/null
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out -o fetch_$name -- python $R/bench.py $B "$@" > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out -o write_$name -- python $R/bench.py $B "$@" > /dev/null 2>&1)
}
pmc c3 --batch 64 --subsampling 4:4:4
pmc c5 --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8
pmc q98 --batch 32 --quality 98
python - <<PY
import csv, json
for name in ("c3", "c5", "q98"):
    b = json.loads(open("$out/bench_%s.json" % name).read().strip().splitlines()[-1])
    def per_launch(f, c):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "jda_decode_tiles" in r["Kernel_Name"] and r["Counter_Name"] == c]
        return sum(v) / len(v), len(v)
    f, n = per_launch("$out/fetch_%s_counter_collection.csv" % name, "FETCH_SIZE")
    w, _ = per_launch("$out/write_%s_counter_collection.csv" % name, "WRITE_SIZE")
    alg = b["roofline"]["algorithmic_bytes_per_launch"]
    o = {"workload": b["config"]["workload"], "launches_sampled": n, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
         "hbm_bytes_per_launch_fetch_x2": (2 * f + w) * 1024, "algorithmic_bytes_per_launch": alg, "ratio": (2 * f + w) * 1024 / alg,
         "write_over_output_bytes": None, "mpix_s": b["value"], "frac": b["roofline"]["frac"],
         "note": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes (KB); FETCH doubled: the gfx950 correction of MI355X_MICROARCH.md"}
    json.dump(o, open("$out/r03_%s_pmc_traffic.json" % name, "w"), indent=1)
    print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in o.items() if k != "note"})
PY
