#!/bin/bash
out=gpurun_out/${1:-r02_bench}
mkdir -p $out; export TMPDIR=/tmp
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.err
JDA_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --no-cpu-baseline > $out/bench_rccl_1rank.json 2> $out/bench_rccl_1rank.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --dist-backend gloo --no-cpu-baseline > $out/bench_2ranks_gloo.json 2> $out/bench_2ranks_gloo.err
timeout 900 python bench.py --workload c4 --total-images 2048 --no-cpu-baseline > $out/bench_c4_2048.json 2> $out/bench_c4.err
for f in $out/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "Mpix/s frac", round(d["roofline"]["frac"],3), "e2e", d["end_to_end_mpix_s"] and round(d["end_to_end_mpix_s"]), "n", d["n_gpus"], d["dist"], d["sharding"]["decoded_exactly_once"], d["parity"])
    if d.get("cpu_baseline"): c=d["cpu_baseline"]; print("   cpu:", c["cores"], "cores", round(c["value"]), "all", round(c["sse2_1_thread_mpix_s"]), "1T", c["scalar_no_simd_1_thread_mpix_s"] and round(c["scalar_no_simd_1_thread_mpix_s"]), "scalar; scaling", round(c["scaling_all_over_1"],2), c["cpu_model"])
    print("   placement", d["sharding"]["host_placement"])
except Exception as e:
    print("$f FAILED", e); print(open("$f".replace(".json",".err")).read()[-1500:])
PY
done
