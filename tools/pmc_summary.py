#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/gpu_profile.sh into the files kept under profiles/.

usage: python tools/pmc_summary.py gpurun_out/<tag> profiles/<prefix>
Writes <prefix>_kernel_stats.csv (copy of the --kernel-trace --stats summary), <prefix>_bench_default.json
and <prefix>_pmc_traffic.json (HBM bytes per launch/image from the FETCH_SIZE and WRITE_SIZE passes, with the
gfx950 FETCH_SIZE correction of /opt/skills/guides/MI355X_MICROARCH.md)."""
import csv
import json
import shutil
import sys


def counter_per_launch(path, kernel_substr, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return vals


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    bench = json.loads(open(f"{src}/bench_default.json").read().strip().splitlines()[-1])
    batch = bench["config"].get("images_per_gpu_per_step") or 64
    kern = "jda_decode_tiles"
    fetch = counter_per_launch(f"{src}/fetch_counter_collection.csv", kern, "FETCH_SIZE")
    write = counter_per_launch(f"{src}/write_counter_collection.csv", kern, "WRITE_SIZE")
    f_kb = sum(fetch) / len(fetch)
    w_kb = sum(write) / len(write)
    out = {
        "command": "tools/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, no trace domains) "
                   "--output-format csv -- python bench.py --steps 10 --warmup 2 --ramp-ms 0 --no-parity --no-cpu-baseline",
        "kernel": "jda_decode_tiles_persistent<2,true,1>",
        "launches_sampled": len(fetch),
        "images_per_launch": batch,
        "workload": bench["config"]["workload"],
        "FETCH_SIZE_KB_per_launch": f_kb,
        "WRITE_SIZE_KB_per_launch": w_kb,
        "write_bytes_per_image": w_kb * 1024 / batch,
        "fetch_bytes_per_image_raw": f_kb * 1024 / batch,
        "fetch_bytes_per_image_corrected_x2": 2 * f_kb * 1024 / batch,
        "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide "
                "coalesced reads by 2x -> doubled.  WRITE_SIZE equals the output surfaces (64 MiB/image): every output "
                "byte is written once, nothing else is written.",
        "hbm_bytes_per_image": (2 * f_kb + w_kb) * 1024 / batch,
        "algorithmic_bytes_per_image": bench["roofline"].get("algorithmic_bytes_per_launch", 0) / batch or None,
    }
    json.dump(out, open(f"{prefix}_pmc_traffic.json", "w"), indent=1)
    shutil.copy(f"{src}/kt_kernel_stats.csv", f"{prefix}_kernel_stats.csv")
    json.dump(bench, open(f"{prefix}_bench_default.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
