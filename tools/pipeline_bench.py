"""End-to-end pipeline throughput (jda_pipeline): files in host memory -> pixels resident in HBM, batches streamed.
usage: python tools/pipeline_bench.py [--batch 64] [--batches 12] [--depth 3] [--threads 8] [--width 4096 --height 4096]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import jpegdec_amd as J  # noqa: E402
from bench import cached_jpeg  # noqa: E402


def run(ctx, jpegs, batch, batches, depth, threads, pt=J.RGB8888, opt=0, warm=0):
    warm = max(warm, depth, 2)          # every slot has grown its arena and page-locked buffer before the clock starts
    info = J.PreparedImage(jpegs[0])
    g = info.geometry(pt, opt)
    info.close()
    pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
    img_bytes = pitch * g["canvas_h"]
    surfaces = [ctx.malloc(img_bytes * batch) for _ in range(depth)]
    pipe = J.Pipeline(ctx, max_images=batch, depth=depth, host_threads=threads)
    files = [jpegs[i % len(jpegs)] for i in range(batch)]

    # one set of C arrays per surface (what a C caller holds anyway: building them is Python's time, not the pipeline's)
    packed = [pipe.pack(files, [(base + i * img_bytes, pitch, g["canvas_w"], g["canvas_h"]) for i in range(batch)], [pt] * batch, [opt] * batch) for base in surfaces]

    def submit(k):
        return pipe.submit_packed(packed[k % depth])

    inflight = []
    t_submit = 0.0
    t0 = None
    done = 0
    for k in range(warm + batches):
        if k == warm:
            while inflight:
                pipe.wait(inflight.pop(0))
            ctx.sync()
            t0 = time.perf_counter()
        if len(inflight) == depth:
            st = pipe.wait(inflight.pop(0))
            assert all(s == 0 for s in st), st
            done += 1
        ts = time.perf_counter()
        inflight.append(submit(k))
        if k >= warm:
            t_submit += time.perf_counter() - ts
    while inflight:
        st = pipe.wait(inflight.pop(0))
        assert all(s == 0 for s in st), st
    ctx.sync()
    dt = time.perf_counter() - t0
    stats = pipe.stats
    pipe.close()
    for s in surfaces:
        ctx.free(s)
    px = g["out_w"] * g["out_h"] * batch * batches
    return {"mpix_s": px / dt / 1e6, "ms_per_image": dt / (batch * batches) * 1e3, "host_submit_ms_per_image": t_submit / (batch * batches) * 1e3,
            "batch": batch, "batches": batches, "depth": depth, "threads": threads, "stats": stats}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--subsampling", default="4:2:0")
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--restart-rows", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic files the batch cycles through (more: more segments that need a late round)")
    ap.add_argument("--no-pin", action="store_true", help="leave the process where the scheduler puts it (default: the CPUs of the GPU's NUMA node)")
    a = ap.parse_args()
    jpegs = [cached_jpeg(a.width, a.height, a.subsampling, 1234 + i, quality=a.quality, restart_rows=a.restart_rows) for i in range(a.distinct)]
    ctx = J.Context(0)
    pinned = None
    if not a.no_pin and hasattr(os, "sched_setaffinity"):     # as bench.py's ranks do: the process (and the pipeline's workers) on the GPU's NUMA node
        from jpegdec_amd.sharding import _parse_cpulist, numa_node_of_pci
        node = numa_node_of_pci(ctx.pci_bus_id())
        try:
            cpus = sorted(set(_parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())) & os.sched_getaffinity(0)) if node is not None else []
            if cpus:
                os.sched_setaffinity(0, cpus)
                pinned = {"numa_node": node, "cpus": len(cpus)}
        except Exception:
            pinned = None
    r = run(ctx, jpegs, a.batch, a.batches, a.depth, a.threads)
    r["distinct"] = a.distinct
    r["pinned"] = pinned
    print(json.dumps(r))
    ctx.close()


if __name__ == "__main__":
    main()
