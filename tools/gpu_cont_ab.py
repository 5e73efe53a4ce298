#!/usr/bin/env python3
"""P1 in chunks (continuation entries) against P1 block by block, same box, same resident images, interleaved:
python tools/gpu_cont_ab.py [device_prescan 0|1] -> kernel ms per step of each workload in both modes + parity of the chunked decode."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jpegdec_amd as J  # noqa: E402
from bench import cached_jpeg, check_against_reference  # noqa: E402

dev_prescan = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
lib = J.load_library()
ctx = J.Context(0)
d = os.path.join(ROOT, "tests", "golden", "ref")
photo = {n: open(os.path.join(d, n + ".jpg"), "rb").read() for n in ("tulips", "zebra", "st_peters", "perf")}
work = [("photos x2048", [photo[n] for n in ("tulips", "zebra", "st_peters", "perf")], 2048),
        ("tulips x2048", [photo["tulips"]], 2048), ("zebra x4096", [photo["zebra"]], 4096), ("perf x1024", [photo["perf"]], 1024),
        ("q98 x32", [cached_jpeg(4096, 4096, "4:2:0", 1234 + i, quality=98) for i in range(2)], 32),
        ("metric x32", [cached_jpeg(4096, 4096, "4:2:0", 1234 + i) for i in range(2)], 32)]
for name, jpegs, n in work:
    files = [jpegs[i % len(jpegs)] for i in range(n)]
    prep = {"blocks": J.prepare_batch(files, device_prescan=dev_prescan, threads=8, flags=J.PREPARE_CONT_NEVER),
            "chunks": J.prepare_batch(files, device_prescan=dev_prescan, threads=8, flags=J.PREPARE_CONT_ALWAYS)}
    prepared = prep["blocks"]
    geos = [p.geometry(J.RGB8888, 0) for p in prepared[: len(jpegs)]]
    pit = [(g["canvas_w"] * 4 + 15) & ~15 for g in geos]
    size = [pit[k] * geos[k]["canvas_h"] for k in range(len(jpegs))]
    offs, total = [], 0
    for i in range(n):
        offs.append(total); total += (size[i % len(jpegs)] + 255) & ~255
    base = ctx.malloc(total)
    dev = {m: J.upload_batch(ctx, prep[m]) for m in prep}
    devimgs = dev["blocks"]
    outs = [(base + offs[i], pit[i % len(jpegs)], geos[i % len(jpegs)]["canvas_w"], geos[i % len(jpegs)]["canvas_h"]) for i in range(n)]
    batches = {}
    for mode in ("blocks", "chunks"):
        batches[mode] = J.Batch(ctx, dev[mode], outs, [J.RGB8888] * n, [0] * n)
    res = {"blocks": [], "chunks": []}
    for rep in range(3):
        for mode in ("blocks", "chunks"):
            b = batches[mode]
            for _ in range(6):
                b.decode()
            ctx.sync(); ctx.timer_start()
            for _ in range(20):
                b.decode()
            ctx.timer_stop(); ctx.sync()
            res[mode].append(ctx.timer_elapsed_ms() / 20)
    batches["chunks"].decode(); ctx.sync()
    sums = ctx.checksums(outs[: len(jpegs)], [geos[k]["canvas_w"] * 4 for k in range(len(jpegs))])
    ok = all(check_against_reference(J, ctx, jpegs[k], J.RGB8888, 0, base + offs[k], size[k], pit[k], geos[k], sums[k])["bit_exact"] for k in range(len(jpegs)))
    mb, mc = min(res["blocks"]), min(res["chunks"])
    print("%-14s index %-6s  blocks %.4f ms  chunks %.4f ms  x%.3f  (launch lists: %d -> %d)  chunked decode bit-exact: %s" % (
        name, "device" if devimgs[0].prescan_on_device else "host", mb, mc, mb / mc, batches["blocks"].stats["n_launches"], batches["chunks"].stats["n_launches"], ok), flush=True)
    for b in batches.values():
        b.close()
    for m in dev:
        for x in dev[m]:
            x.close()
        for p in prep[m]:
            p.close()
    ctx.free(base)
