#!/usr/bin/env python3
"""Profiling aid: per-phase shader-clock timeline of jda_decode_tiles (every 16th workgroup, waves 0-3).
Usage (GPU box): python tools/phase_trace.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import jpegdec_amd as J  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = J.Context(0)
lib = ctx.lib
jp = [bench.cached_jpeg(4096, 4096, "4:2:0", 1234 + i) for i in range(2)]
prep = [J.PreparedImage(j) for j in jp]
g = prep[0].geometry(J.RGB8888, 0)
pitch = g["canvas_w"] * 4
img_bytes = pitch * g["canvas_h"]
out = ctx.malloc(img_bytes * nb)
dev = [J.DeviceImage(ctx, prep[i % 2]) for i in range(nb)]
b = J.Batch(ctx, dev, [(out + i * img_bytes, pitch, g["canvas_w"], g["canvas_h"]) for i in range(nb)], [J.RGB8888] * nb, [0] * nb)
n_wg = b.stats["n_workgroups"]
n_tr = (n_wg + 15) // 16     # (the persistent grid is smaller: only its first entries are written)
buf = ctx.malloc(n_tr * 4 * 16 * 8)
ctx.memset(buf, 0, n_tr * 4 * 16 * 8)
b.decode(); ctx.sync()
lib.jda_internal_set_trace.argtypes = [C.c_void_p]
assert lib.jda_internal_set_trace(buf) == 0
b.decode(); ctx.sync()
tr = ctx.to_host(buf, n_tr * 4 * 16 * 8).view(np.uint64).reshape(n_tr, 4, 16).astype(np.int64)
lib.jda_internal_set_trace(None)
names = (["entry->setup", "P0 issue", "P0 barrier", "P1", "P2", "P3", "P4"] if os.environ.get("JDA_KERNEL", "")[:1] == "s"
         else ["A+B issue", "P1", "C issue", "P2", "D store", "P3", "P4"])
d = np.diff(tr[:, :, :8], axis=2)                # (wg, wave, 7)
ok = (tr[:, :, 7] > 0).all(axis=1)
d = d[ok]
print("traced workgroups:", d.shape[0], "of", n_tr)
tot = (tr[ok][:, :, 7] - tr[ok][:, :, 0])
for k, nm in enumerate(names):
    print("%-14s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (nm, d[:, :, k].mean(), np.median(d[:, :, k]), np.percentile(d[:, :, k], 90)))
print("%-14s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % ("TOTAL", tot.mean(), np.median(tot), np.percentile(tot, 90)))

w0 = tr[ok][:, 0, :]
if (w0[:, 11] > 0).all():
    print("inside P1 (wave 0, last tile): setup %.0f  block decode %.0f  list append %.0f cycles"
          % ((w0[:, 8] - w0[:, 10]).mean(), (w0[:, 9] - w0[:, 8]).mean(), (w0[:, 11] - w0[:, 9]).mean()))
