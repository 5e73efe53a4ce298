#!/usr/bin/env python3
"""Profiling aid: when does every wave of the persistent decode kernel start and finish (100 MHz wall clock)?
Shows how evenly the static split of the batch's tiles loads the CUs.  Usage (GPU box): python tools/wg_balance.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import jpegdec_amd as J  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
n = 1024 * 16 * 2
buf = ctx.malloc(n * 8)
ctx.memset(buf, 0, n * 8)
b.decode(); ctx.sync()
lib.jda_internal_set_wgtrace.argtypes = [C.c_void_p]
assert lib.jda_internal_set_wgtrace(buf) == 0
b.decode(); ctx.sync()
tr = ctx.to_host(buf, n * 8).view(np.uint64).reshape(-1, 16, 2).astype(np.int64)
lib.jda_internal_set_wgtrace(None)
ok = tr[:, 0, 0] > 0
tr = tr[ok]
t0 = tr[:, :, 0].min()
start = (tr[:, :, 0] - t0) / 100.0        # us
end = (tr[:, :, 1] - t0) / 100.0
print("workgroups: %d   kernel span %.1f us" % (tr.shape[0], end.max()))
print("wave start  : min %.1f  p50 %.1f  max %.1f us" % (start.min(), np.median(start), start.max()))
print("wave finish : min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
wg_end = end.max(axis=1)
print("workgroup finish: min %.1f  mean %.1f  max %.1f us  -> mean CU busy fraction %.3f" % (wg_end.min(), wg_end.mean(), wg_end.max(), (end - start).mean() / end.max()))
h, e = np.histogram(wg_end, bins=12)
for c, lo, hi in zip(h, e[:-1], e[1:]):
    print("  %7.1f - %7.1f us : %d" % (lo, hi, c))
