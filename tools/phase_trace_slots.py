import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench, jpegdec_amd as J
nb = 64
ctx = J.Context(0); lib = ctx.lib
jp = [bench.cached_jpeg(4096, 4096, "4:2:0", 1234 + i) for i in range(2)]
prep = [J.PreparedImage(j) for j in jp]
g = prep[0].geometry(J.RGB8888, 0); pitch = g["canvas_w"] * 4; img_bytes = pitch * g["canvas_h"]
out = ctx.malloc(img_bytes * nb)
dev = [J.DeviceImage(ctx, prep[i % 2]) for i in range(nb)]
b = J.Batch(ctx, dev, [(out + i * img_bytes, pitch, g["canvas_w"], g["canvas_h"]) for i in range(nb)], [J.RGB8888] * nb, [0] * nb)
n_wg = b.stats["n_workgroups"]; n_tr = (n_wg + 15) // 16
buf = ctx.malloc(n_tr * 4 * 16 * 8); ctx.memset(buf, 0, n_tr * 4 * 16 * 8)
b.decode(); ctx.sync()
lib.jda_internal_set_trace.argtypes = [C.c_void_p]
assert lib.jda_internal_set_trace(buf) == 0
b.decode(); ctx.sync()
tr = ctx.to_host(buf, n_tr * 4 * 16 * 8).view(np.uint64).reshape(n_tr, 4, 16).astype(np.int64)
lib.jda_internal_set_trace(None)
t = tr[:, :, :8].reshape(-1, 8)
t = t[(t > 0).all(axis=1)]
print("traced waves:", t.shape[0])
d = np.diff(t, axis=1)
for k in range(7):
    print("slot %d->%d mean %8.0f p50 %8.0f p90 %8.0f" % (k, k + 1, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90)))
print("total", (t[:, 7] - t[:, 0]).mean())
