import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
import jpegdec_amd as J
ctx = J.Context(0)
for (w, h, n) in ((1280, 720, 1024), (1920, 1080, 1024)):
    jp = [bench.cached_jpeg(w, h, "4:2:0", 1234 + i) for i in range(8)]
    for depth in (3, 4, 5, 6, 8):
        r = bench.e2e_config_leg(J, ctx, "x", jp, n, J.RGB8888, 8, depth=depth, batches=16)
        print(w, h, "depth", depth, round(r["mpix_s"]), "ms/batch %.3f" % r["ms_per_batch"], "submit %.2f" % r["host_submit_ms_per_batch"], flush=True)
jp = [bench.cached_jpeg(4096, 4096, "4:2:0", 1234 + i) for i in range(16)]
for depth in (4, 6, 8):
    r = bench.e2e_config_leg(J, ctx, "x", jp, 64, J.RGB8888, 8, depth=depth, batches=24)
    print(4096, "depth", depth, round(r["mpix_s"]), "ms/batch %.3f" % r["ms_per_batch"], flush=True)
