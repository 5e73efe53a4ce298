import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import jpegdec_amd as J
ctx = J.Context(0)
for name in ("tulips", "zebra"):
    jpeg = open("tests/golden/ref/%s.jpg" % name, "rb").read()
    out = None
    for i in range(30):
        if i == 29: sys.stderr.write("---- %s %d bytes\n" % (name, len(jpeg)))
        os.environ["X"] = "1"
        rc, out, g = J.decode_to_host(ctx, jpeg, J.RGB565_LE, 0, out=out)
    t0 = time.perf_counter()
    for i in range(200):
        rc, out, g = J.decode_to_host(ctx, jpeg, J.RGB565_LE, 0, out=out)
    print(name, "decode_to_host %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
