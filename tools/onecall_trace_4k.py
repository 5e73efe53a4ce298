import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import jpegdec_amd as J
from bench import cached_jpeg
ctx = J.Context(0)
jp = cached_jpeg(4096, 4096, "4:2:0", 1234)
out = None
for i in range(8):
    sys.stderr.write("---- iteration %d\n" % i)
    rc, out, g = J.decode_to_host(ctx, jp, J.RGB565_LE, 0, out=out)
