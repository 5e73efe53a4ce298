"""one call through the product's JPEGDEC class (tests/libjpegdec_class_shim.so): python tools/probe_class_case.py image pt opt max_mcus x y [cx cy cw ch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.loader import RefDecoder
from tests.cases import jpeg_for
from tests.ref_fixtures import ref_jpeg
a = sys.argv
name = a[1]
j = ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name)
crop = [int(v) for v in a[7:11]] if len(a) >= 11 else None
os.environ["JDA_CLASS_TRACE"] = "1"
p = RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))
r = p.decode_cb(j, int(a[2]), int(a[3]), max_mcus=int(a[4]), xoff=int(a[5]), yoff=int(a[6]), crop=crop, want_log=True, used_only=True)
print("rc", r["rc"], "err", r["last_error"], "calls", r["n_calls"], "log", r["log"][:3].tolist() if r["log"] is not None else None)
