"""The reference's own test fixtures (binary copies under tests/golden/ref/, hashes of what the real reference makes of
them in tests/golden/ref/ref_golden.json; generator: tests/golden/ref/make_ref_golden.py)."""
import functools
import json
import os

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref")

# fixture name -> path below the reference root (C array headers / plain files)
HEADER_SOURCES = {
    "tulips": "test_images/tulips.h", "zebra": "test_images/zebra.h", "st_peters": "test_images/st_peters.h",
    "sciopero": "test_images/sciopero.h", "thumb_test": "test_images/thumb_test.h",
    "croptest": "examples/crop_area/croptest.h",
    "corrupt1": "MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt1.h", "corrupt2": "MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt2.h",
    "corrupt3": "MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt3.h", "corrupt4": "MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt4.h",
    "corrupt5": "MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt5.h",
}
FILE_SOURCES = {"demo": "demo.jpg", "perf": "perf.jpg", "squirrel_dither": "squirrel_dither.jpg"}

# the fixtures the reference decodes completely (corrupt5 is "FPE2": a valid 128x85 image once its header survives)
GOOD = ("tulips", "zebra", "st_peters", "sciopero", "croptest", "demo", "perf", "corrupt5")
BIG = ("squirrel_dither",)                                      # 3596x2840: a few modes only
REJECTED_AT_OPEN = ("corrupt1", "corrupt4")                     # JPEGParseInfo fails (openFLASH returns 0)
FAIL_IN_DECODE = ("corrupt2", "corrupt3", "thumb_test")         # decode() returns 0 with JPEG_DECODE_ERROR

CROP_INO = (120, 65, 119, 110)                                  # examples/crop_area/crop_area.ino:92


@functools.lru_cache(maxsize=None)
def ref_jpeg(name: str) -> bytes:
    return open(os.path.join(REF_DIR, name + ".jpg"), "rb").read()


@functools.lru_cache(maxsize=None)
def ref_golden() -> dict:
    return json.load(open(os.path.join(REF_DIR, "ref_golden.json")))


def frame_of(r):
    """The image area (H >> s x W >> s pixels) of a RefDecoder.decode_cb result: (frame, w, h)."""
    inf, sh, bpp = r["info"], r["scale_shift"], r["bpp"]
    adj = (1 << sh) - 1
    w, h = (inf["width"] + adj) >> sh, (inf["height"] + adj) >> sh
    return r["canvas"][:h, : w * bpp], w, h


def modes_of(name):
    """(pixel type, options) pairs recorded for a fixture."""
    return [tuple(int(v) for v in k.split(":")) for k in sorted(ref_golden()[name]["frames"])]
