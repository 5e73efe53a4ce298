#!/usr/bin/env python3
"""Golden vectors of the reference's OWN fixtures, made by the REAL reference (oracle/_ref, scalar integer build).

Run in the build container (needs /root/reference):   python tests/golden/ref/make_ref_golden.py

1. extracts the JPEG bytes of the reference's test fixtures into tests/golden/ref/*.jpg (binary copies of test
   VECTORS, not of source code):
     test_images/{tulips,zebra,st_peters,sciopero,thumb_test}.h, examples/crop_area/croptest.h,
     MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt1-5.h, demo.jpg, perf.jpg, squirrel_dither.jpg
2. records in ref_golden.json what the unmodified reference does with each of them: the header fields, and for
   every pixel type x option the decode verdict (return value, getLastError(), number of draw callbacks) and the
   sha256 prefix of the delivered frame; the cropped decode of croptest with crop_area.ino's rectangle; the EXIF
   thumbnail of thumb_test (reference test 10); framebuffer mode for the fixtures whose width is an MCU multiple.
The GPU box has no /root/reference: tests/test_gpu_ref_fixtures.py compares the HIP path with these hashes (and
with oracle/_ref itself where the prebuilt .so travelled along).
"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

from oracle.loader import EXIF_THUMBNAIL, RefDecoder, digest, load_c_array_header  # noqa: E402
from tests.cases import OPTIONS, PIXEL_TYPES  # noqa: E402
from tests.ref_fixtures import CROP_INO, HEADER_SOURCES, FILE_SOURCES, frame_of  # noqa: E402

REF = "/root/reference"


def main():
    ref = RefDecoder(simd=False)
    for name, rel in HEADER_SOURCES.items():
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(load_c_array_header(os.path.join(REF, rel)))
    for name, rel in FILE_SOURCES.items():
        shutil.copyfile(os.path.join(REF, rel), os.path.join(HERE, name + ".jpg"))
        os.chmod(os.path.join(HERE, name + ".jpg"), 0o644)
    out = {}
    for name in sorted(list(HEADER_SOURCES) + list(FILE_SOURCES)):
        jpeg = open(os.path.join(HERE, name + ".jpg"), "rb").read()
        inf = ref.info(jpeg)
        entry = {"jpeg_sha": digest(jpeg), "jpeg_len": len(jpeg), "info": inf, "frames": {}}
        if inf["ok"] and inf["subsample"] in (0, 0x11, 0x12, 0x21, 0x22):
            big = inf["width"] * inf["height"] > (1 << 22)
            for pt in PIXEL_TYPES:
                for opt in OPTIONS:
                    if big and (pt, opt) not in ((2, 0), (0, 2), (3, 8)):
                        continue                                    # the 10-Mpixel squirrel / 12-Mpixel thumb_test header: three modes
                    r = ref.decode_cb(jpeg, pt, opt, want_log=True)
                    fr = {"rc": r["rc"], "err": r["last_error"], "draw_calls": r["n_calls"]}
                    if r["rc"] == 1:
                        frame, w, h = frame_of(r)
                        fr.update(sha=digest(frame), w=w, h=h, bpp=r["bpp"], log_sha=digest(r["log"]))
                    entry["frames"]["%d:%d" % (pt, opt)] = fr
        out[name] = entry
        print(name, len(jpeg), inf["width"], inf["height"], hex(inf["subsample"]), len(entry["frames"]))
    # crop_area.ino:92 -- setCropArea(120, 65, 119, 110)
    jpeg = open(os.path.join(HERE, "croptest.jpg"), "rb").read()
    crops = {}
    for pt in (0, 1, 2, 3):
        r = ref.decode_cb(jpeg, pt, 0, crop=CROP_INO, want_log=True)
        crops[str(pt)] = {"rc": r["rc"], "err": r["last_error"], "draw_calls": r["n_calls"], "sha": digest(r["canvas"]), "log_sha": digest(r["log"])}
    out["croptest"]["crop_ino"] = crops
    # reference test 10: the EXIF thumbnail of thumb_test is a 320x240 JPEG
    jpeg = open(os.path.join(HERE, "thumb_test.jpg"), "rb").read()
    th = {}
    for pt in (0, 1, 2, 3):
        for opt in (0, 2, 4, 8):
            r = ref.decode_cb(jpeg, pt, opt | EXIF_THUMBNAIL, canvas_shape=(256, 336), want_log=True)
            th["%d:%d" % (pt, opt)] = {"rc": r["rc"], "err": r["last_error"], "draw_calls": r["n_calls"], "size_after": list(r["size_after"]),
                                       "sha": digest(r["canvas"]), "log_sha": digest(r["log"])}
    out["thumb_test"]["exif_thumbnail"] = th
    with open(os.path.join(HERE, "ref_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
