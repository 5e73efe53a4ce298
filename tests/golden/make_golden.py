#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container (needs oracle/_ref, i.e. /root/reference at build time):
    python tests/golden/make_golden.py
For every case of tests/cases.py it freezes the JPEG bytes (so the tests do not depend on the
Pillow version that encoded them) and records, for every pixel type x option, the sha256 prefix of
the frame the unmodified reference (scalar integer build, -DNO_SIMD) produces through its draw
callbacks, plus the number of callbacks.  The GPU box has no /root/reference: tests there compare
against these committed hashes (and against oracle/_ref when the prebuilt .so travelled along).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from jpegdec_amd.synth import synth_jpeg  # noqa: E402
from oracle.loader import RefDecoder, digest  # noqa: E402
from tests.cases import OPTIONS, PIXEL_TYPES, PROGRESSIVE_CASES, SYNTH_CASES, progressive_modes  # noqa: E402


def main():
    ref = RefDecoder(simd=False)
    out = {}
    for name, kw in sorted(SYNTH_CASES.items()):
        path = os.path.join(HERE, name + ".jpg")
        if os.path.exists(path):
            jpeg = open(path, "rb").read()
        else:
            jpeg = synth_jpeg(**kw)
            if len(jpeg) <= 160 * 1024:          # keep the repository small: big inputs are regenerated
                with open(path, "wb") as f:
                    f.write(jpeg)
        entry = {"jpeg_sha": digest(jpeg), "jpeg_len": len(jpeg), "info": ref.info(jpeg), "frames": {}}
        for pt in PIXEL_TYPES:
            for opt in OPTIONS:
                if entry["info"]["subsample"] == 0 and pt == 2:
                    continue                      # gray JPEG + RGB8888: reference emits 565 with iBpp=32
                if entry["info"]["subsample"] == 0x12 and pt == 2 and (opt & 4):
                    continue                      # 4:4:0, 1/4 scale, RGB8888: undefined behaviour in the reference (jpeg.inl:4620)
                r = ref.decode_cb(jpeg, pt, opt)
                assert r["rc"] == 1, (name, pt, opt)
                inf, sh, bpp = r["info"], r["scale_shift"], r["bpp"]
                adj = (1 << sh) - 1
                w, h = (inf["width"] + adj) >> sh, (inf["height"] + adj) >> sh
                frame = r["canvas"][:h, : w * bpp]
                entry["frames"]["%d:%d" % (pt, opt)] = {"sha": digest(frame), "w": w, "h": h, "bpp": bpp,
                                                        "draw_calls": r["n_calls"]}
        out[name] = entry
        print(name, len(jpeg), len(entry["frames"]))
    for name, kw in sorted(PROGRESSIVE_CASES.items()):       # first-scan (DC-only) thumbnails
        path = os.path.join(HERE, name + ".jpg")
        if os.path.exists(path):
            jpeg = open(path, "rb").read()
        else:
            jpeg = synth_jpeg(**kw)
            if len(jpeg) <= 160 * 1024:
                with open(path, "wb") as f:
                    f.write(jpeg)
        entry = {"jpeg_sha": digest(jpeg), "jpeg_len": len(jpeg), "info": ref.info(jpeg), "frames": {}}
        for pt, opt in progressive_modes(name):
            r = ref.decode_cb(jpeg, pt, opt)
            assert r["rc"] == 1, (name, pt, opt)
            inf, sh, bpp = r["info"], r["scale_shift"], r["bpp"]
            adj = (1 << sh) - 1
            w, h = (inf["width"] + adj) >> sh, (inf["height"] + adj) >> sh
            frame = r["canvas"][:h, : w * bpp]
            entry["frames"]["%d:%d" % (pt, opt)] = {"sha": digest(frame), "w": w, "h": h, "bpp": bpp, "draw_calls": r["n_calls"]}
        out[name] = entry
        print(name, len(jpeg), len(entry["frames"]))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
