#!/usr/bin/env python3
"""Random SEQUENCES of calls on one JPEGDEC object (setPixelType / setMaxOutputSize / setCropArea / decode / getters / close + reopen),
run through the unmodified reference (oracle/_ref, scalar build) in a process of its own each, every observable value recorded
(ref_run_script in oracle/ref_shim.cpp).  What state does one decode leave behind for the next?

    python tests/golden/make_script_walk_golden.py     (needs /root/reference; writes tests/golden/script_walks.json)

Kept out of the scripts: what DESIGN.md 3 lists as undefined in the reference (a crop together with a scale option or reaching over the
image, a gray JPEG to RGB8888, two scale bits on a baseline image, 4:4:0 -> RGB8888 at 1/4, a colour progressive file to gray)."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
IMAGES = ["c420_333x217", "c444_333x217", "gray_333x217", "c422_333x217", "c440_200x120", "c420_640x368_rstrow", "c420_250x250_q10",
          "c420_16x16", "p420_200x120", "pgray_100x100", "ref:tulips", "ref:sciopero", "ref:croptest", "ref:thumb_test", "ref:corrupt5", "ref:corrupt3"]
N_SCRIPTS = 260
OUT = os.path.join(ROOT, "tests", "golden", "script_walks.json")


def jpeg_of(name):
    from tests.cases import jpeg_for
    from tests.ref_fixtures import ref_jpeg
    return ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name)


def make_scripts():
    import jpegdec_amd as J
    rng = np.random.default_rng(77)
    scripts = []
    for i in range(N_SCRIPTS):
        name = IMAGES[int(rng.integers(0, len(IMAGES)))]
        info = J.parse(jpeg_of(name))
        gray, prog = info["ncomp"] == 1, info["jpeg_type"] == 1
        ops = []
        for _open in range(int(rng.integers(1, 4))):                 # open -> setters -> ONE decode -> getters, then close + open again
            if _open:
                ops.append([6, 0, 0, 0, 0])
            cropped, pt = False, 0                                   # (what open resets is recorded by the getters / the decode, not assumed)
            for _ in range(int(rng.integers(0, 5))):
                k = rng.random()
                if k < 0.4:
                    pt = int(rng.integers(0, 4))
                    if gray and pt == 2:
                        pt = 0
                    if prog and not gray and pt == 3:
                        pt = 1
                    ops.append([1, pt, 0, 0, 0])
                elif k < 0.6:
                    ops.append([2, int(rng.integers(1, 40)), 0, 0, 0])
                elif k < 0.85 and not prog and info["width"] > 96 and info["height"] > 96:
                    x, y = int(rng.integers(0, info["width"] - 80)), int(rng.integers(0, info["height"] - 80))
                    ops.append([3, x, y, int(rng.integers(8, info["width"] - x - 40)), int(rng.integers(8, info["height"] - y - 40))])
                    cropped = True
                else:
                    ops.append([5, 0, 0, 0, 0])
            opt = 0
            if not cropped:
                opt = int((0, 0, 2, 4, 8)[int(rng.integers(0, 5))])
                if prog and opt == 4:
                    opt = 2
                if info["subsample"] == 0x12 and pt == 2 and opt == 4:
                    opt = 0
            if rng.random() < 0.15 and not (prog and not gray):
                opt |= 64
            if rng.random() < 0.15:
                opt |= 128
            if name == "ref:thumb_test" and rng.random() < 0.6 and not cropped:
                opt |= 32
                if opt & 2:
                    opt &= ~2                                        # (the thumbnail is a baseline image: no second scale bit, see the module text)
            x, y = (int(rng.integers(0, 50)), int(rng.integers(0, 30))) if rng.random() < 0.25 else (0, 0)
            ops.append([4, x, y, opt, 0])
            ops.append([5, 0, 0, 0, 0])
        ops.append([5, 0, 0, 0, 0])
        scripts.append(dict(i=i, image=name, ops=ops))
    return scripts


def run_one(sc, q):
    from oracle.loader import RefDecoder
    q.put(RefDecoder(False).run_script(jpeg_of(sc["image"]), sc["ops"]))


def main():
    ctx = mp.get_context("spawn")
    out = []
    for sc in make_scripts():
        q = ctx.Queue()
        p = ctx.Process(target=run_one, args=(sc, q))
        p.start(); p.join(120)
        if p.is_alive():
            p.kill(); res = dict(crashed="timeout")
        elif p.exitcode != 0:
            res = dict(crashed="exit %d" % p.exitcode)
        else:
            res = dict(values=q.get())
        out.append(dict(script=sc, ref=res))
        print(sc["i"], sc["image"], len(sc["ops"]), res.get("crashed") or len(res["values"]), flush=True)
    json.dump(dict(generator="tests/golden/make_script_walk_golden.py", reference="oracle/_ref scalar build (-DNO_SIMD) of /root/reference", scripts=out),
              open(OUT, "w"), indent=0)


if __name__ == "__main__":
    main()
