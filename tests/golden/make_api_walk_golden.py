#!/usr/bin/env python3
"""Random walks over the reference's public API (pixel type x option bits x crop x decode offsets x max output size), each run through
the UNMODIFIED reference (oracle/_ref, scalar build) in a process of its own -- some combinations crash it -- and recorded:
return code, last error, number of draw calls, hashes of the JPEGDRAW sequence and of the assembled canvas.

    python tests/golden/make_api_walk_golden.py        (needs /root/reference; writes tests/golden/api_walks.json)

tests/test_gpu_api_walks.py replays the walks through the product's JPEGDEC class on the GPU box (where the reference does not exist).
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

IMAGES = ["c420_333x217", "c444_333x217", "gray_333x217", "c422_333x217", "c440_200x120", "c420_640x368_rstrow", "c444_384x192_q100_rst7",
          "c420_250x250_q10", "c420_16x16", "gray_64x64_rst3", "p420_200x120", "p444_333x217", "pgray_100x100", "c420_1100x48"]
REF_IMAGES = ["tulips", "zebra", "sciopero", "croptest", "thumb_test", "corrupt2", "corrupt3", "corrupt5", "demo"]      # the reference's own fixtures (tests/golden/ref/)
N_WALKS = 420
N_REF_WALKS = 160
N_FB_WALKS = 160                      # setFramebuffer() instead of a draw callback
SCALES = (0, 0, 0, 2, 4, 8)            # full size as often as the three reduced ones together
OUT = os.path.join(ROOT, "tests", "golden", "api_walks.json")


def make_walks():
    rng = np.random.default_rng(20260925)
    from tests.cases import jpeg_for
    import jpegdec_amd as J

    walks = []
    for i in range(N_WALKS + N_REF_WALKS + N_FB_WALKS):
        fb = i >= N_WALKS + N_REF_WALKS
        name = IMAGES[int(rng.integers(0, len(IMAGES)))] if (i < N_WALKS or (fb and rng.random() < 0.7)) else "ref:" + REF_IMAGES[int(rng.integers(0, len(REF_IMAGES)))]
        info = J.parse(jpeg_of(name))
        opt = int(SCALES[int(rng.integers(0, len(SCALES)))])
        if rng.random() < 0.06:
            opt |= int(SCALES[int(rng.integers(3, len(SCALES)))])          # now and then a second scale bit
        if rng.random() < 0.2:
            opt |= 64                                                        # JPEG_LUMA_ONLY
        if rng.random() < 0.2:
            opt |= 128                                                       # JPEG_USES_DMA
        if name == "ref:thumb_test" and rng.random() < 0.6:
            opt |= 32                                                        # JPEG_EXIF_THUMBNAIL
        w = dict(i=i, image=name, pixel_type=int(rng.integers(0, 4)), options=opt, max_mcus=0, xoff=0, yoff=0, crop=None)
        if info["ncomp"] == 1 and w["pixel_type"] == 2:
            w["pixel_type"] = 0        # (a gray JPEG to RGB8888: the reference reports 32 bpp and writes 16-bit pixels, SURVEY C.5 -- the other half of each strip is whatever the buffer held)
        if rng.random() < 0.3:
            w["max_mcus"] = int(rng.integers(1, 40))
        if rng.random() < 0.3:
            w["xoff"], w["yoff"] = int(rng.integers(0, 70)), int(rng.integers(0, 40))
        if rng.random() < 0.35 and info["width"] > 48 and info["height"] > 48:
            cx, cy = int(rng.integers(0, info["width"] - 32)), int(rng.integers(0, info["height"] - 32))
            w["crop"] = [cx, cy, int(rng.integers(8, info["width"] - cx + 16)), int(rng.integers(8, info["height"] - cy + 16))]
        if fb:
            w["fb"] = True
            w["max_mcus"] = 0; w["xoff"] = w["yoff"] = 0; w["options"] &= ~(128 | 32)
            if w["crop"] is not None:                                        # (inside the image: an overhanging request is undefined, see the test)
                cx, cy = w["crop"][0], w["crop"][1]
                w["crop"][2] = min(w["crop"][2], info["width"] - cx); w["crop"][3] = min(w["crop"][3], info["height"] - cy)
        walks.append(w)
    return walks


def jpeg_of(name):
    from tests.cases import jpeg_for
    from tests.ref_fixtures import ref_jpeg
    return ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name)


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:20]


def run_one(w, q):
    from oracle.loader import RefDecoder
    from tests.cases import jpeg_for

    ref = RefDecoder(False)
    if w.get("fb"):
        import jpegdec_amd as J
        from tests.fb_region import fb_defined_bytes
        jpeg = jpeg_of(w["image"])
        rc, fb = ref.decode_fb(jpeg, w["pixel_type"], w["options"], crop=w["crop"])
        p = J.PreparedImage(jpeg)
        n = fb_defined_bytes(J, p.info, w["pixel_type"], w["options"], w["crop"])      # (behind them the reference overruns the image)
        q.put(dict(rc=int(rc), last_error=int(ref.last_error), fb=sha(fb[:n]) if fb is not None else None, fb_bytes=int(n)))
        return
    r = ref.decode_cb(jpeg_of(w["image"]), w["pixel_type"], w["options"], max_mcus=w["max_mcus"], xoff=w["xoff"], yoff=w["yoff"],
                      crop=w["crop"], want_log=True, used_only=True)     # (a strip can be wider than what the reference writes into it -- crops, decode offsets at reduced scale: only iWidthUsed pixels are defined)
    q.put(dict(rc=int(r["rc"]), last_error=int(r["last_error"]), n_calls=int(r["n_calls"]), dma_reuse=int(r["dma_reuse"]),
               log=sha(r["log"]) if r["log"] is not None else None, canvas=sha(r["canvas"]) if r["canvas"] is not None else None,
               canvas_shape=list(r["canvas"].shape) if r["canvas"] is not None else None))


def main():
    ctx = mp.get_context("spawn")
    out = []
    for w in make_walks():
        q = ctx.Queue()
        p = ctx.Process(target=run_one, args=(w, q))
        p.start()
        p.join(120)
        res = None
        if p.is_alive():
            p.kill()
            res = dict(crashed="timeout")
        elif p.exitcode != 0:
            res = dict(crashed="exit %d" % p.exitcode)                      # the reference died (segmentation fault, SIGFPE)
        else:
            res = q.get()
        out.append(dict(walk=w, ref=res))
        print(w["i"], w["image"], w["pixel_type"], w["options"], res.get("crashed") or (res["rc"], res["last_error"], res.get("n_calls")), flush=True)
    json.dump(dict(generator="tests/golden/make_api_walk_golden.py", reference="oracle/_ref scalar build (-DNO_SIMD) of /root/reference", walks=out),
              open(OUT, "w"), indent=0)


if __name__ == "__main__":
    main()
