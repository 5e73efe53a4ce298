"""Replaying the walks recorded from the unmodified reference (tests/golden/api_walks.json, script_walks.json; made by
tests/golden/make_*_walk_golden.py where /root/reference exists, every walk in a process of its own because some kill the reference)
through a build of the product's JPEGDEC class: on the GPU box the real one (tests/test_gpu_api_walks.py, tests/test_gpu_script_walks.py),
here the class's host logic over a CPU stand-in for the device entry points (tests/test_class_walks_cpu.py)."""
import hashlib
import json
import os

import numpy as np

from tests.cases import jpeg_for
from tests.ref_fixtures import ref_jpeg

HERE = os.path.dirname(os.path.abspath(__file__))


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:20]

def _documented_refusal(w, info):
    """what the product refuses with JPEG_UNSUPPORTED_FEATURE because the reference's own behaviour is undefined there"""
    opt, scale = w["options"], w["options"] & 14
    progressive = info["jpegtype"] == 1
    if not progressive:
        return bool(scale & (scale - 1))                            # two scale bits on a baseline image
    gray_out = w["pixel_type"] == 3 or (opt & 64)
    return (info["subsample"] != 0 and gray_out) or bool((opt & 4) and not (opt & 2))


def _crop_reaches_below_the_image(w, jpeg):
    import jpegdec_amd as J
    if w["crop"] is None or (w["options"] & 32):
        return False
    p = J.PreparedImage(jpeg)
    try:
        x, y, cw, ch = J.crop_round(p.info, *w["crop"])
        return y + ch > p.info.mcus_y * p.info.mcu_h
    finally:
        p.close()


def _fb_undefined_in_the_reference(w, jpeg):
    """framebuffer mode: DESIGN.md 3's known divergences"""
    import jpegdec_amd as J
    p = J.PreparedImage(jpeg)
    try:
        info = p.info
        if info.subsample == 0x12 and w["pixel_type"] == 2 and (w["options"] & 14) == 4 and not (w["options"] & 64):
            return True                                              # JPEGPutMCU12's pointer bug (see below)
        # a 4:4:4 image whose width is not a multiple of 8, full size, colour output: the reference's clipped last MCU advances
        # pCb / pCr but not pY (jpeg.inl:3521-3557); the product delivers the correct pixels there
        if info.subsample == 0x11 and info.ncomp == 3 and (info.width & 7) and not (w["options"] & 14) and w["pixel_type"] != 3 and not (w["options"] & 64):
            return True
        if w["crop"] is not None and (w["options"] & 14):
            return True
    finally:
        p.close()
    return False


def _pixels_undefined_in_the_reference(w, jpeg):
    """where the reference's strips hold bytes it never wrote (or wrote somewhere else): only the draw sequence is compared"""
    import jpegdec_amd as J
    p = J.PreparedImage(jpeg)
    try:
        info = p.info
        if w["crop"] is not None:
            x, y, cw, ch = J.crop_round(info, *w["crop"])
            if x + cw > info.width or y + ch > info.height:          # an overhanging request: jpeg.inl:716-719 subtracts one MCU, not the overhang --
                return True                                          # the strips are wider than the MCUs the reference decodes into them
            if w["options"] & 14:                                    # a crop with a scale option: jpeg.inl:5135 compares SCALED MCU positions with the
                return True                                          # UNSCALED crop rectangle -- the row ends before the strip is full
        # JPEGPutMCU12, 1/4 scale, RGB8888 (jpeg.inl:4627): "(uint32_t *)&pOutput" -- the pixel is stored over the local POINTER, whose
        # low half then aims the following stores somewhere else; the product delivers the pixels the code meant
        if info.subsample == 0x12 and w["pixel_type"] == 2 and (w["options"] & 14) == 4 and not (w["options"] & 64):
            return True
    finally:
        p.close()
    return False


def check_api_walks(product_class):
    G = json.load(open(os.path.join(HERE, "golden", "api_walks.json")))
    compared = refused = crashed = loose = 0
    wrong = []
    for item in G["walks"]:
        w, ref = item["walk"], item["ref"]
        if "crashed" in ref:
            crashed += 1
            continue
        jpeg = ref_jpeg(w["image"][4:]) if w["image"].startswith("ref:") else jpeg_for(w["image"])
        info = product_class.info(jpeg)
        if w.get("fb"):
            rc, fb = product_class.decode_fb(jpeg, w["pixel_type"], w["options"], crop=w["crop"])
            if rc == 0 and product_class.last_error == 3 and _documented_refusal(w, info):
                refused += 1
                continue
            if _crop_reaches_below_the_image(w, jpeg):
                continue
            got = dict(rc=int(rc), last_error=int(product_class.last_error), fb=_sha(fb[: ref["fb_bytes"]]) if fb is not None else None)
            want = {k: ref[k] for k in got}
            if ref["rc"] != 1:
                got.pop("fb"); want.pop("fb")
            elif _fb_undefined_in_the_reference(w, jpeg):
                got.pop("fb"); want.pop("fb")
                loose += 1
            if got != want:
                wrong.append((w, {k: (got[k], want[k]) for k in got if got[k] != want[k]}))
            compared += 1
            continue
        r = product_class.decode_cb(jpeg, w["pixel_type"], w["options"], max_mcus=w["max_mcus"], xoff=w["xoff"], yoff=w["yoff"],
                                    crop=w["crop"], want_log=True, used_only=True)
        if r["rc"] == 0 and r["last_error"] == 3 and _documented_refusal(w, info):
            refused += 1
            continue
        if _crop_reaches_below_the_image(w, jpeg):                   # the reference decodes the bytes behind the scan as extra rows
            continue
        got = dict(rc=int(r["rc"]), last_error=int(r["last_error"]), n_calls=int(r["n_calls"]), dma_reuse=int(r["dma_reuse"]),
                   log=_sha(r["log"]) if r["log"] is not None else None, canvas=_sha(r["canvas"]) if r["canvas"] is not None else None)
        want = {k: ref[k] for k in got}
        if ref["rc"] != 1:                                           # a failed decode: the verdict (how many strips the reference still delivers
            got = {k: got[k] for k in ("rc", "last_error")}          # depends on what its 2 KB file buffer holds behind the data: DESIGN.md 3)
            want = {k: want[k] for k in got}
        elif _pixels_undefined_in_the_reference(w, jpeg):
            got.pop("canvas"); want.pop("canvas")
            loose += 1
        if got != want:
            wrong.append((w, {k: (got[k], want[k]) for k in got if got[k] != want[k]}))
        compared += 1
    if wrong and os.environ.get("JDA_API_WALK_DUMP"):
        json.dump(wrong, open(os.environ["JDA_API_WALK_DUMP"], "w"))
    assert not wrong, (len(wrong), wrong[:12])
    assert compared >= 600 and refused <= 30 and loose <= 100, (compared, refused, crashed, loose)      # (loose: draw sequence compared, pixels undefined in the reference)


def check_script_walks(product_class):
    G = json.load(open(os.path.join(HERE, "golden", "script_walks.json")))
    compared, wrong = 0, []
    for item in G["scripts"]:
        sc, ref = item["script"], item["ref"]
        if "crashed" in ref:
            continue
        jpeg = ref_jpeg(sc["image"][4:]) if sc["image"].startswith("ref:") else jpeg_for(sc["image"])
        got = product_class.run_script(jpeg, sc["ops"])
        if got != ref["values"]:
            first = next((k for k in range(min(len(got), len(ref["values"]))) if got[k] != ref["values"][k]), min(len(got), len(ref["values"])))
            wrong.append(dict(i=sc["i"], image=sc["image"], ops=sc["ops"], first_difference=first, got=got[max(0, first - 3): first + 6], want=ref["values"][max(0, first - 3): first + 6]))
        compared += 1
    if wrong and os.environ.get("JDA_API_WALK_DUMP"):
        json.dump(wrong, open(os.environ["JDA_API_WALK_DUMP"], "w"))
    assert not wrong, (len(wrong), wrong[:6])
    assert compared >= 240
