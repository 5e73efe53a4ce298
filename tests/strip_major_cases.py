"""Large images through the class with draw callbacks: the decoded image leaves the device STRIP-MAJOR (jda_decode_to_host_strips) and
the callback gets pointers into it.  Shared by the CPU run (class over the stand-in device) and the GPU run."""
import numpy as np

from tests.cases import jpeg_for

RGB565_LE, RGB565_BE, RGB8888, GRAY8 = 0, 1, 2, 3
SCALE_HALF, SCALE_QUARTER, SCALE_EIGHTH, USES_DMA = 2, 4, 8, 128


def check_strip_major_decodes(product, ref):
    """`product` and `ref`: two RefDecoder-shaped drivers (oracle/loader.py).  Every strip of every decode -- position, size, pixels -- and
    the assembled canvas must be the reference's: canvases of 2 MB and more (the strip-major path), all layouts, RGB8888 / RGB565 /
    8-bit gray strips (4 / 8 / 16 MCUs wide), JPEG_USES_DMA, a user cap on the strip width, a decode offset, half size, and a
    damaged stream (the strips before the bad MCU, then JPEG_DECODE_ERROR)."""
    from jpegdec_amd.synth import synth_jpeg
    big = {"c420": synth_jpeg(1296, 1000, "4:2:0", seed=5, quality=80), "c444": synth_jpeg(1000, 808, "4:4:4", seed=6, quality=80),
           "c422": synth_jpeg(1296, 808, "4:2:2", seed=7, quality=80), "gray": synth_jpeg(2100, 1100, "gray", seed=8, quality=80)}
    bad = bytearray(big["c420"])
    at = len(bad) * 2 // 3
    bad[at:at + 48] = bytes((i * 37 + 11) & 0xFF for i in range(48))
    runs = [("c420", RGB8888, 0, {}), ("c420", RGB565_LE, 0, {}), ("c420", RGB565_BE, USES_DMA, {}), ("c420", GRAY8, 0, {}),
            ("c420", RGB8888, 0, {"max_mcus": 3}), ("c420", RGB8888, 0, {"xoff": 5, "yoff": 3}), ("c420", RGB8888, SCALE_HALF, {}),
            ("c444", RGB8888, 0, {}), ("c444", RGB565_LE, 0, {}), ("c422", RGB8888, 0, {}), ("gray", GRAY8, 0, {}), ("gray", RGB565_LE, 0, {}),
            # scaled outputs that are still 2 MB and more: 1/4 and 1/8 have kernels of their own for row-major surfaces, a strip-major one
            # stays with the decode kernel
            ("wide", RGB8888, SCALE_QUARTER, {}), ("wide", RGB565_LE, SCALE_QUARTER, {}), ("huge", RGB8888, SCALE_EIGHTH, {})]
    big["wide"] = synth_jpeg(4096, 2048, "4:2:0", seed=9, quality=80)
    big["huge"] = synth_jpeg(8192, 4096, "4:2:0", seed=10, quality=60)
    n = 0
    for name, pt, opt, kw in runs:
        a = product.decode_cb(big[name], pt, opt, want_log=True, **kw)
        b = ref.decode_cb(big[name], pt, opt, want_log=True, **kw)
        assert a["rc"] == b["rc"] == 1, (name, pt, opt, kw, a["rc"], a["last_error"])
        assert a["n_calls"] == b["n_calls"] and np.array_equal(a["log"], b["log"]), (name, pt, opt, kw)
        assert np.array_equal(a["canvas"], b["canvas"]), (name, pt, opt, kw, int(np.count_nonzero(a["canvas"] != b["canvas"])))
        n += a["n_calls"]
    # a damaged stream: the reference walks on through whatever the bits decode to (here: to the image's end, one MCU short of bits); the
    # product stops where the stream runs out (JPEG_DECODE_ERROR) -- the strips in front of that are the same, in the same order
    a = product.decode_cb(bytes(bad), RGB8888, 0, want_log=True)
    b = ref.decode_cb(bytes(bad), RGB8888, 0, want_log=True)
    assert (a["rc"], a["last_error"]) in ((b["rc"], b["last_error"]), (0, 2))
    k = min(a["n_calls"], b["n_calls"])
    assert k > 1000 and np.array_equal(a["log"][:k], b["log"][:k])
    rows = int(a["log"][k - 1][1])                    # the strips' rows above the last delivered one are complete in both
    assert rows > 0 and np.array_equal(a["canvas"][:rows], b["canvas"][:rows])
    assert n > 5000
    return n
