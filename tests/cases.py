"""Shared test inputs: small deterministic synthetic JPEGs covering the shapes the path supports."""
import functools
import os

from jpegdec_amd.synth import synth_jpeg

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> kwargs for synth_jpeg
SYNTH_CASES = {
    "c420_333x217": dict(width=333, height=217, subsampling="4:2:0", seed=11),
    "c444_333x217": dict(width=333, height=217, subsampling="4:4:4", seed=12),
    "gray_333x217": dict(width=333, height=217, subsampling="gray", seed=13),
    "c420_640x368_rstrow": dict(width=640, height=368, subsampling="4:2:0", seed=14, restart_rows=1),
    "c420_256x256_q98": dict(width=256, height=256, subsampling="4:2:0", seed=15, quality=98),
    "c444_256x256_q100_opt": dict(width=256, height=256, subsampling="4:4:4", seed=16, quality=100, optimize=True),
    "gray_64x64_rst3": dict(width=64, height=64, subsampling="gray", seed=17, restart_blocks=3),
    "c420_1100x48": dict(width=1100, height=48, subsampling="4:2:0", seed=18),       # > 32 MCUs per row: 3 tiles
    "gray_1100x24": dict(width=1100, height=24, subsampling="gray", seed=19),        # 138 MCUs per row: one partial tile
    "gray_1600x16": dict(width=1600, height=16, subsampling="gray", seed=23),        # a full 192-MCU gray tile + remainder
    "c444_600x16": dict(width=600, height=16, subsampling="4:4:4", seed=24),         # a full 64-MCU 4:4:4 tile + remainder
    "c444_384x192_q100_rst7": dict(width=384, height=192, subsampling="4:4:4", seed=36, quality=100, restart_blocks=7),  # truncated reads + restarts
    "c420_512x256_q98_rstrow": dict(width=512, height=256, subsampling="4:2:0", seed=31, quality=98, restart_rows=1),
    "c420_16x16": dict(width=16, height=16, subsampling="4:2:0", seed=20),           # single MCU
    "c444_8x8_q30": dict(width=8, height=8, subsampling="4:4:4", seed=21, quality=30),
    "c420_1280x720": dict(width=1280, height=720, subsampling="4:2:0", seed=1234),   # BASELINE config 2 shape
    # 4:2:2 (h2v1, Pillow) and 4:4:0 (h1v2, our own encoder: Pillow cannot write it)
    "c422_333x217": dict(width=333, height=217, subsampling="4:2:2", seed=41),
    "c422_1100x24_rstrow": dict(width=1100, height=24, subsampling="4:2:2", seed=42, restart_rows=1),   # 69 MCUs per row: 5 tiles
    "c440_200x120": dict(width=200, height=120, subsampling="4:4:0", seed=43, quality=90),
    "c440_300x64_rst5": dict(width=300, height=64, subsampling="4:4:0", seed=44, restart_blocks=5),
    "c420_250x250_q10": dict(width=250, height=250, subsampling="4:2:0", seed=22, quality=10),  # many DC-only blocks
    # word-precision DQT (jpeg.inl:1742-1750): quantisers x 400 / x 3000 over uniform noise -> max |coef| x max |q'| >= 2^21, the
    # 32-bit-multiply kernels (jda_image_fast_mul == 0: asserted by tests/test_frontend.py::test_word_precision_dqt_cases)
    "w16_gray_200x120_x400": dict(width=200, height=120, subsampling="gray", seed=61, quality=90, noise=True, dqt16=400),
    "w16_c444_136x88_x3000": dict(width=136, height=88, subsampling="4:4:4", seed=62, quality=90, noise=True, dqt16=3000),
    "w16_c420_333x217_x400": dict(width=333, height=217, subsampling="4:2:0", seed=63, quality=90, noise=True, dqt16=400),
    "w16_c422_200x72_x3000": dict(width=200, height=72, subsampling="4:2:2", seed=64, quality=90, noise=True, dqt16=3000),
    "w16_c440_120x96_x400": dict(width=120, height=96, subsampling="4:4:0", seed=65, quality=90, noise=True, dqt16=400),
}
WORD_DQT_CASES = sorted(k for k in SYNTH_CASES if k.startswith("w16_"))

# progressive files (decoded from their first, DC-only scan as a 1/8 thumbnail, jpeg.inl:4964-4966)
PROGRESSIVE_CASES = {
    "p420_200x120": dict(width=200, height=120, subsampling="4:2:0", seed=51, progressive=True),
    "p444_333x217": dict(width=333, height=217, subsampling="4:4:4", seed=52, progressive=True),
    "p422_640x368": dict(width=640, height=368, subsampling="4:2:2", seed=53, progressive=True),
    "pgray_100x100": dict(width=100, height=100, subsampling="gray", seed=54, progressive=True),
    "p420_1280x720_q95": dict(width=1280, height=720, subsampling="4:2:0", seed=55, quality=95, progressive=True),
}
SYNTH_ALL = dict(SYNTH_CASES, **PROGRESSIVE_CASES)

PIXEL_TYPES = (0, 1, 2, 3)                 # RGB565_LE, RGB565_BE, RGB8888, GRAY8
OPTIONS = (0, 2, 4, 8, 64, 64 | 2)         # full, 1/2, 1/4, 1/8, luma-only, luma-only 1/2


@functools.lru_cache(maxsize=None)
def jpeg_for(name: str) -> bytes:
    path = os.path.join(GOLDEN_DIR, name + ".jpg")
    if os.path.exists(path):               # committed fixture wins (keeps tests independent of Pillow's version)
        return open(path, "rb").read()
    return synth_jpeg(**SYNTH_ALL[name])


def all_modes(name):
    for pt in PIXEL_TYPES:
        for opt in OPTIONS:
            if "c440" in name and pt == 2 and (opt & 4):
                continue       # JPEGPutMCU12, 1/4 scale, RGB8888 writes through the address of a local (jpeg.inl:4620): UB in the reference
            yield pt, opt


def progressive_modes(name):
    """pixel type x options a progressive file can be decoded with: everything except what the reference itself cannot do
    -- a colour file to 8-bit gray (it crashes: JPEGDecodeMCU_P(MCU_SKIP) stores far outside the object) and
    JPEG_SCALE_QUARTER (uninitialised sample bytes in its output); see DESIGN.md 3."""
    gray = name.startswith("pgray")
    for pt in PIXEL_TYPES:
        for opt in OPTIONS:
            if (opt & 4) and not (opt & 2):
                continue
            if not gray and (pt == 3 or (opt & 64)):
                continue
            if gray and pt == 2:
                continue       # gray JPEG + RGB8888: the reference emits 565 with iBpp = 32
            yield pt, opt
