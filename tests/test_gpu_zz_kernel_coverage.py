"""Every kernel of the product's code object is launched -- and checked against the oracle -- by a deterministic test.

The library counts its launches per kernel function (jda_kernel_launch_counts, C-ABI); the list of kernels comes from the code
object itself (the gfx950 ELF inside libjpegdec_amd.so: every *.kd symbol).  This file runs one decode per routing class -- layout x
(plain case | general | 32-bit multiplies | large window | P1 in chunks | 1/4 | 1/8) plus the streamed pipeline's filter and pre-scan
kernels with and without restart markers --, compares each with the oracle, and then holds the PROCESS's counters (this file's
launches and whatever the suite ran before it) to the whole list: a kernel nothing can reach has no business in the library.
The report goes to gpurun_out/kernel_coverage.txt (copied to profiles/ by hand)."""
import os
import re
import subprocess

import numpy as np
import pytest

import jpegdec_amd as J
from tests.cases import jpeg_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object_kernels(tmp_path):
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "k.co")
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", J.library_path(), fat], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    syms = subprocess.run([LLVM + "/llvm-readelf", "-sW", co], check=True, capture_output=True, text=True).stdout
    return sorted(set(m.group(1) for m in re.finditer(r"\s(\S+)\.kd\s*$", syms, re.M)))


def _same(ctx, oracle, jpeg, pt, opt, prepared=None, what=""):
    orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
    assert orc == 1, (what, err)
    if prepared is None:
        rc, got, g = J.decode_to_host(ctx, jpeg, pt, opt)
    else:
        rc, got, g = J.decode_resident(ctx, prepared, pt, opt)
    assert rc == 0, (what, rc)
    assert np.array_equal(got, want), (what, pt, opt, int(np.count_nonzero(got != want)))


def test_every_kernel_of_the_code_object_is_launched(gpu_ctx, oracle, tmp_path):
    from jpegdec_amd.synth import synth_jpeg
    kernels = code_object_kernels(tmp_path)
    assert len(kernels) >= 30
    layouts = {"gray": "gray_333x217", "444": "c444_333x217", "420": "c420_333x217", "422": "c422_333x217", "440": "c440_200x120"}
    w16 = {"gray": "w16_gray_200x120_x400", "444": "w16_c444_136x88_x3000", "420": "w16_c420_333x217_x400", "422": "w16_c422_200x72_x3000",
           "440": "w16_c440_120x96_x400"}
    for lay, name in layouts.items():
        jpeg = jpeg_for(name)
        gray = lay == "gray"
        # plain cases (RGB8888 / RGB565 LE / GRAY8 at full size), the general kernel (big-endian 565, half size), 1/4, 1/8
        for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, 0), (J.GRAY8, 0), (J.RGB565_BE, 0), (J.RGB565_LE, J.SCALE_HALF), (J.RGB565_LE, J.SCALE_QUARTER),
                        (J.RGB565_BE, J.SCALE_EIGHTH), (J.GRAY8, J.SCALE_QUARTER), (J.GRAY8, J.SCALE_EIGHTH)):
            if gray and pt == J.RGB8888:
                continue
            _same(gpu_ctx, oracle, jpeg, pt, opt, what=name)
        # 32-bit multiplies: word-precision quantisers (full size through the general kernel; the scaled kernels on the same files)
        jw = jpeg_for(w16[lay])
        for pt, opt in ((J.RGB565_LE if gray else J.RGB8888, 0), (J.GRAY8, 0), (J.RGB565_BE, J.SCALE_HALF), (J.RGB565_LE, J.SCALE_QUARTER), (J.GRAY8, J.SCALE_EIGHTH)):
            _same(gpu_ctx, oracle, jw, pt, opt, what=w16[lay])
    # a whole gray or 4:2:0 image at 1/8 is the flat kernels' (above); a cropped one keeps the tile kernel: the rectangle's pixels are the oracle's
    jg = jpeg_for("gray_333x217")
    for pt, bpp in ((J.GRAY8, 1), (J.RGB565_BE, 2)):
        orc, want, err = oracle.decode_canvas(jg, pt, J.SCALE_EIGHTH)
        rc, part, g, tiles = J.binding.decode_to_host_rect(gpu_ctx, jg, pt, J.SCALE_EIGHTH, (3, 2, 30, 20))
        assert orc == 1 and rc == 0
        assert np.array_equal(part[2:20, 3 * bpp:30 * bpp], want[2:20, 3 * bpp:30 * bpp]), pt
    jc = jpeg_for("c420_333x217")                                # (.. and of a 4:2:0 image: 2 x 2 pixels an MCU)
    orc, want, err = oracle.decode_canvas(jc, J.RGB8888, J.SCALE_EIGHTH)
    rc, part, g, tiles = J.binding.decode_to_host_rect(gpu_ctx, jc, J.RGB8888, J.SCALE_EIGHTH, (2, 1, 15, 9))
    assert orc == 1 and rc == 0 and np.array_equal(part[2:18, 4 * 4:30 * 4], want[2:18, 4 * 4:30 * 4])
    # the large window (one wavefront less per workgroup): uniform noise at a quality whose tiles' scan slices pass the small window and
    # fit the large one (5.0 / 10.3 / 3.5 bits per pixel), and at quality 95, where they pass both (the bit reader's fall-back to HBM)
    for lay, sub, q in (("420", "4:2:0", 75), ("444", "4:4:4", 75), ("gray", "gray", 50)):
        for quality in (q, 95):
            jb = synth_jpeg(320, 64, sub, seed=71, quality=quality, noise=True)
            for pt in ((J.RGB565_LE, J.GRAY8) if lay == "gray" else (J.RGB8888, J.RGB565_BE)):
                _same(gpu_ctx, oracle, jb, pt, 0, what="noise q%d %s" % (quality, lay))
    # P1 in chunks: the RGB8888 kernels of 4:2:0 and 4:4:4 over an image that was given continuation entries
    for name in ("c420_256x256_q98", "c444_256x256_q100_opt"):
        jpeg = jpeg_for(name)
        p = J.PreparedImage(jpeg, flags=J.PREPARE_CONT_ALWAYS)
        assert len(p.block_cont()[1]) > 0
        _same(gpu_ctx, oracle, jpeg, J.RGB8888, 0, prepared=p, what=name + " in chunks")
        p.close()
    # the streamed pipeline: filter, walk tables, both pre-scan walks (with and without restart markers), tail rounds, sums, finalize,
    # candidates, tile lists; and the surface checksum
    files = [jpeg_for(n) for n in ("c420_1280x720", "c420_640x368_rstrow", "c444_384x192_q100_rst7", "c444_256x256_q100_opt", "gray_333x217", "c420_512x256_q98_rstrow")]
    geos = [J.PreparedImage(f).geometry(J.RGB565_LE, 0) for f in files]
    pit = [(g["canvas_w"] * 2 + 15) & ~15 for g in geos]
    offs, total = [], 0
    for g, p in zip(geos, pit):
        offs.append(total)
        total += (p * g["canvas_h"] + 255) & ~255
    base = gpu_ctx.malloc(total)
    pipe = J.Pipeline(gpu_ctx, max_images=len(files), depth=2)
    outs = [(base + offs[i], pit[i], geos[i]["canvas_w"], geos[i]["canvas_h"]) for i in range(len(files))]
    st = pipe.wait(pipe.submit(files, outs, [J.RGB565_LE] * len(files), [0] * len(files)))
    assert list(st) == [0] * len(files), st
    assert pipe.stats["device_images"] == len(files), pipe.stats
    for i, f in enumerate(files):
        orc, want, err = oracle.decode_canvas(f, J.RGB565_LE, 0)
        got = gpu_ctx.to_host(base + offs[i], pit[i] * geos[i]["canvas_h"]).reshape(geos[i]["canvas_h"], pit[i])[:, : geos[i]["canvas_w"] * 2]
        assert np.array_equal(got, want), ("pipeline", i)
    sums = gpu_ctx.checksums(outs[:1], [geos[0]["canvas_w"] * 2])
    assert sums[0] == J.surface_checksum_host(oracle.decode_canvas(files[0], J.RGB565_LE, 0)[1])
    pipe.close()
    gpu_ctx.free(base)

    counts = J.kernel_launch_counts()
    missing = [k for k in kernels if counts.get(k, 0) == 0]
    unknown = [k for k in counts if k not in kernels]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kernel_coverage.txt"), "w") as f:
        f.write("kernels of the gfx950 code object in libjpegdec_amd.so and the launches this pytest process made of each\n"
                "(tests/test_gpu_zz_kernel_coverage.py; every launch above is of a decode compared with the oracle)\n")
        names = subprocess.run(["c++filt"] + kernels, capture_output=True, text=True).stdout.splitlines() if kernels else []
        for k, n in zip(kernels, names):
            f.write("%8d  %s\n" % (counts.get(k, 0), re.sub(r"\(.*$", "", n).replace("void ", "")))
        f.write("%d of %d kernels launched\n" % (len(kernels) - len(missing), len(kernels)))
    assert not unknown, unknown
    assert not missing, "kernels no test reaches: %s" % missing


def test_launch_count_report_truncates_at_whole_lines(gpu_ctx):
    """jda_kernel_launch_counts into a buffer that is too small: whole lines only, none behind the first that did not fit, and the string
    ends where its content ends (the bytes behind it are the caller's, untouched); the return value is what the whole report needs."""
    import ctypes as C
    import jpegdec_amd as J
    from jpegdec_amd.synth import synth_jpeg
    rc, _, _ = J.decode_to_host(gpu_ctx, synth_jpeg(64, 64, "4:2:0", seed=1), J.RGB8888, 0)      # (at least one kernel has been launched)
    assert rc == 0
    lib = J.load_library()
    need = lib.jda_kernel_launch_counts(None, 0)
    full = C.create_string_buffer(need)
    assert lib.jda_kernel_launch_counts(full, need) == need
    lines = full.value.decode().splitlines(keepends=True)
    assert lines and len(full.value) == need - 1 and all(ln.endswith("\n") and ln.rpartition(" ")[2].strip().isdigit() for ln in lines)
    for cap in (1, 2, len(lines[0]), len(lines[0]) + 1, need // 2, need - 1):
        buf = C.create_string_buffer(b"\x55" * (cap + 8), cap + 8)
        assert lib.jda_kernel_launch_counts(buf, cap) == need
        got = buf.raw[:cap].split(b"\0", 1)[0].decode()
        assert b"\0" in buf.raw[:cap] and buf.raw[cap:] == b"\x55" * 8, cap
        assert full.value.decode().startswith(got) and (got == "" or got.endswith("\n")), (cap, got[-80:])
        assert buf.raw[len(got) + 1:cap] == b"\x55" * (cap - len(got) - 1), cap      # nothing written behind the terminator
