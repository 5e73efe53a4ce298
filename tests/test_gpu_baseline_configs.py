"""BASELINE.json configurations at their full image sizes, bit-exact on the GPU.

C2  1280x720 4:2:0 -> RGB8888, a batch of resident images in one launch plan
C3  4096x4096 4:4:4 -> RGB8888
C4  1920x1080 4:2:0 -> RGB8888 (per-GPU shard of the 8-GPU config; height is not an MCU multiple)
C5  8192x8192 gray -> GRAY8 at 1, 1/2, 1/4, 1/8 (DC-only fast path at 1/8)
plus the metric image 4096x4096 4:2:0.  Inputs come from bench.py's cache (same synthetic recipe).
Checker: the oracle restatement (and the real reference where oracle/_ref travelled)."""
import numpy as np
import pytest

import bench
import jpegdec_amd as J

pytestmark = pytest.mark.gpu


def _decode_batch(ctx, jpegs, pts, opts):
    prepared = [J.PreparedImage(j) for j in jpegs]
    dev = [J.DeviceImage(ctx, p) for p in prepared]
    outs, ptrs, geos = [], [], []
    for p, pt, opt in zip(prepared, pts, opts):
        g = p.geometry(pt, opt)
        pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
        ptr = ctx.malloc(pitch * g["canvas_h"])
        outs.append((ptr, pitch, g["canvas_w"], g["canvas_h"]))
        ptrs.append(ptr)
        geos.append((g, pitch))
    b = J.Batch(ctx, dev, outs, pts, opts)
    b.decode()
    ctx.sync()
    res = []
    for ptr, (g, pitch) in zip(ptrs, geos):
        res.append(ctx.to_host(ptr, pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * g["bpp"]].copy())
        ctx.free(ptr)
    b.close()
    for d in dev:
        d.close()
    return res


def _check(got, jpeg, pt, opt, oracle):
    rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
    assert rc == 1
    assert got.shape == want.shape and np.array_equal(got, want)


def test_c2_batch_of_720p(gpu_ctx, oracle):
    jpegs = [bench.cached_jpeg(1280, 720, "4:2:0", 1234 + i) for i in range(4)]
    batch = [jpegs[i % 4] for i in range(32)]                      # one launch plan over 32 resident images
    res = _decode_batch(gpu_ctx, batch, [J.RGB8888] * 32, [0] * 32)
    for i in (0, 1, 2, 3, 31):
        _check(res[i], batch[i], J.RGB8888, 0, oracle)
    assert all(np.array_equal(res[i], res[i % 4]) for i in range(32))   # copies of one image decode identically


def test_c3_4096_444(gpu_ctx, oracle):
    jpeg = bench.cached_jpeg(4096, 4096, "4:4:4", 1234)
    (got,) = _decode_batch(gpu_ctx, [jpeg], [J.RGB8888], [0])
    _check(got, jpeg, J.RGB8888, 0, oracle)


def test_c4_1080p(gpu_ctx, oracle, ref_scalar):
    jpeg = bench.cached_jpeg(1920, 1080, "4:2:0", 1234)
    (got,) = _decode_batch(gpu_ctx, [jpeg], [J.RGB8888], [0])
    _check(got, jpeg, J.RGB8888, 0, oracle)
    r = ref_scalar.decode_cb(jpeg, J.RGB8888, 0)                    # the real reference, valid rows only
    assert np.array_equal(got[:1080], r["canvas"][:1080, : got.shape[1]])


def test_metric_image_4096_420(gpu_ctx, oracle):
    jpeg = bench.cached_jpeg(4096, 4096, "4:2:0", 1235)
    got, got565 = _decode_batch(gpu_ctx, [jpeg, jpeg], [J.RGB8888, J.RGB565_LE], [0, 0])
    _check(got, jpeg, J.RGB8888, 0, oracle)
    _check(got565, jpeg, J.RGB565_LE, 0, oracle)


def test_c5_8192_gray_all_scales(gpu_ctx, oracle):
    jpeg = bench.cached_jpeg(8192, 8192, "gray", 1234)
    opts = [0, J.SCALE_HALF, J.SCALE_QUARTER, J.SCALE_EIGHTH]
    res = _decode_batch(gpu_ctx, [jpeg] * 4, [J.GRAY8] * 4, opts)
    for got, opt in zip(res, opts):
        _check(got, jpeg, J.GRAY8, opt, oracle)
    (got565,) = _decode_batch(gpu_ctx, [jpeg], [J.RGB565_BE], [J.SCALE_EIGHTH])
    _check(got565, jpeg, J.RGB565_BE, J.SCALE_EIGHTH, oracle)


def test_device_prescan_at_full_sizes(gpu_ctx):
    """SURVEY 8f N2 at BASELINE sizes: a mixed batch (720p, 1080p, 4096x4096 4:2:0 and 4:4:4, 8192x8192 gray) uploaded with
    JDA_PREPARE_DEVICE_PRESCAN in ONE jda_upload_batch -- every per-block index made on the GPU must equal the serial host
    pre-scan's (size-independent property: equality of the whole index, DC predictors and MCU count)."""
    jpegs = [bench.cached_jpeg(1280, 720, "4:2:0", 1234), bench.cached_jpeg(1920, 1080, "4:2:0", 1235),
             bench.cached_jpeg(4096, 4096, "4:2:0", 1234), bench.cached_jpeg(4096, 4096, "4:4:4", 1234),
             bench.cached_jpeg(8192, 8192, "gray", 1234), bench.cached_jpeg(1280, 720, "4:2:0", 1236)]
    preps = [J.PreparedImage(j, device_prescan=True) for j in jpegs]
    assert all(p.prescan_pending for p in preps)
    dimgs = J.upload_batch(gpu_ctx, preps)
    rounds = gpu_ctx.lib.jda_last_prescan_rounds(gpu_ctx.handle)
    assert 1 <= rounds <= 24, rounds
    for j, d in zip(jpegs, dimgs):
        assert d.prescan_on_device
        host = J.PreparedImage(j)
        want_idx, nok = host.block_index()
        got_idx, got_dc = d.read_index()
        assert d.n_mcus_ok == nok
        assert J.index_equivalent(got_idx, want_idx) and np.array_equal(got_dc, host.block_dc())
        d.close(); host.close()
