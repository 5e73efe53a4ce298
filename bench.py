#!/usr/bin/env python3
"""bench.py -- Mpixels/s decoded on the BASELINE.json metric workload.

A "step" is one pass of the hot path (the MCU loops of DecodeJPEG, reference jpeg.inl:5109-5353,
here one kernel launch) over one batch of synthetic baseline JPEGs whose inputs (filtered scan,
per-block index, tables) are already resident in HBM.  Default workload = the metric's own
configuration: 4096x4096 baseline 4:2:0 -> RGB8888.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  ONE image list (N x batch images; --workload c4: BASELINE config 4's 8192 x 1920x1080) is
sharded into contiguous blocks over the ranks (jpegdec_amd/sharding.py); a rank prepares its shard on its share of
the host cores (pinned to its GPU's NUMA node), keeps it resident and decodes it -- no data-path collective, the
pixels stay where they were decoded.  RCCL carries the barrier that brackets the timed region, the max-over-ranks
of the elapsed time, the work counters, and the all-reduce of the per-image checksum / decode-count vectors that
proves every image of the list was decoded exactly once and equal to its single-GPU decode.  Rank 0 prints ONE
JSON line.  Besides the kernel-only `value` it carries `end_to_end` -- the same images streamed from host memory
through jda_pipeline (device filter + pre-scan + decode, batches overlapped), host work included.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cached_jpeg(width, height, subsampling, seed, quality=85, restart_rows=0):
    """Synthetic input (jpegdec_amd/synth.py recipe), cached under bench_cache/ so the GPU box does
    not spend its minutes on Pillow."""
    from jpegdec_amd.synth import synth_jpeg

    d = os.path.join(ROOT, "bench_cache")
    os.makedirs(d, exist_ok=True)
    name = "synth_%dx%d_%s_q%d_s%d%s.jpg" % (width, height, subsampling.replace(":", ""), quality, seed,
                                             "_rst%d" % restart_rows if restart_rows else "")
    for dd in (d, os.path.join(ROOT, "bench_cold")):
        if os.path.exists(os.path.join(dd, name)):
            return open(os.path.join(dd, name), "rb").read()
    path = os.path.join(d, name)
    data = synth_jpeg(width, height, subsampling, seed=seed, quality=quality, restart_rows=restart_rows)
    with open(path + ".tmp%d" % os.getpid(), "wb") as f:
        f.write(data)
    os.replace(path + ".tmp%d" % os.getpid(), path)
    return data


def cpu_baseline(jpegs, pixel_type, cores, detail, model, wall_target=5.0):
    """The reference itself (oracle/_ref, built from /root/reference) timed on this host's cores: its default x86-64
    build (SSE2) on one thread and on all usable cores, and its scalar integer build (-DNO_SIMD, the parity target)
    on one thread -- linux/examples/jpeg_perf_test's convention (one JPEGDEC object per thread, no-op draw callback).
    Test-infrastructure use of oracle/: a reported baseline, never the measured product."""
    from oracle.loader import RefDecoder, ref_available

    if not ref_available(simd=True):
        return None
    ref = RefDecoder(simd=True)

    def timed(dec, threads, wall):
        r0 = dec.bench(jpegs[:1], pixel_type, 0, 1, 1)              # calibrate: one decode, one thread
        per_image = max(r0["seconds"], 1e-4)
        per_thread = max(8, int(round(wall / per_image)))           # >= 8 decodes per thread, >= `wall` seconds
        imgs = [jpegs[i % len(jpegs)] for i in range(threads)]      # one image per thread and repetition
        r = dec.bench(imgs, pixel_type, 0, per_thread, threads)
        return {"mpix_s": r["pixels"] / r["seconds"] / 1e6, "threads": threads, "decodes_per_thread": per_thread,
                "wall_s": round(r["seconds"], 2), "failures": r["failures"]}

    one = timed(ref, 1, 3.0)
    allc = timed(ref, cores, wall_target)
    scalar = timed(RefDecoder(simd=False), 1, 3.0) if ref_available(simd=False) else None
    out = {
        "value": allc["mpix_s"],
        "unit": "Mpixels/s",
        "cores": cores,
        "kind": "reference",
        "sample": "%d threads x %d decodes of the workload images (%d distinct), JPEGDEC default x86-64 build (SSE2), one JPEGDEC "
                  "object per thread, no-op draw callback, %.1f s wall" % (cores, allc["decodes_per_thread"], len(jpegs), allc["wall_s"]),
        "cpu_model": model,
        "cores_detail": detail,
        "sse2_1_thread_mpix_s": one["mpix_s"],
        "sse2_all_cores_mpix_s": allc["mpix_s"],
        "scaling_all_over_1": allc["mpix_s"] / one["mpix_s"],
        "scalar_no_simd_1_thread_mpix_s": scalar["mpix_s"] if scalar else None,
        "runs": {"sse2_1": one, "sse2_all": allc, "scalar_1": scalar},
    }
    return out


def check_against_reference(J, ctx, jpeg, pt, options, base, img_bytes, pitch, geo, device_sum=None):
    """One decoded surface against the checker -- oracle/_ref (the unmodified reference, scalar build) where it travelled, else the
    oracle restatement: test-infrastructure use of oracle/, outside every timed region."""
    try:
        from oracle.loader import OracleDecoder, RefDecoder, ref_available
        got = ctx.to_host(base, img_bytes).reshape(geo["canvas_h"], pitch)[:, : geo["canvas_w"] * geo["bpp"]]
        if ref_available(False):
            want = RefDecoder(False).decode_cb(jpeg, pt, options)["canvas"][: geo["canvas_h"], : geo["canvas_w"] * geo["bpp"]]
            res = {"checker": "oracle/_ref scalar reference", "bit_exact": bool(np.array_equal(got[: geo["out_h"]], want[: geo["out_h"]]))}
        else:
            rc, want, _ = OracleDecoder().decode_canvas(jpeg, pt, options)
            res = {"checker": "oracle restatement", "bit_exact": bool(rc == 1 and np.array_equal(got, want))}
        if device_sum is not None:
            res["device_checksum_equals_host_checksum"] = bool(J.surface_checksum_host(got) == device_sum)
        return res
    except Exception as e:  # checker missing: report, do not fail the measurement
        return {"checker": "unavailable: %s" % e, "bit_exact": None}


def config_leg(J, ctx, name, what, jpegs, n_images, pt, options, threads, steps=20, ramp_ms=150.0):
    """A short kernel-only leg over one of the other BASELINE.json configurations (inputs resident, index made by the device
    pre-scan at upload): ms per launch from HIP events, fraction of the HBM roofline on the algorithmic bytes, image 0 against
    the reference.  Reported under `configs` in the line; the headline stays the metric workload."""
    t0 = time.perf_counter()
    files = [jpegs[i % len(jpegs)] for i in range(n_images)]
    prepared = J.prepare_batch(files, device_prescan=True, threads=threads)
    geo = prepared[0].geometry(pt, options)
    pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15
    img_bytes = pitch * geo["canvas_h"]
    base = ctx.malloc(img_bytes * n_images)
    dev = J.upload_batch(ctx, prepared)
    outs = [(base + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for i in range(n_images)]
    batch = J.Batch(ctx, dev, outs, [pt] * n_images, [options] * n_images)
    st = batch.stats
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < ramp_ms:
        for _ in range(4):
            batch.decode()
        ctx.sync()
    ctx.timer_start()
    for _ in range(steps):
        batch.decode()
    ctx.timer_stop()
    ctx.sync()
    ms = ctx.timer_elapsed_ms() / steps
    # algorithmic bytes: output + filtered scan + 4 B/MCU of index; at 1/8 the pixels are the blocks' DC values (index
    # format 2 carries them): output + 2 B per block, no scan, no index entry
    if options & J.SCALE_EIGHTH and not options & (J.SCALE_HALF | J.SCALE_QUARTER):
        algo = st["output_bytes"] + 2 * sum(p.n_blocks for p in prepared)
    else:
        algo = st["output_bytes"] + st["scan_bytes"] + 4 * sum(p.n_mcus for p in prepared)
    # every surface's checksum, made where the pixels are: all decodes of one file must be one value, image 0's is checked on the host
    sums = ctx.checksums(outs, [geo["canvas_w"] * geo["bpp"]] * n_images)
    nd = len(jpegs)
    all_equal = all(sums[i] == sums[i % nd] for i in range(n_images))
    res = {"workload": what, "images": n_images, "distinct_images": len(jpegs), "options": options,
           "bits_per_pixel": round(8.0 * sum(len(j) for j in jpegs) / (len(jpegs) * prepared[0].info.width * prepared[0].info.height), 3),
           "kernel_ms_per_launch": ms, "mpix_s": st["source_pixels"] / (ms * 1e-3) / 1e6,
           "algorithmic_bytes_per_launch": algo, "achieved_gb_s": algo / (ms * 1e-3) / 1e9, "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "index": "device pre-scan at upload" if dev[0].prescan_on_device else "serial host pre-scan",
           "parity_image_0": check_against_reference(J, ctx, files[0], pt, options, base, img_bytes, pitch, geo, sums[0]),
           "every_surface_equals_the_first_decode_of_its_file": bool(all_equal)}
    batch.close()
    for d in dev:
        d.close()
    for p_ in prepared:
        p_.close()
    ctx.free(base)
    res["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    return res


def photos_leg(J, ctx, threads, n_images=2048, steps=20, ramp_ms=150.0):
    """Real photographs: the reference's own fixtures (tests/golden/ref: tulips 640x480, zebra, st_peters, perf.jpg) tiled to one
    batch, kernel-only (inputs resident, index by the device pre-scan), RGB8888, every distinct file against the reference.  Their
    blocks are long and uneven (1.5-4 bit/px): what the synthetic legs do not show."""
    t0 = time.perf_counter()
    names = ("tulips", "zebra", "st_peters", "perf")
    d = os.path.join(ROOT, "tests", "golden", "ref")
    jpegs = [open(os.path.join(d, n + ".jpg"), "rb").read() for n in names]
    files = [jpegs[i % len(jpegs)] for i in range(n_images)]
    prepared = J.prepare_batch(files, device_prescan=True, threads=threads)
    pt = J.RGB8888
    geos, pitches, sizes = [], [], []
    for p in prepared[: len(jpegs)]:
        g = p.geometry(pt, 0)
        geos.append(g); pitches.append((g["canvas_w"] * g["bpp"] + 15) & ~15); sizes.append(pitches[-1] * g["canvas_h"])
    offs, total = [], 0
    for i in range(n_images):
        offs.append(total); total += (sizes[i % len(jpegs)] + 255) & ~255
    base = ctx.malloc(total)
    dev = J.upload_batch(ctx, prepared)
    outs = [(base + offs[i], pitches[i % len(jpegs)], geos[i % len(jpegs)]["canvas_w"], geos[i % len(jpegs)]["canvas_h"]) for i in range(n_images)]
    batch = J.Batch(ctx, dev, outs, [pt] * n_images, [0] * n_images)
    st = batch.stats
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < ramp_ms:
        for _ in range(4):
            batch.decode()
        ctx.sync()
    ctx.timer_start()
    for _ in range(steps):
        batch.decode()
    ctx.timer_stop()
    ctx.sync()
    ms = ctx.timer_elapsed_ms() / steps
    algo = st["output_bytes"] + st["scan_bytes"] + 4 * sum(p.n_mcus for p in prepared)
    sums = ctx.checksums(outs, [geos[i % len(jpegs)]["canvas_w"] * geos[i % len(jpegs)]["bpp"] for i in range(n_images)])
    nd = len(jpegs)
    checks = {names[k]: check_against_reference(J, ctx, jpegs[k], pt, 0, base + offs[k], sizes[k], pitches[k], geos[k], sums[k]) for k in range(nd)}
    res = {"workload": "%d photographs -> RGB8888: the reference's fixtures %s tiled, inputs resident" % (n_images, ", ".join(names)),
           "images": n_images, "distinct_images": nd,
           "bits_per_pixel": {names[k]: round(8.0 * len(jpegs[k]) / (prepared[k].info.width * prepared[k].info.height), 2) for k in range(nd)},
           "subsampling": {names[k]: "0x%02x" % prepared[k].info.subsample for k in range(nd)},
           "launches_per_step": st.get("n_launches"),
           "kernel_ms_per_step": ms, "mpix_s": st["source_pixels"] / (ms * 1e-3) / 1e6,
           "algorithmic_bytes_per_step": algo, "achieved_gb_s": algo / (ms * 1e-3) / 1e9, "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "index": "device pre-scan at upload" if dev[0].prescan_on_device else "serial host pre-scan",
           "parity": checks, "bit_exact": all(c["bit_exact"] for c in checks.values()),
           "every_surface_equals_the_first_decode_of_its_file": bool(all(sums[i] == sums[i % nd] for i in range(n_images)))}
    batch.close()
    for d_ in dev:
        d_.close()
    for p_ in prepared:
        p_.close()
    ctx.free(base)
    res["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    return res


def e2e_config_leg(J, ctx, what, jpegs, n_images, pt, threads, depth=4, batches=12):
    """One of the other BASELINE.json configurations END TO END: the batch's files in page-locked host memory (a buffer per image,
    side by side: a loader's arena) -> pixels resident in HBM through jda_pipeline (host parse + tables, H2D of the unfiltered scans,
    device filter + pre-scan + decode; `depth` batches overlapped), host work included.  Image 0 of the last batch against the
    reference, every surface of the last batch against the first decode of its file."""
    t0 = time.perf_counter()
    nd = len(jpegs)
    files = [jpegs[i % nd] for i in range(n_images)]
    one = J.PreparedImage(jpegs[0])
    geo = one.geometry(pt, 0)
    w, h = one.info.width, one.info.height
    one.close()
    pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15
    img_bytes = pitch * geo["canvas_h"]
    surf = [ctx.malloc(img_bytes * n_images) for _ in range(depth)]
    outs_of = [[(b + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for i in range(n_images)] for b in surf]
    arena = J.PinnedFiles(files)
    pipe = J.Pipeline(ctx, max_images=n_images, depth=depth, host_threads=min(threads, 8))
    packed = [pipe.pack_pinned(arena, list(range(n_images)), o, [pt] * n_images, [0] * n_images) for o in outs_of]
    inflight, failed, warm, tb0, t_submit = [], 0, depth + 3, 0.0, 0.0
    for k in range(warm + batches):
        if k == warm:
            while inflight:
                pipe.wait(inflight.pop(0))
            ctx.sync()
            tb0 = time.perf_counter()
        if len(inflight) == depth:
            failed += sum(1 for s_ in pipe.wait(inflight.pop(0)) if s_ != 0)
        ts = time.perf_counter()
        inflight.append(pipe.submit_packed(packed[k % depth], J.SUBMIT_PINNED_INPUT))
        if k >= warm:
            t_submit += time.perf_counter() - ts
    while inflight:
        failed += sum(1 for s_ in pipe.wait(inflight.pop(0)) if s_ != 0)
    ctx.sync()
    dt = time.perf_counter() - tb0
    last = outs_of[(warm + batches - 1) % depth]
    sums = ctx.checksums(last, [geo["canvas_w"] * geo["bpp"]] * n_images)
    pst = pipe.stats
    res = {"workload": what, "images_per_batch": n_images, "distinct_images": nd, "depth": depth, "batches": batches,
           "bits_per_pixel": round(8.0 * sum(len(j) for j in jpegs) / (nd * w * h), 3),
           "mpix_s": float(w) * h * n_images * batches / dt / 1e6, "images_per_s": n_images * batches / dt, "ms_per_batch": dt / batches * 1e3,
           "host_submit_ms_per_batch": t_submit / batches * 1e3, "failed_images": failed,
           "device_path_images": pst["device_images"], "host_path_images": pst["host_path_images"],
           "parity_image_0": check_against_reference(J, ctx, files[0], pt, 0, last[0][0], img_bytes, pitch, geo, sums[0]),
           "every_surface_equals_the_first_decode_of_its_file": bool(all(sums[i] == sums[i % nd] for i in range(n_images)))}
    pipe.close()
    arena.close()
    for b in surf:
        ctx.free(b)
    res["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    return res


def c1_leg(J):
    """BASELINE config 1, the reference's one published benchmark (README.md:32-38, examples/jpeg_perf_test/jpeg_perf_test.ino:8-53):
    test_images/tulips (640x480 4:2:0, restart interval per MCU row) -> RGB565 through the drop-in API with a draw callback that does
    nothing -- here JPEG_openRAM + JPEG_decode + JPEG_close of libjpegdec_amd.so from a C program (tests/capi_c/perf_user.c), full size
    and the three scaled decodes of the sketch; beside it the reference's own builds (oracle/_ref: SSE2 and scalar) on ONE thread of
    this host, the same convention.  A latency figure: one image at a time, files and pixels in host memory."""
    import subprocess

    t0 = time.perf_counter()
    path = os.path.join(ROOT, "tests", "golden", "ref", "tulips.jpg")
    exe = os.path.join(ROOT, "tests", "capi_c", "perf_user")
    if not os.path.exists(exe):
        subprocess.run(["make", "perfuser"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    res = {"workload": "tulips 640x480 4:2:0 (test_images/tulips.h) -> RGB565, one image at a time, open + decode(0, 0, options) + close with a no-op draw "
                       "callback (jpeg_perf_test.ino); the GPU path through the C flavour of the drop-in API (host memory -> JPEGDRAW strips in host memory)",
           "gpu_us_per_decode": {}, "reference_sse2_1_thread_us": {}, "reference_scalar_1_thread_us": {}}
    for tag, opt in (("full", 0), ("half", 2), ("quarter", 4), ("eighth", 8)):
        r = subprocess.run([exe, path, "0", str(opt), "1500" if opt == 0 else "500"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            j = {"error": r.stdout[-300:]}
        res["gpu_us_per_decode"][tag] = j.get("us_per_decode", j)
        if tag == "full":
            res["gpu_best_us"], res["draw_calls_per_decode"] = j.get("best_us"), j.get("draw_calls_per_decode")
    try:
        from oracle.loader import RefDecoder, ref_available
        jpeg = open(path, "rb").read()
        for key, simd in (("reference_sse2_1_thread_us", True), ("reference_scalar_1_thread_us", False)):
            if not ref_available(simd):
                continue
            dec = RefDecoder(simd)
            for tag, opt in (("full", 0), ("half", 2), ("quarter", 4), ("eighth", 8)):
                dec.bench([jpeg], 0, opt, 20, 1)
                r = dec.bench([jpeg], 0, opt, 400, 1)
                res[key][tag] = round(r["seconds"] / 400 * 1e6, 1)
    except Exception as e:
        res["reference_error"] = "%s: %s" % (type(e).__name__, e)
    g, c = res["gpu_us_per_decode"].get("full"), res["reference_sse2_1_thread_us"].get("full")
    if isinstance(g, float) and c:
        res["gpu_over_sse2_1_thread"] = round(c / g, 2)
    res["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    return res


def kernel_sources_sha():
    """What the decode kernel is compiled from: the PMC traffic file under profiles/ names the hash it was measured at."""
    import hashlib

    h = hashlib.sha256()
    for f in ("jda_kernels.hip", "jda_device_core.h"):
        h.update(open(os.path.join(ROOT, "jpegdec_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def run_config_legs(J, ctx, threads, only=None):
    """BASELINE.json configs 2, 3, 4 (one GPU's shard), 5 (full, 1/2, 1/4, 1/8) and the metric image at q98, one short leg each."""
    legs = [
        ("c2", "1024 x 1280x720 4:2:0 -> RGB8888 (BASELINE config 2)", (1280, 720, "4:2:0", 85, 4), 1024, J.RGB8888, 0),
        ("c3", "256 x 4096x4096 4:4:4 -> RGB8888 (BASELINE config 3)", (4096, 4096, "4:4:4", 85, 2), 256, J.RGB8888, 0),
        ("c4_shard", "1024 x 1920x1080 4:2:0 -> RGB8888 (one GPU's eighth of BASELINE config 4)", (1920, 1080, "4:2:0", 85, 2), 1024, J.RGB8888, 0),
        ("c5", "16 x 8192x8192 gray -> GRAY8, full size (BASELINE config 5)", (8192, 8192, "gray", 85, 2), 16, J.GRAY8, 0),
        ("c5_half", "16 x 8192x8192 gray -> GRAY8 at 1/2", (8192, 8192, "gray", 85, 2), 16, J.GRAY8, J.SCALE_HALF),
        ("c5_quarter", "16 x 8192x8192 gray -> GRAY8 at 1/4", (8192, 8192, "gray", 85, 2), 16, J.GRAY8, J.SCALE_QUARTER),
        ("c5_eighth", "16 x 8192x8192 gray -> GRAY8 at 1/8 (DC only)", (8192, 8192, "gray", 85, 2), 16, J.GRAY8, J.SCALE_EIGHTH),
        # the two scaled kernels as THROUGHPUT: a launch of 16 images is 23-97 us, launch-scale latency; 256 images make it 0.4-1.5 ms
        ("c5_quarter_256", "256 x 8192x8192 gray -> GRAY8 at 1/4 (a launch long enough to be a throughput figure)", (8192, 8192, "gray", 85, 2), 256, J.GRAY8, J.SCALE_QUARTER),
        ("c5_eighth_256", "256 x 8192x8192 gray -> GRAY8 at 1/8 (DC only; a launch long enough to be a throughput figure)", (8192, 8192, "gray", 85, 2), 256, J.GRAY8, J.SCALE_EIGHTH),
        ("q98", "64 x 4096x4096 4:2:0 at quality 98 (3.7 bit/px) -> RGB8888", (4096, 4096, "4:2:0", 98, 2), 64, J.RGB8888, 0),
        # colour thumbnails: the metric's images at 1/8 (what every progressive file's default decode is, too: DC values only)
        ("metric_eighth", "64 x 4096x4096 4:2:0 -> RGB8888 at 1/8 (DC only)", (4096, 4096, "4:2:0", 85, 16), 64, J.RGB8888, J.SCALE_EIGHTH),
    ]
    e2e_legs = [
        ("c2_e2e", "1024 x 1280x720 4:2:0 -> RGB8888 END TO END through jda_pipeline (BASELINE config 2)", (1280, 720, "4:2:0", 85, 8), 1024, J.RGB8888),
        ("c4_e2e", "1024 x 1920x1080 4:2:0 -> RGB8888 END TO END through jda_pipeline (one GPU's eighth of BASELINE config 4)", (1920, 1080, "4:2:0", 85, 8), 1024, J.RGB8888),
        ("vga_e2e", "2048 x 640x480 4:2:0 -> RGB8888 END TO END through jda_pipeline (the size of BASELINE config 1's image, in batches)", (640, 480, "4:2:0", 85, 8), 2048, J.RGB8888),
    ]
    out, cache = {}, {}
    for name, what, (w, h, sub, q, nd), n, pt, opt in legs:
        if only and name not in only:
            continue
        try:
            key = (w, h, sub, q, nd)
            if key not in cache:
                cache[key] = [cached_jpeg(w, h, sub, 1234 + i, quality=q) for i in range(nd)]
            out[name] = config_leg(J, ctx, name, what, cache[key], n, pt, opt, threads)
        except Exception as e:  # a leg that fails is reported, the headline stands
            out[name] = {"workload": what, "error": "%s: %s" % (type(e).__name__, e)}
    if not only or "photos" in only:
        try:
            out["photos"] = photos_leg(J, ctx, threads)
        except Exception as e:
            out["photos"] = {"workload": "photographs", "error": "%s: %s" % (type(e).__name__, e)}
    for name, what, (w, h, sub, q, nd), n, pt in e2e_legs:
        if only and name not in only:
            continue
        try:
            key = (w, h, sub, q, nd)
            if key not in cache:
                cache[key] = [cached_jpeg(w, h, sub, 1234 + i, quality=q) for i in range(nd)]
            out[name] = e2e_config_leg(J, ctx, what, cache[key], n, pt, threads)
        except Exception as e:
            out[name] = {"workload": what, "error": "%s: %s" % (type(e).__name__, e)}
    if not only or "c1" in only:
        try:
            out["c1"] = c1_leg(J)
        except Exception as e:
            out["c1"] = {"workload": "BASELINE config 1", "error": "%s: %s" % (type(e).__name__, e)}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="metric", choices=["metric", "c4"],
                    help="metric: batch x N images of --width x --height (weak scaling); c4: BASELINE config 4, 8192 x 1920x1080 4:2:0 split over the ranks (strong scaling)")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step (metric workload)")
    ap.add_argument("--total-images", type=int, default=8192, help="size of the list of the c4 workload")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic images cycled through the list (0: 16 for metric, 8 for c4)")
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--subsampling", default="4:2:0", choices=["4:2:0", "4:4:4", "4:2:2", "gray"])
    ap.add_argument("--pixel-type", default="rgb8888", choices=["rgb8888", "rgb565", "gray8"])
    ap.add_argument("--options", type=int, default=0)
    ap.add_argument("--quality", type=int, default=85, help="JPEG quality of the synthetic inputs (85 = the headline config, SURVEY 8d)")
    ap.add_argument("--restart-rows", type=int, default=0, help="encode the inputs with a restart marker every N MCU rows (0 = none, the headline config)")
    ap.add_argument("--ramp-ms", type=float, default=400.0, help="untimed decode launches before the warm-up steps until the GPU clocks have ramped (0: none)")
    ap.add_argument("--device-prescan", action="store_true", help="resident inputs: the block index is made on the GPU at upload (default: serial host pre-scan; the streamed pipeline always uses the device)")
    ap.add_argument("--e2e-batches", type=int, default=24, help="batches streamed through jda_pipeline for the end-to-end figure (0: skip)")
    ap.add_argument("--e2e-depth", type=int, default=4)
    ap.add_argument("--no-e2e-sweep", action="store_true", help="end-to-end leg: skip the host-thread sweep, the pageable-input and the cold-input runs")
    ap.add_argument("--e2e-cold-gb", type=float, default=2.0, help="end-to-end leg, cold input: GB of distinct page-locked buffers the batches cycle through")
    ap.add_argument("--e2e-distinct", type=int, default=16, help="distinct files a batch of the end-to-end leg cycles through (metric workload; the resident batch keeps --distinct)")
    ap.add_argument("--configs", default="", help="comma-separated subset of the config legs (default: all of c2,c3,c4_shard,c5,c5_half,c5_quarter,c5_eighth,c5_quarter_256,c5_eighth_256,q98,metric_eighth,photos,c2_e2e,c4_e2e,vga_e2e,c1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs over the other BASELINE.json configurations (`configs` in the line; N = 1, metric workload only)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the control collectives (nccl = RCCL)")
    ap.add_argument("--device-module", default="jpegdec_amd", help="module providing the device half (tests: tests.stub_device, to run the N-rank control flow without a GPU)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves -- one process per GPU,
    rank r -> GPU r, rendezvous on 127.0.0.1 at a free port -- by re-executing this script under torch.distributed.run.
    Rank 0's JSON line is the only thing on stdout; the exit status is the launcher's (non-zero when any rank fails)."""
    import socket
    import subprocess

    if args.dist_backend == "nccl":           # RCCL wants one device per rank ("Duplicate GPU detected" otherwise): say so before forking
        import importlib

        n_dev = importlib.import_module(args.device_module).load_library().jda_device_count()
        if n_dev < args.gpus:
            print("bench.py: --gpus %d but %d HIP device(s) visible; RCCL needs one GPU per rank" % (args.gpus, n_dev), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what the host driver supports (RCCL fails without it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def run(args, J, out=sys.stdout):
    """The benchmark proper.  J: the jpegdec_amd module (tests pass a stub of its device half to run this control flow
    under gloo without a GPU)."""
    from jpegdec_amd.sharding import Group, cpu_model, env_rank_world, place_rank, shard_range, verify_exactly_once

    rank, world, local_rank = env_rank_world()
    if world != args.gpus:                    # (main() starts the ranks itself when there is no launcher: a mismatch is a wrong command line)
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    on_gpu_backend = args.dist_backend == "nccl"
    dist_on = world > 1 or bool(os.environ.get("JDA_FORCE_DIST"))
    if dist_on and on_gpu_backend:
        import torch

        torch.cuda.set_device(local_rank)
    group = Group(backend=args.dist_backend, device="cuda" if (dist_on and on_gpu_backend) else None)   # RCCL over xGMI; control traffic only

    pt = {"rgb8888": J.RGB8888, "rgb565": J.RGB565_LE, "gray8": J.GRAY8}[args.pixel_type]
    if args.subsampling == "gray" and pt == J.RGB8888:
        pt = J.GRAY8   # JPEGPutMCUGray never writes 32-bit pixels

    n_dev = max(1, J.load_library().jda_device_count())
    ctx = J.Context(local_rank % n_dev)     # one process per GPU; raises without a GPU: there is no CPU fallback
    threads, placement = place_rank(group, ctx.pci_bus_id())   # this rank's share of the host cores, on its GPU's NUMA node

    # ---- ONE list of images for the whole job; this rank's contiguous shard of it
    if args.workload == "c4":
        width, height, sub = 1920, 1080, "4:2:0"
        n_total = args.total_images
        n_distinct = args.distinct or 8
    else:
        width, height, sub = args.width, args.height, args.subsampling
        n_total = args.batch * world
        n_distinct = args.distinct or min(16, args.batch)      # (the 4096x4096 q85 set is cached under bench_cache/; other shapes are made on first use)
    jpegs = [cached_jpeg(width, height, sub, 1234 + i, quality=args.quality, restart_rows=args.restart_rows) for i in range(n_distinct)]
    bits_px = 8.0 * sum(len(j) for j in jpegs) / (len(jpegs) * width * height)
    lo, hi = shard_range(n_total, rank, world)
    my_files = [jpegs[i % n_distinct] for i in range(lo, hi)]
    n_mine = hi - lo

    # ---- host prepare of the shard on this rank's threads, upload, launch plan
    t_prep0 = time.perf_counter()
    one = J.PreparedImage(jpegs[0], device_prescan=args.device_prescan)
    t_prep1 = time.perf_counter() - t_prep0                                    # one image, one thread
    one.close()
    t_par0 = time.perf_counter()
    prepared = J.prepare_batch(my_files, device_prescan=args.device_prescan, threads=threads)
    t_par = (time.perf_counter() - t_par0) / max(n_mine, 1)
    geo = prepared[0].geometry(pt, args.options)
    pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15
    img_bytes = pitch * geo["canvas_h"]
    out_base = ctx.malloc(img_bytes * n_mine)
    t_up0 = time.perf_counter()
    dev_images = J.upload_batch(ctx, prepared)
    t_up = (time.perf_counter() - t_up0) / max(n_mine, 1)
    outputs = [(out_base + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for i in range(n_mine)]
    batch = J.Batch(ctx, dev_images, outputs, [pt] * n_mine, [args.options] * n_mine)
    st = batch.stats

    def barrier():
        ctx.sync()                       # our launches go to the context's own HIP stream
        if dist_on:
            group.barrier()
            if on_gpu_backend:
                import torch

                torch.cuda.synchronize()

    # Clock ramp, untimed and outside the W warm-up steps: a step is ~1.2 ms of GPU work, so W = 3 steps end long before the
    # GPU's power management has raised the clocks to their level under load (measured: a 5-step run straight after the
    # upload is 15 % slower per launch than a 20-step one).  The same launches, kept going for --ramp-ms.
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(8):
            batch.decode()
        ctx.sync()
    for _ in range(args.warmup):
        batch.decode()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                    # HIP events on the launch stream
    for _ in range(args.steps):
        batch.decode()
    ctx.timer_stop()
    barrier()
    t1 = time.perf_counter()
    elapsed = group.max(t1 - t0)         # slowest rank
    kernel_ms = ctx.timer_elapsed_ms() / args.steps
    total_pixels = group.sum(st["source_pixels"])       # whole job, all ranks

    # ---- the sharding, proved: per-image checksums made where the pixels are, all-reduced; every image exactly once and
    # equal to the single-GPU decode of its file (rank 0's own decode of the distinct files, broadcast with the same reduce)
    row_bytes = geo["canvas_w"] * geo["bpp"]
    sums = ctx.checksums(outputs, [row_bytes] * n_mine)
    ref_sums = np.zeros(n_distinct, dtype=np.uint64)
    if rank == 0:
        have = {}
        for i in range(lo, hi):
            have.setdefault(i % n_distinct, sums[i - lo])
        missing = [d for d in range(n_distinct) if d not in have]
        if missing:                      # (a shard smaller than the set of distinct files: decode the rest once, here)
            extra_prep = [J.PreparedImage(jpegs[d]) for d in missing]
            extra_dev = J.upload_batch(ctx, extra_prep)
            ex_base = ctx.malloc(img_bytes * len(missing))
            ex_out = [(ex_base + k * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for k in range(len(missing))]
            eb = J.Batch(ctx, extra_dev, ex_out, [pt] * len(missing), [args.options] * len(missing))
            eb.decode(); ctx.sync()
            for d, v in zip(missing, ctx.checksums(ex_out, [row_bytes] * len(missing))):
                have[d] = v
            eb.close()
            for d_ in extra_dev:
                d_.close()
            for p_ in extra_prep:
                p_.close()
            ctx.free(ex_base)
        for d in range(n_distinct):
            ref_sums[d] = have[d]
    ref_sums = group.sum_int64_vector(ref_sums.view(np.int64)).view(np.uint64)      # = a broadcast from rank 0
    sharding = verify_exactly_once(group, n_total, lo, sums, expected_of=lambda i: int(ref_sums[i % n_distinct]))
    sharding.update({"list": "%d images (%d distinct files), rank r owns the contiguous block shard_range(n, r, %d)" % (n_total, n_distinct, world),
                     "images_this_rank": n_mine, "host_placement": placement})

    # ---- parity spot check outside the timed region: first image of this rank vs the oracle
    # (every DISTINCT file of the timed batch is checked, at its first place in this rank's shard)
    parity = None
    if not args.no_parity and rank == 0:
        first = {}
        for i in range(lo, hi):
            first.setdefault(i % n_distinct, i - lo)
        checks = [check_against_reference(J, ctx, jpegs[d], pt, args.options, out_base + k * img_bytes, img_bytes, pitch, geo, sums[k]) for d, k in sorted(first.items())]
        parity = dict(checks[0])
        parity["distinct_files_checked"] = len(checks)
        parity["bit_exact"] = None if any(c["bit_exact"] is None for c in checks) else all(c["bit_exact"] for c in checks)
        parity["device_checksum_equals_host_checksum"] = all(c.get("device_checksum_equals_host_checksum", False) for c in checks)

    # ---- end to end: the same files streamed from host memory through jda_pipeline (host parse + tables, H2D of the unfiltered
    # scans, device filter + pre-scan + decode; batches overlapped on three streams), host work included, pixels stay in HBM.
    # The input lies in page-locked memory (what a loader that feeds a GPU reads into) and is submitted with JDA_SUBMIT_PINNED_INPUT:
    # the copy engine takes the files where they are, no host core copies them.  Besides the headline figure (this rank's threads,
    # at most 8): the same with 1, 2, 4, 8 host threads -- an 8-GPU node leaves a rank a few cores --, with COLD input (>= 2 GB of
    # distinct buffers: nothing the host or the copy engine reads is in a cache), and with pageable input (the pipeline's own
    # page-locked mirror, filled by the workers).
    e2e = None
    if args.e2e_batches > 0:
        eb = min(n_mine, args.batch) if args.workload == "metric" else min(n_mine, 256)
        depth = max(1, min(args.e2e_depth, 4))
        e2e_distinct = n_distinct
        pool = jpegs
        if args.workload == "metric" and args.e2e_distinct > n_distinct:
            # more distinct files per batch than the resident leg keeps: the pre-scan's late rounds are longer (its chains of changed
            # entry states differ from file to file), which a batch of two files flatters
            e2e_distinct = min(args.e2e_distinct, eb)
            pool = [cached_jpeg(width, height, sub, 1234 + i, quality=args.quality, restart_rows=args.restart_rows) for i in range(e2e_distinct)]
        picks = [(lo + i) % e2e_distinct for i in range(eb)]
        surf = [out_base] if eb * depth > n_mine else [out_base + k * eb * img_bytes for k in range(depth)]
        extra = [ctx.malloc(img_bytes * eb) for _ in range(depth - len(surf))]
        surf += extra
        outs_of = [[(base + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for i in range(eb)] for base in surf]
        have_pinned = hasattr(J, "PinnedFiles")
        # every image of a batch has a page-locked buffer of its own, as a loader's arena has (copies of the distinct files, side by side:
        # one copy command a batch; 64 images pointing into 16 buffers would be 64 runs that cannot be joined)
        hot = J.PinnedFiles([pool[k] for k in picks]) if have_pinned else None

        def e2e_run(host_threads, batches, source):
            """source: "pinned" (hot: the same page-locked files every batch), "pageable" (Python bytes through the mirror),
            or a PinnedFiles arena of many copies (cold)."""
            pipe = J.Pipeline(ctx, max_images=eb, depth=depth, host_threads=host_threads)
            if source == "pageable" or not have_pinned:
                packed = [pipe.pack([pool[k] for k in picks], o, [pt] * eb, [args.options] * eb) for o in outs_of]
                flags, n_sets = 0, len(packed)
            elif source == "pinned":
                packed = [pipe.pack_pinned(hot, list(range(eb)), o, [pt] * eb, [args.options] * eb) for o in outs_of]
                flags, n_sets = J.SUBMIT_PINNED_INPUT, len(packed)
            else:                                   # cold: batch k reads copies the copy engine has not seen for len(source.addrs) / eb batches
                n_sets = max(depth, len(source.addrs) // eb)
                packed = [pipe.pack_pinned(source, [(k * eb + i) % len(source.addrs) for i in range(eb)], outs_of[k % depth], [pt] * eb, [args.options] * eb) for k in range(n_sets)]
                flags = J.SUBMIT_PINNED_INPUT
            submit = (lambda k: pipe.submit_packed(packed[k % n_sets], flags)) if have_pinned else (lambda k: pipe.submit_packed(packed[k % n_sets]))
            inflight, t_submit, warm = [], 0.0, max(2, depth)
            tb0, n_failed = 0.0, 0                # (failures are counted, the barriers completed, and the ranks fail together afterwards)
            for k in range(warm + batches):
                if k == warm:
                    while inflight:
                        pipe.wait(inflight.pop(0))
                    barrier()
                    tb0 = time.perf_counter()
                if len(inflight) == depth:
                    n_failed += sum(1 for s_ in pipe.wait(inflight.pop(0)) if s_ != 0)
                ts = time.perf_counter()
                inflight.append(submit(k))
                if k >= warm:
                    t_submit += time.perf_counter() - ts
            while inflight:
                n_failed += sum(1 for s_ in pipe.wait(inflight.pop(0)) if s_ != 0)
            barrier()
            dt = group.max(time.perf_counter() - tb0)
            n_failed = int(group.sum(float(n_failed)))
            if n_failed:
                raise SystemExit("bench.py: %d image(s) of the end-to-end leg did not decode" % n_failed)
            n_img = eb * batches
            px_all = group.sum(float(geo["out_w"] * geo["out_h"] * n_img))
            pst = pipe.stats
            pipe.close()
            return {"mpix_s": px_all / dt / 1e6, "ms_per_image": dt / n_img * 1e3, "host_submit_ms_per_image": t_submit / n_img * 1e3,
                    "host_threads": host_threads, "batches": batches, "device_path_images": pst["device_images"], "host_path_images": pst["host_path_images"]}

        main_threads = min(threads, 8)
        # (untimed, like the clock ramp: the first few dozen batches that read a fresh page-locked arena run 10-25 % below the rate
        # every later one has -- profiles/r04_e2e_input_modes.txt, first against second pass over the same six runs)
        if have_pinned and not args.no_e2e_sweep:
            e2e_run(main_threads, 2 * args.e2e_batches, "pinned")
        e2e = e2e_run(main_threads, args.e2e_batches, "pinned")
        e2e.update({"images_per_batch": eb, "depth": depth, "distinct_images": min(e2e_distinct, eb),
                    "input": "page-locked, JDA_SUBMIT_PINNED_INPUT: a buffer per image of the batch, side by side (DMA from where they lie, one copy command a batch)" if have_pinned else "pageable",
                    "what": "files in host memory -> pixels resident in HBM through jda_pipeline: host parse + tables, H2D of the unfiltered scans, "
                            "device marker filter + per-block index + decode, batches overlapped; whole job, all ranks; the same files are submitted every batch"})
        if have_pinned and rank == 0 and world == 1 and not args.no_e2e_sweep:
            short = max(8, args.e2e_batches // 2)
            e2e["host_thread_sweep"] = {str(t_): round(e2e_run(t_, short, "pinned")["mpix_s"]) for t_ in (1, 2, 4, 8)}
            e2e["pageable_input"] = {str(t_): round(e2e_run(t_, short, "pageable")["mpix_s"]) for t_ in (2, 8)}
            # cold input: >= 2 GB of distinct page-locked buffers (copies of the distinct files), each read once per ~2 GB of traffic
            t_c = time.perf_counter()
            n_copies = max(eb * depth, int(args.e2e_cold_gb * (1 << 30) / (sum(len(f) for f in pool) / len(pool))) + 1)
            cold = J.PinnedFiles([pool[k % len(pool)] for k in range(n_copies)])
            r = e2e_run(main_threads, max(short, 2 * n_copies // eb), cold)
            e2e["cold_input"] = {"mpix_s": r["mpix_s"], "vs_hot": r["mpix_s"] / e2e["mpix_s"], "distinct_buffers": n_copies, "bytes": cold.bytes,
                                 "batches": r["batches"], "setup_s": round(time.perf_counter() - t_c, 2),
                                 "what": "every batch reads page-locked buffers nobody has touched for %d batches (%.1f GB in all)" % (n_copies // eb, cold.bytes / 2 ** 30)}
            cold.close()
        if hot:
            hot.close()
        for p_ in extra:
            ctx.free(p_)

    configs = None
    if not args.no_configs and rank == 0 and world == 1 and args.workload == "metric":
        configs = run_config_legs(J, ctx, threads, only=[c for c in args.configs.split(",") if c] or None)

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        try:
            cpu = cpu_baseline(jpegs, pt, placement["cores_usable"], {k: placement[k] for k in ("affinity_cpus", "cgroup_cpu_quota", "os_cpu_count")}, cpu_model())
        except Exception as e:
            cpu = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "reference", "sample": "failed: %s" % e}

    if rank == 0:
        px_per_step = st["source_pixels"]
        value = total_pixels * args.steps / elapsed / 1e6
        # algorithmic bytes per launch: output + filtered scan + 4 B/MCU index
        n_mcus = sum(p.n_mcus for p in prepared)
        algo_bytes = st["output_bytes"] + st["scan_bytes"] + 4 * n_mcus
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch: PMC counters cannot be read from inside this process, so the figure
        # comes from the committed rocprofv3 --pmc passes of the same workload (profiles/), scaled to
        # this batch; null for any other workload
        # The file names the hash of the kernel sources it was measured at (tools/gpu_profile.sh): with other sources in the tree the
        # figure is stale and the line says null + why.
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
        if (os.path.exists(tp) and args.workload == "metric" and (args.width, args.height, args.subsampling, args.pixel_type, args.options, args.quality)
                == (4096, 4096, "4:2:0", "rgb8888", 0, 85)):
            tj = json.load(open(tp))
            if tj.get("kernel_sources_sha16") == kernel_sources_sha():
                traffic = tj["hbm_bytes_per_image"] * n_mine
                traffic_src = ("profiles/r06_pmc_traffic.json (tools/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 "
                               "correction; measured at kernel sources %s = this tree's)" % tj["kernel_sources_sha16"])
            else:
                traffic_src = ("stale: profiles/r06_pmc_traffic.json was measured at kernel sources %s, this tree has %s -- rerun tools/gpu_profile.sh"
                               % (tj.get("kernel_sources_sha16"), kernel_sources_sha()))
                print("bench.py: " + traffic_src, file=sys.stderr)
        line = {
            "metric": "Mpixels/s decoded",
            "value": value,
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.workload == "metric" else "strong",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": "list of %d %dx%d baseline %s JPEGs (%d per GPU) -> %s, inputs resident in HBM"
                            % (n_total, width, height, sub, n_mine, args.pixel_type),
                "images_per_gpu_per_step": n_mine,
                "distinct_images": len(jpegs),
                "bits_per_pixel": round(bits_px, 3),
                "options": args.options,
                "parallelism": "one image list sharded in contiguous blocks over %d GPU(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "jda_decode_tiles_persistent<2, 1, 1, 0> (4:2:0, 24-bit multiplies, RGB8888 plain case)",
                "kernel_ms_per_launch": kernel_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "cpu_baseline": cpu,
            "parity": parity,
            "sharding": sharding,
            "configs": configs,
            "end_to_end": e2e,
            "end_to_end_mpix_s": e2e["mpix_s"] if e2e else None,
            "host_prepare_ms_per_image": t_prep1 * 1e3,
            "host_prepare_threads_ms_per_image": t_par * 1e3,
            "host_threads": threads,
            "upload_ms_per_image": t_up * 1e3,
            "device_prescan": bool(dev_images[0].prescan_on_device),
            "clock_ramp_ms": args.ramp_ms,                # untimed launches before the W warm-up steps (see above)
            "kernel_only_mpix_s": px_per_step / (kernel_ms * 1e-3) / 1e6,
            "dist": {"backend": (args.dist_backend if dist_on else "none"), "ranks": world},
        }
        print(json.dumps(line), file=out)
        out.flush()

    batch.close()
    for d in dev_images:
        d.close()
    for p_ in prepared:
        p_.close()
    ctx.free(out_base)
    ctx.close()
    group.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    import importlib

    run(args, importlib.import_module(args.device_module))


if __name__ == "__main__":
    main()
