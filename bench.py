#!/usr/bin/env python3
"""bench.py -- Mpixels/s decoded on the BASELINE.json metric workload.

A "step" is one pass of the hot path (the MCU loops of DecodeJPEG, reference jpeg.inl:5109-5353,
here one kernel launch) over one batch of synthetic baseline JPEGs whose inputs (filtered scan,
per-MCU index, tables) are already resident in HBM.  Default workload = the metric's own
configuration: 4096x4096 baseline 4:2:0 -> RGB8888.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; images are independent so each rank decodes its own batch (weak scaling, no
data-path collective); RCCL is used only for the barrier that brackets the timed region and for
the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
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
    path = os.path.join(d, name)
    if os.path.exists(path):
        return open(path, "rb").read()
    data = synth_jpeg(width, height, subsampling, seed=seed, quality=quality, restart_rows=restart_rows)
    with open(path + ".tmp%d" % os.getpid(), "wb") as f:
        f.write(data)
    os.replace(path + ".tmp%d" % os.getpid(), path)
    return data


def cpu_baseline(jpegs, pixel_type, threads, target_cpu_seconds=16.0):
    """The reference's own default (SSE2) build from oracle/_ref timed on this host's cores.
    Test-infrastructure use of oracle/: a reported baseline, never the measured product."""
    from oracle.loader import RefDecoder, ref_available

    if not ref_available(simd=True):
        return None
    ref = RefDecoder(simd=True)
    r0 = ref.bench(jpegs[:1], pixel_type, 0, 1, 1)                  # calibrate: one decode, one thread
    per_image = max(r0["seconds"], 1e-4)
    # every thread walks the image list with stride `threads`: give each thread >= 1 image
    imgs = list(jpegs) * max(1, (threads + len(jpegs) - 1) // len(jpegs))
    reps = max(1, int(round(target_cpu_seconds / per_image / len(imgs))))
    r = ref.bench(imgs, pixel_type, 0, reps, threads)
    return {
        "value": r["pixels"] / r["seconds"] / 1e6,
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": "reference",
        "sample": "%d decodes of the workload images (%d distinct), JPEGDEC default x86-64 build (SSE2), "
                  "one JPEGDEC object per thread, no-op draw callback, %.1f s wall"
                  % (reps * len(imgs), len(jpegs), r["seconds"]),
        "single_thread_mpix_s": r0["pixels"] / r0["seconds"] / 1e6,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic images cycled through the batch")
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--subsampling", default="4:2:0", choices=["4:2:0", "4:4:4", "4:2:2", "gray"])
    ap.add_argument("--pixel-type", default="rgb8888", choices=["rgb8888", "rgb565", "gray8"])
    ap.add_argument("--options", type=int, default=0)
    ap.add_argument("--quality", type=int, default=85, help="JPEG quality of the synthetic inputs (85 = the headline config, SURVEY 8d)")
    ap.add_argument("--restart-rows", type=int, default=0, help="encode the inputs with a restart marker every N MCU rows (0 = none, the headline config)")
    ap.add_argument("--ramp-ms", type=float, default=400.0, help="untimed decode launches before the warm-up steps until the GPU clocks have ramped (0: none)")
    ap.add_argument("--device-prescan", action="store_true", help="JDA_PREPARE_DEVICE_PRESCAN: the block index is made on the GPU at upload (restart intervals, or the self-synchronising segment walk)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the control collectives (nccl = RCCL)")
    args = ap.parse_args()

    from jpegdec_amd.sharding import Group, env_rank_world

    rank, world, local_rank = env_rank_world()
    if world != args.gpus and rank == 0:
        print("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    on_gpu_backend = args.dist_backend == "nccl"
    if world > 1 and on_gpu_backend:
        import torch

        torch.cuda.set_device(local_rank)
    group = Group(backend=args.dist_backend, device="cuda" if (world > 1 and on_gpu_backend) else None)   # RCCL over xGMI; control traffic only

    import jpegdec_amd as J

    pt = {"rgb8888": J.RGB8888, "rgb565": J.RGB565_LE, "gray8": J.GRAY8}[args.pixel_type]
    if args.subsampling == "gray" and pt == J.RGB8888:
        pt = J.GRAY8   # JPEGPutMCUGray never writes 32-bit pixels

    # ---- inputs: `distinct` synthetic JPEGs, prepared on the host, `batch` resident copies in HBM
    jpegs = [cached_jpeg(args.width, args.height, args.subsampling, 1234 + i, quality=args.quality, restart_rows=args.restart_rows) for i in range(args.distinct)]
    bits_px = 8.0 * sum(len(j) for j in jpegs) / (len(jpegs) * args.width * args.height)
    n_dev = max(1, J.load_library().jda_device_count())
    ctx = J.Context(local_rank % n_dev)     # one process per GPU; raises without a GPU: there is no CPU fallback
    t_prep0 = time.perf_counter()
    prepared = [J.PreparedImage(j, device_prescan=args.device_prescan) for j in jpegs]
    t_prep = (time.perf_counter() - t_prep0) / len(jpegs)
    # the same on all host threads (jda_prepare_batch): what the host stage sustains for a batch
    t_par = float("nan")
    if world == 1:                                       # (N > 1: the ranks would only fight over the same host cores)
        n_par = max(len(jpegs), min(4 * args.batch, 4 * (os.cpu_count() or 1)))
        t_par0 = time.perf_counter()
        par = J.prepare_batch([jpegs[i % len(jpegs)] for i in range(n_par)], device_prescan=args.device_prescan, threads=0)
        t_par = (time.perf_counter() - t_par0) / n_par
        for p_ in par:
            p_.close()
        del par
    geo = prepared[0].geometry(pt, args.options)
    pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15
    img_bytes = pitch * geo["canvas_h"]
    dev_images, outputs = [], []
    out_base = ctx.malloc(img_bytes * args.batch)
    t_up0 = time.perf_counter()
    dev_images = J.upload_batch(ctx, [prepared[i % len(prepared)] for i in range(args.batch)])
    for i in range(args.batch):
        outputs.append((out_base + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]))
    t_up = (time.perf_counter() - t_up0) / args.batch      # H2D (+ the device pre-scan when it is used), the whole batch at once
    prescan_rounds = int(ctx.lib.jda_last_prescan_rounds(ctx.handle)) if args.device_prescan else 0
    batch = J.Batch(ctx, dev_images, outputs, [pt] * args.batch, [args.options] * args.batch)
    st = batch.stats

    def barrier():
        ctx.sync()                       # our launches go to the context's own HIP stream
        if world > 1:
            group.barrier()
            if on_gpu_backend:
                import torch

                torch.cuda.synchronize()

    # Clock ramp, untimed and outside the W warm-up steps: a step is ~1.3 ms of GPU work, so W = 3 steps end long before the
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

    # ---- parity spot check outside the timed region: first image of this rank vs the oracle
    parity = None
    if not args.no_parity and rank == 0:
        try:
            from oracle.loader import OracleDecoder, RefDecoder, ref_available
            got = ctx.to_host(out_base, img_bytes).reshape(geo["canvas_h"], pitch)[:, : geo["canvas_w"] * geo["bpp"]]
            if ref_available(False):
                want = RefDecoder(False).decode_cb(jpegs[0], pt, args.options)["canvas"][: geo["canvas_h"], : geo["canvas_w"] * geo["bpp"]]
                got = got[: geo["out_h"]]
                want = want[: geo["out_h"]]
                parity = {"checker": "oracle/_ref scalar reference", "bit_exact": bool(np.array_equal(got, want))}
            else:
                rc, want, _ = OracleDecoder().decode_canvas(jpegs[0], pt, args.options)
                parity = {"checker": "oracle restatement", "bit_exact": bool(rc == 1 and np.array_equal(got, want))}
        except Exception as e:  # checker missing: report, do not fail the measurement
            parity = {"checker": "unavailable: %s" % e, "bit_exact": None}

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        try:
            cpu = cpu_baseline(jpegs, pt, threads=os.cpu_count() or 1)
        except Exception as e:
            cpu = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "reference", "sample": "failed: %s" % e}

    if rank == 0:
        px_per_step = st["source_pixels"]
        value = total_pixels * args.steps / elapsed / 1e6
        # algorithmic bytes per launch: output + filtered scan + 4 B/MCU index
        n_mcus = sum(p.n_mcus for p in prepared) * (args.batch // len(prepared)) + sum(
            p.n_mcus for p in prepared[: args.batch % len(prepared)])
        algo_bytes = st["output_bytes"] + st["scan_bytes"] + 4 * n_mcus
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch: PMC counters cannot be read from inside this process, so the figure
        # comes from the committed rocprofv3 --pmc passes of the same workload (profiles/), scaled to
        # this batch; null for any other workload
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")
        if (os.path.exists(tp) and (args.width, args.height, args.subsampling, args.pixel_type, args.options)
                == (4096, 4096, "4:2:0", "rgb8888", 0)):
            traffic = json.load(open(tp))["hbm_bytes_per_image"] * args.batch
            traffic_src = "profiles/r01_final_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)"
        line = {
            "metric": "Mpixels/s decoded",
            "value": value,
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": "batch of %d %dx%d baseline %s JPEGs per GPU -> %s, inputs resident in HBM"
                            % (args.batch, args.width, args.height, args.subsampling, args.pixel_type),
                "images_per_gpu_per_step": args.batch,
                "distinct_images": len(jpegs),
                "bits_per_pixel": round(bits_px, 3),
                "options": args.options,
                "parallelism": "images sharded, %d GPU(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "jda_decode_tiles_persistent<MODE,FAST>",
                "kernel_ms_per_launch": kernel_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "cpu_baseline": cpu,
            "parity": parity,
            "host_prepare_ms_per_image": t_prep * 1e3,
            "host_prepare_all_threads_ms_per_image": (t_par * 1e3) if t_par == t_par else None,
            "host_threads": os.cpu_count(),
            "upload_ms_per_image": t_up * 1e3,
            "device_prescan": bool(dev_images[0].prescan_on_device),
            "device_prescan_rounds": prescan_rounds,      # speculative rounds of the marker-less segment walk (0: not used)
            "clock_ramp_ms": args.ramp_ms,                # untimed launches before the W warm-up steps (see above)
            "kernel_only_mpix_s": px_per_step / (kernel_ms * 1e-3) / 1e6,
            # host prepare (all threads) + upload + kernel, one after the other (no overlap between the stages)
            "end_to_end_mpix_s_no_overlap": (args.width * args.height / 1e6) / ((t_par if t_par == t_par else t_prep) + t_up + kernel_ms * 1e-3 / args.batch),
        }
        print(json.dumps(line))

    batch.close()
    for d in dev_images:
        d.close()
    ctx.free(out_base)
    ctx.close()
    group.close()


if __name__ == "__main__":
    main()
