"""bench.py's N > 1 control flow under gloo, world_size 2, without a GPU: the device half of jpegdec_amd is replaced by
tests/stub_device.py, everything else is the real code -- ONE image list sharded with shard_range, host placement, barrier /
max / sum collectives, the all-reduced checksum + count vectors (every image exactly once, equal to rank 0's single-"GPU"
values), the JSON record.  Plus the negative cases of the proof itself."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path, workload):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    from tests import stub_device

    argv = ["--gpus", str(world), "--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--dist-backend", "gloo", "--no-parity",
            "--no-cpu-baseline", "--e2e-batches", "2"]
    if workload == "c4":
        argv += ["--workload", "c4", "--total-images", "21", "--distinct", "5"]
    else:
        argv += ["--width", "333", "--height", "217", "--batch", "5", "--distinct", "2"]
    args = bench.parse_args(argv)
    if workload == "c4":
        # a small stand-in for the 1920x1080 files: the flow is what is under test
        real = bench.cached_jpeg
        bench.cached_jpeg = lambda w, h, s, seed, quality=85, restart_rows=0: real(160, 96, s, seed, quality=quality, restart_rows=restart_rows)
    with open(out_path if rank == 0 else os.devnull, "w") as f:
        bench.run(args, stub_device, out=f)


@pytest.mark.parametrize("workload", ["metric", "c4"])
def test_bench_flow_two_ranks(workload, tmp_path, product_lib):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    port = 29600 + (os.getpid() % 2000) + (7 if workload == "c4" else 0)
    out_path = str(tmp_path / "line.json")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out_path, workload)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    line = json.loads(open(out_path).read().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["metric"] == "Mpixels/s decoded"
    sh = line["sharding"]
    assert sh["decoded_exactly_once"] and sh["checked_against_single_gpu"]
    assert sh["images"] == (21 if workload == "c4" else 10)
    assert line["scaling"] == ("strong" if workload == "c4" else "weak")
    assert line["dist"] == {"backend": "gloo", "ranks": 2}
    assert line["end_to_end"]["batches"] == 2
    assert sh["host_placement"]["threads"] >= 1


def test_exactly_once_proof_catches_violations():
    from jpegdec_amd.sharding import Group, verify_exactly_once

    class Two(Group):                     # two "ranks" folded into one process: the sum is done by hand
        def __init__(self, other):
            self.rank, self.world, self.local_rank, self.dist, self.device, self.backend = 0, 2, 0, None, None, "none"
            self.other = other

        def sum_int64_vector(self, values):
            return np.asarray(values, dtype=np.int64) + next(self.other)

    def peer(n, lo, sums):
        cs = np.zeros(n, np.uint64); cnt = np.zeros(n, np.int64)
        cs[lo: lo + len(sums)] = sums; cnt[lo: lo + len(sums)] = 1
        yield cs.view(np.int64)
        yield cnt

    exp = [11, 22, 33, 44, 55]
    ok = verify_exactly_once(Two(peer(5, 3, exp[3:])), 5, 0, exp[:3], expected_of=lambda i: exp[i])
    assert ok["decoded_exactly_once"] and ok["images"] == 5
    with pytest.raises(AssertionError, match="decoded"):          # image 2 taken by both ranks
        verify_exactly_once(Two(peer(5, 2, exp[2:])), 5, 0, exp[:3], expected_of=lambda i: exp[i])
    with pytest.raises(AssertionError, match="decoded"):          # image 3 taken by nobody
        verify_exactly_once(Two(peer(5, 4, exp[4:])), 5, 0, exp[:3], expected_of=lambda i: exp[i])
    with pytest.raises(AssertionError, match="checksum"):         # a rank's pixels differ from the single-GPU decode
        verify_exactly_once(Two(peer(5, 3, [44, 56])), 5, 0, exp[:3], expected_of=lambda i: exp[i])


def test_cpu_quota_and_placement_helpers():
    from jpegdec_amd.sharding import _parse_cpulist, cpu_model, cpu_quota

    cores, detail = cpu_quota()
    assert 1 <= cores <= detail["affinity_cpus"]
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert isinstance(cpu_model(), str)


def test_bench_py_gpus_2_starts_its_own_ranks(tmp_path, product_lib):
    """`python bench.py --gpus 2` as the driver runs it -- no torch.distributed.run around it, no WORLD_SIZE -- must start two
    ranks itself and print ONE line with n_gpus == 2 (round 2's bench silently ran world = 1)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ramp-ms", "0",
           "--dist-backend", "gloo", "--device-module", "tests.stub_device", "--no-parity", "--no-cpu-baseline", "--no-configs", "--e2e-batches", "2",
           "--width", "333", "--height", "217", "--batch", "5", "--distinct", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dist"] == {"backend": "gloo", "ranks": 2}
    assert line["sharding"]["decoded_exactly_once"] and line["sharding"]["images"] == 10


def test_bench_py_refuses_a_launcher_world_that_is_not_gpus(tmp_path, product_lib):
    import subprocess

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device-module", "tests.stub_device", "--dist-backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
