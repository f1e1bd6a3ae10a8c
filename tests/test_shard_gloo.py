"""Sharding over ranks: bounds arithmetic, and a world_size-2 run over gloo."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pygtsc import shard, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bounds_cover_and_balance():
    b = synth.skewed(total_points=1 << 16, seed=3)
    for world in (1, 2, 3, 8):
        bd = shard.shard_bounds(b.pt_off, world)
        assert bd[0] == 0 and bd[-1] == b.nseries and np.all(np.diff(bd) >= 0)
        pts = np.array([int(b.pt_off[bd[r + 1]]) - int(b.pt_off[bd[r]]) for r in range(world)])
        assert pts.sum() == b.npoints
        longest = int(np.diff(b.pt_off.astype(np.int64)).max())
        assert pts.max() <= b.npoints // world + longest + 1  # within one series of ideal


def test_bounds_uniform_and_degenerate():
    assert list(shard.shard_bounds(10, 4)) == [0, 2, 5, 7, 10]
    assert list(shard.shard_bounds(np.array([0], dtype=np.uint64), 2)) == [0, 0, 0]   # no series
    assert list(shard.shard_bounds(np.array([0, 5], dtype=np.uint64), 4))[-1] == 1      # fewer series than ranks
    # shards re-based to zero and contiguous
    b = synth.ragged_mix(seed=2, nseries=17, pmax=50)
    got_ts = np.concatenate([shard.shard_of(r, 3, b.block_ts, b.ts, b.val, b.pt_off)[1] for r in range(3)])
    assert np.array_equal(got_ts, b.ts)
    for r in range(3):
        off = shard.shard_of(r, 3, b.block_ts, b.ts, b.val, b.pt_off)[3]
        assert off[0] == 0 and len(off) >= 1


def test_concat_images():
    w, off = shard.concat_images([np.arange(5, dtype=np.uint64), np.arange(3, dtype=np.uint64)],
                                 [np.array([0, 2, 5], dtype=np.uint64), np.array([0, 3], dtype=np.uint64)])
    assert list(off) == [0, 2, 5, 8] and len(w) == 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(backend, tmp_path, world=2):
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "shard_worker.py"), "--backend", backend, "--out", str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res["ok"] and res["world"] == world
    return res


def test_world2_gloo_cpu(tmp_path):
    """Two ranks over gloo: shards coded separately and concatenated equal the whole batch."""
    _run_world("cpu", tmp_path)


@pytest.mark.gpu
def test_world2_gloo_gpu(tmp_path):
    """Same, each rank through the HIP library on the one GPU of the box (gloo for the control plane)."""
    _run_world("gpu", tmp_path)


@pytest.mark.gpu
def test_bench_command_with_two_ranks(tmp_path):
    """The multi-GPU benchmark command itself, as the driver launches it (torch.distributed.run, one rank per
    GPU), rehearsed with two ranks on the one GPU of the box: gloo carries the barrier and the MAX of the
    timing instead of RCCL, everything else is the code that runs on 8 GPUs."""
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--series", "20000", "--steps", "2", "--warmup", "1"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints, once
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["unit"] == "GB/s" and j["value"] > 0 and j["higher_is_better"] is True
    assert j["config"]["series_per_gpu"] == 20000 and j["config"]["points_per_series"] == 120
    # whole-job value: both ranks' points over the slower rank's time
    assert abs(j["value"] - 2 * 2 * 16.0 * 20000 * 120 * 2 / (j["ms_per_step"] * 2 * 1e-3) / 1e9) / j["value"] < 0.02
    for key in ("roofline", "raw_only_frac_of_hbm_peak", "compress_GBps_per_gpu", "decompress_GBps_per_gpu"):
        assert key in j
    assert "cpu_baseline" not in j and "other_workloads" not in j   # single-GPU extras only


@pytest.mark.gpu
def test_rccl_timing_reduction_with_one_rank():
    """bench.py's only communication is a barrier and a MAX all-reduce of the elapsed time over RCCL (backend "nccl").
    A one-GPU box cannot hold two RCCL ranks, so this runs the same calls with a world of one, in a child process:
    the process group comes up on the GPU, the reduction runs on a device tensor and returns the value."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = (
        "import sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from pygtsc import shard\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "dist.barrier()\n"
        "assert shard.max_over_ranks(1.25, dist, dev) == 1.25\n"
        "assert shard.sum_over_ranks(7.0, dist, dev) == 7.0\n"
        "dist.destroy_process_group()\n"
        "print('rccl ok')\n" % os.path.join(root, "gorilla-tsc_amd"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
