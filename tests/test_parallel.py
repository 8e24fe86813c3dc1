"""Data-parallel gradient exchange on the CPU: bucket construction and a 2-rank gloo run of the
bucketed all-reduce (the same code path RCCL takes on the GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pose_estimation_amd.parallel import GradBucketer, make_buckets


def test_buckets_tile_the_arena_in_reverse_order():
    bounds = [0, 100, 250, 260, 900, 1000, 1800]
    total = 2000
    for size in (1, 150, 500, 5000):
        b = make_buckets(bounds, total, size)
        assert b[0][1] == total and b[-1][0] == 0
        for (s0, e0), (s1, e1) in zip(b, b[1:]):
            assert s0 == e1 and s1 < s0          # contiguous, descending
        assert all(s in bounds for s, _ in b)    # cut only at layer starts
        assert all((e - s) >= size or s == 0 for s, e in b)  # big enough unless it hits the start
    assert make_buckets(bounds, total, 1) == [(1800, 2000), (1000, 1800), (900, 1000), (260, 900),
                                              (250, 260), (100, 250), (0, 100)]
    assert make_buckets(bounds, total, 10**9) == [(0, 2000)]
    # a short tail: the bucket that finishes last is cut at the first layer start that leaves it
    # at least tail_elems long; everything still tiles the arena in descending order
    assert make_buckets(bounds, total, 10**9, 200) == [(250, 2000), (0, 250)]
    assert make_buckets(bounds, total, 500, 90) == [(1000, 2000), (260, 1000), (100, 260), (0, 100)]
    assert make_buckets(bounds, total, 500, 10**9) == make_buckets(bounds, total, 500)


def test_real_model_layout_buckets():
    from pose_estimation_amd import model as pm

    specs = pm.resnet50_pose_specs()
    convs = pm.all_convs(specs)
    # recompute the arena layout the way PoseResNet does, without a GPU
    off, starts = 0, []
    for c in convs:
        starts.append(off)
        n = 1
        for s in pm.packed_shape(c):
            n *= s
        off += (n + 7) // 8 * 8
        off += ((c.cout + 7) // 8 * 8) * (2 if c.bn else 1)
    b = make_buckets(starts, off, (32 << 20) // 4)
    assert sum(e - s for s, e in b) == off
    assert 3 <= len(b) <= 20
    assert 34.0e6 < off < 34.2e6
    # the all-reduce nothing can hide (stem and the first blocks) is about 4 MB instead of 32
    bt = make_buckets(starts, off, (32 << 20) // 4, (4 << 20) // 4)
    assert sum(e - s for s, e in bt) == off and len(bt) == len(b) + 1
    assert (4 << 20) // 4 <= bt[-1][1] - bt[-1][0] < (8 << 20) // 4 and bt[-1][0] == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        n = 5000
        arena = torch.randn(n)
        mine = arena.clone()
        bounds = [0, 64, 640, 1000, 3000, 4096]
        bk = GradBucketer(arena, bounds, bucket_bytes=4 * 700)
        assert bk.world == world and bk.grad_scale == 1.0 / world
        # backward order: mark layers from the last to the first
        for s in reversed(bounds):
            bk.layer_done(s)
        bk.finish()
        gathered = [torch.empty(n) for _ in range(world)]
        dist.all_gather(gathered, mine)
        want = torch.stack(gathered).sum(0)
        q.put((rank, torch.allclose(arena, want, rtol=0, atol=1e-6), len(bk.buckets)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] for r in results), "bucketed all-reduce differs from the sum over ranks"
    assert all(r[2] >= 3 for r in results)


@pytest.mark.gpu
def test_native_rccl_communicator_single_rank():
    """pe_comm_* over RCCL with one rank (a one-GPU box cannot hold two): the all-reduce is the
    identity and the folded scale is applied, for both transport dtypes."""
    import torch

    from pose_estimation_amd.parallel import NativeComm

    torch.cuda.set_device(0)
    uid = NativeComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = NativeComm(0, 1, uid)
    g = torch.Generator().manual_seed(3)
    a = torch.randn(100003, generator=g).cuda()
    want = a * 0.25
    comm.all_reduce_(a, scale=0.25)
    b = torch.randn(4096, generator=g).to(torch.bfloat16).cuda()
    wantb = b.clone()
    comm.all_reduce_(b)
    torch.cuda.synchronize()
    assert torch.equal(a, want) and torch.equal(b, wantb)
    comm.close()


def test_bench_refuses_more_ranks_than_gpus():
    """Without N visible GPUs `bench.py --gpus N` exits non-zero and prints no result line (here:
    no GPU at all, or one)."""
    import subprocess
    import sys
    from pathlib import Path

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("eight GPUs are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK",
                                                           "PE_REHEARSE_ON_ONE_GPU")}
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 8" in r.stderr and "{" not in r.stdout


def _identity_worker(rank, world, port, q):
    import importlib.util
    from pathlib import Path

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location(
            "bench", Path(__file__).resolve().parent.parent / "bench.py")
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        same = torch.linspace(-1.0, 1.0, 1000)
        a = bench._arena_identical_across_ranks(same, dist, True)
        differs = same.clone()
        differs[17] += 1e-3 * rank          # one element, one rank
        b = bench._arena_identical_across_ranks(differs, dist, True)
        broken = same.clone()
        broken[3] = float("nan")
        c = bench._arena_identical_across_ranks(broken, dist, True)
        q.put((rank, a, b, c))
    finally:
        dist.destroy_process_group()


def test_bench_cross_rank_parameter_check_two_ranks_gloo():
    """bench.py's check behind the data-parallel legs: identical arenas pass, one differing
    element on one rank or a NaN fails, on every rank alike."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_identity_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, a, b, c in results:
        assert a["params_identical_across_ranks"] is True and a["ranks"] == 2
        assert b["params_identical_across_ranks"] is False
        assert c["params_identical_across_ranks"] is False
