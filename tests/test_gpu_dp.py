"""The data-parallel training path with TWO ranks (the reference trains under Horovod,
test.ipynb:12).  A one-GPU box cannot hold two RCCL ranks, so both ranks share cuda:0 and the
collectives go through gloo; everything else -- buckets announced from the weight-gradient
stream as backward finishes them, out-of-order layer completion in the branched graphs, the
exchange running behind an event beside backward, finish() before the optimizer -- is the code
RCCL runs.  tests/rehearse_dp.py is started in a fresh interpreter whose parent process never
touches the GPU and spawns the two ranks itself."""
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _run(*args):
    r = subprocess.run([sys.executable, "-u", str(ROOT / "tests" / "rehearse_dp.py"), *args],
                       capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "rehearsal ok" in r.stdout, out[-4000:]
    return r.stdout


@pytest.mark.parametrize("which", ["resnet", "hrnet", "mobilenet"])
def test_two_ranks_sum_gradients_and_stay_identical(which):
    out = _run(which, "rehearse", "f32")
    assert out.count("parameters identical on both ranks") == 2


def test_two_ranks_with_bf16_gradient_transport():
    out = _run("resnet", "rehearse", "bf16")
    assert out.count("parameters identical on both ranks") == 2


@pytest.mark.parametrize("which", ["resnet", "hrnet"])
def test_two_half_batches_equal_one_full_batch_with_frozen_batch_norm(which):
    out = _run(which, "equiv", "f32")
    assert out.count("images vs 1 x 8") == 2


@pytest.mark.parametrize("which,transport", [("resnet", "f32"), ("resnet", "bf16"),
                                             ("hrnet", "f32")])
def test_torch_distributed_rccl_path_with_one_rank(which, transport):
    """torch.distributed's nccl backend (= RCCL) itself, world size 1, exchange forced on: the
    communicator is initialised with device_id, the asynchronous work handles are issued from
    the weight-gradient stream and waited for before the optimizer; results must equal the
    exchange-free run bit for bit.  What stays unexercised on a one-GPU box is N > 1 alone."""
    out = _run(which, "rccl1", transport)
    assert "gradients through RCCL equal the local ones" in out


def test_bucketer_over_the_native_communicator_single_rank():
    """GradBucketer driven by the library's own RCCL communicator (pe_allreduce_bucket on a
    stream of its own, ordered by events): with one rank the exchanged arena equals the local
    one, float32 exactly and bf16 transport to one rounding; the pre-multiplied scale of
    pe_allreduce_bucket is applied for both dtypes."""
    from pose_estimation_amd.parallel import GradBucketer, NativeComm

    torch.cuda.set_device(0)
    comm = NativeComm(0, 1, NativeComm.unique_id())
    g = torch.Generator().manual_seed(5)
    bounds = [0, 64, 640, 1000, 3000, 4096]
    for transport in ("f32", "bf16"):
        arena = torch.randn(5000, generator=g).cuda()
        want = arena.clone() if transport == "f32" else arena.to(torch.bfloat16).float()
        bk = GradBucketer(arena, bounds, bucket_bytes=4 * 700, comm=comm, transport=transport)
        assert bk.world == 1 and bk.backend == "native" and len(bk.buckets) >= 3
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):          # announced from another stream, as backward does
            side.wait_stream(torch.cuda.default_stream())
            for s in reversed(bounds):
                bk.layer_done(s)
        torch.cuda.current_stream().wait_stream(side)
        bk.finish()
        torch.cuda.synchronize()
        assert torch.equal(arena, want), transport
    a = torch.randn(4099, generator=g).cuda()
    b = a.to(torch.bfloat16)
    comm.all_reduce_(a.clone(), scale=1.0)
    wa = a * 0.25
    comm.all_reduce_(a, scale=0.25)
    wb = (b.float() * 0.25).to(torch.bfloat16)
    comm.all_reduce_(b, scale=0.25)
    torch.cuda.synchronize()
    assert torch.equal(a, wa) and torch.equal(b, wb)
    comm.close()


def _bench(*args, env=None):
    import os

    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-u", str(ROOT / "bench.py"), *args], env=e,
                          capture_output=True, text=True, timeout=280, cwd=str(ROOT))


def test_bench_launches_its_own_ranks_and_refuses_to_misreport():
    """`bench.py --gpus 2` started plainly spawns two ranks itself.  On a one-GPU box that is an
    error (exit status 2, no JSON line); with the rehearsal switch the two ranks share the GPU,
    run the data-parallel training step and report two ranks."""
    import json

    if torch.cuda.device_count() < 2:
        r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1")
        assert r.returncode == 2 and "only 1 GPU" in r.stderr and "{" not in r.stdout
    r = _bench("--gpus", "2", "--workload", "train", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", env={"PE_REHEARSE_ON_ONE_GPU": "1"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["collective_backend"] == "gloo"
    assert rec["config"]["workload"].startswith("resnet50_pose_train") and rec["value"] > 0
    # the line names the backend that really ran and explains its own exchange
    assert "(gloo)" in rec["config"]["sharding"] and "RCCL" not in rec["config"]["sharding"]
    assert rec["dp_check"]["params_identical_across_ranks"] is True
    ex = rec["exchange"]
    assert ex["backend"] == "gloo" and ex["world"] == 2 and ex["buckets"] >= 4
    assert ex["bytes_per_step"] == sum(ex["bucket_bytes"])
    assert abs(ex["bytes_per_step"] - 136.0e6) < 2e6          # 34.0 M float32 gradients
    assert ex["ms_per_step_without_exchange"] > 0 and "exchange_exposed_ms" in ex
    assert isinstance(ex["collective_env"], dict)
    assert 0 < rec["rank_ms_per_step_min"] <= rec["rank_ms_per_step_max"]
    # a launcher's world size that contradicts --gpus is refused too
    r = _bench("--gpus", "1", "--steps", "1", "--warmup", "0", "--no-also", "--no-cpu-baseline",
               env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
