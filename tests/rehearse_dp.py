"""Two-rank rehearsal of the data-parallel training path on ONE GPU: both ranks share cuda:0 and
exchange gradients through gloo (RCCL refuses two ranks per device).  Run by
tests/test_gpu_dp.py in a fresh interpreter (the parent of the two ranks never touches the GPU),
or by hand:
    python tests/rehearse_dp.py [resnet|hrnet|mobilenet] [rehearse|equiv|rccl1] [f32|bf16]

rehearse: after backward with the bucketed exchange every rank holds the SUM of the ranks' local
          gradients (each rank trains on its own images), and after three optimizer steps the
          ranks' parameters are identical.
equiv   : the reference's data parallelism (Horovod, test.ipynb:12) averages per-rank gradients.
          With batch-norm statistics frozen, 2 ranks x B/2 images must give the gradient of
          1 rank x B images: rank r takes half r of one global batch, the exchanged sum times
          1/2 (what the optimizer applies) is compared with a single-rank pass over the whole
          batch, <= 1e-5 in relative L2 over the arena.
rccl1   : ONE rank, torch.distributed backend "nccl" (= RCCL) with world size 1 and the exchange
          forced on: init_process_group("nccl", device_id=...), the asynchronous all_reduce work
          handles issued from the weight-gradient stream, their wait() before the optimizer and
          (bf16) the packed transport all run through RCCL; only N > 1 itself is left out.  A
          sum over one rank is the identity, so gradients and three steps' parameters must
          equal, bit for bit, those of the same network without an exchange (bf16 transport:
          gradients rounded once to bf16).
"""
import faulthandler
import os
import socket
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def log(rank, msg):
    print(f"[rank {rank} {time.strftime('%H:%M:%S')}] {msg}", flush=True)


def build(which, transport, alone=False):
    from pose_estimation_amd import model as pm

    stride, sigma, radius = 4.0, 2.0, 6
    if which == "hrnet":       # branched tape graph: layers finish out of arena order
        from pose_estimation_amd import hrnet

        net = hrnet.TrainablePoseHRNet("cuda:0", width=32, modules=(1, 1, 1), blocks=1,
                                       layer1_blocks=1, sigma=sigma)
        state = hrnet.init_state(net.spec, seed=3)
        bucket = 1 << 20
    elif which == "mobilenet":
        from pose_estimation_amd import mobilenet as mb

        net = mb.TrainablePoseMobileNetV2("cuda:0", setting=mb.MBV2_SETTING[:4])
        state = mb.init_state(net.blocks, 17, 0, seed=3)
        bucket = 1 << 18
        stride, sigma, radius = 8.0, 1.0, 1
    else:
        from pose_estimation_amd.training import TrainablePoseResNet

        net = TrainablePoseResNet("cuda:0", layers=(1, 1, 1, 1))
        state = pm.init_state(net.specs, seed=3)
        bucket = 4 << 20
    pm.randomize_bn(state, seed=4)     # running statistics that frozen batch norm can tell apart
    net.load_state(state)
    net.enable_data_parallel(bucket_bytes=bucket, transport=transport, exchange_when_alone=alone)
    return net, stride, sigma, radius


def worker(rank, world, port, which, mode, transport):
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pose_estimation_amd import ops, synthetic

    torch.cuda.set_device(0)
    net, stride, sigma, radius = build(which, transport)
    log(rank, f"{which} built, {len(net.bucketer.buckets)} buckets, {transport} transport, "
              f"backend {net.bucketer.backend}")
    H, W = 64, 64

    def backward(grad):
        if which == "resnet":
            net.backward(grad)
        else:
            net.backward({"final" if which == "hrnet" else "heat": grad})

    def fwd_bwd(img, joints):
        hm = net.forward_train(img)
        _, _, g, _ = ops.mse_fwd_bwd(hm, joints=joints, stride=stride, sigma=sigma, radius=radius)
        backward(g)
        if net.bucketer is not None:
            net.bucketer.finish()
        torch.cuda.synchronize()
        return net.grads.clone()

    def gather(t):
        t = t.cpu()        # gloo collectives on host tensors only
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return out

    if mode == "equiv":
        B = 8
        net.freeze_batch_norm(True)
        img_all = torch.from_numpy(synthetic.images_u8(B, H, W, seed=10)).cuda()
        j_all = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=20)).cuda()
        h = B // world
        summed = fwd_bwd(img_all[rank * h:(rank + 1) * h].contiguous(),
                         j_all[rank * h:(rank + 1) * h].contiguous())
        averaged = summed * net.bucketer.grad_scale
        bucketer, net.bucketer = net.bucketer, None
        full = fwd_bwd(img_all, j_all)           # one rank, the whole batch, no exchange
        net.bucketer = bucketer
        err = ((averaged - full).norm() / full.norm()).item()
        worst = ((averaged - full).abs().max() / full.abs().max()).item()
        log(rank, f"2 x {h} images vs 1 x {B}: rel L2 {err:.2e}, max {worst:.2e}")
        tol = 1e-5 if transport == "f32" else 1e-2
        assert full.abs().max().item() > 0 and err <= tol, (err, tol)
    else:
        B = 4
        img = torch.from_numpy(synthetic.images_u8(B, H, W, seed=10 + rank)).cuda()
        joints = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=20 + rank)).cuda()
        bucketer, net.bucketer = net.bucketer, None
        local = fwd_bwd(img, joints)             # this rank's own gradient, no exchange
        net.bucketer = bucketer
        got = fwd_bwd(img, joints)               # bucketed exchange behind backward
        want = torch.stack(gather(local)).sum(0)
        err = (got.cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-30)
        log(rank, f"summed gradient rel err {err:.2e}")
        assert err < (1e-5 if transport == "f32" else 1e-2), err
        for i in range(3):
            loss, _ = net.train_step(img, joints, lr=1e-3)
            log(rank, f"step {i} loss {loss.item():.5f}")
        masters = gather(net.master)
        assert torch.equal(masters[0], masters[1]), "ranks diverged"
        log(rank, "parameters identical on both ranks")
    net.bucketer.close()
    dist.barrier()
    dist.destroy_process_group()


def rccl_single_rank(which, transport):
    """See the module docstring, mode rccl1.  Runs in this process (no spawn): one rank."""
    from datetime import timedelta

    faulthandler.dump_traceback_later(200, exit=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev,
                            timeout=timedelta(seconds=120))
    from pose_estimation_amd import ops, synthetic

    net, stride, sigma, radius = build(which, transport, alone=True)
    bk = net.bucketer
    log(0, f"{which}: backend {bk.backend}, {len(bk.buckets)} buckets, {transport} transport")
    assert bk.backend == "nccl" and bk.world == 1 and len(bk.buckets) >= 2
    H = W = 64
    B = 4
    img = torch.from_numpy(synthetic.images_u8(B, H, W, seed=10)).cuda()
    joints = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=20)).cuda()

    def fwd_bwd():
        hm = net.forward_train(img)
        _, _, g, _ = ops.mse_fwd_bwd(hm, joints=joints, stride=stride, sigma=sigma, radius=radius)
        if which == "resnet":
            net.backward(g)
        else:
            net.backward({"final" if which == "hrnet" else "heat": g})
        if net.bucketer is not None:
            assert len(net.bucketer._pending) == len(net.bucketer.buckets), "buckets not issued"
            net.bucketer.finish()
        torch.cuda.synchronize()
        return net.grads.clone()

    net.bucketer = None
    local = fwd_bwd()
    net.bucketer = bk
    got = fwd_bwd()
    want = local if transport == "f32" else local.to(torch.bfloat16).float()
    assert torch.equal(got, want), "exchange over one RCCL rank changed the gradients"
    log(0, "gradients through RCCL equal the local ones")
    if transport == "f32":
        state0 = (net.master.clone(), net.opt_m.clone(), net.opt_v.clone(), net.work.clone())
        step0 = net.step_count
        with_x = []
        for _ in range(3):
            loss, _ = net.train_step(img, joints, lr=1e-3)
            with_x.append(loss.item())
        torch.cuda.synchronize()     # the side stream's weight re-pack of the last step is done
        after = net.master.clone()
        net.master.copy_(state0[0]); net.opt_m.copy_(state0[1]); net.opt_v.copy_(state0[2])
        net.work.copy_(state0[3]); net.step_count = step0
        net.refresh_transposes()
        net.bucketer = None
        without = []
        for _ in range(3):
            loss, _ = net.train_step(img, joints, lr=1e-3)
            without.append(loss.item())
        net.bucketer = bk
        log(0, f"losses with exchange {with_x} without {without}")
        # (running statistics were not rewound, and training-mode batch norm does not read them)
        assert with_x == without and torch.equal(after, net.master)
    bk.close()
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(f"rehearsal ok: {which} rccl1 {transport}", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "rccl1":
        rccl_single_rank(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 else "f32")
        return
    which = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("REH_MODEL", "resnet")
    mode = sys.argv[2] if len(sys.argv) > 2 else "rehearse"
    transport = sys.argv[3] if len(sys.argv) > 3 else "f32"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, which, mode, transport), nprocs=2, join=True)
    print(f"rehearsal ok: {which} {mode} {transport}", flush=True)


if __name__ == "__main__":
    main()
