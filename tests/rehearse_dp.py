"""Rehearsal of the data-parallel training path on ONE GPU: two ranks share cuda:0 and exchange
gradients through gloo (RCCL refuses two ranks per device).  Not collected by pytest; run as
    python tests/rehearse_dp.py            (REH_MODEL=hrnet | mobilenet for the tape-graph networks)
It checks that after a step both ranks hold identical parameters and that the averaged gradient
equals the mean of the two ranks' local gradients.
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


def worker(rank, world, port):
    faulthandler.dump_traceback_later(90, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log(rank, "process group up")
    from pose_estimation_amd import model as pm
    from pose_estimation_amd import synthetic
    from pose_estimation_amd.training import TrainablePoseResNet

    torch.cuda.set_device(0)
    which = os.environ.get("REH_MODEL", "resnet")
    stride, sigma, radius = 4.0, 2.0, 6
    if which == "hrnet":       # branched tape graph: layers finish out of arena order
        from pose_estimation_amd import hrnet

        net = hrnet.TrainablePoseHRNet("cuda:0", width=32, modules=(1, 1, 1), blocks=1,
                                       layer1_blocks=1, sigma=sigma)
        net.load_state(hrnet.init_state(net.spec, seed=3))
        net.enable_data_parallel(bucket_bytes=1 << 20)
    elif which == "mobilenet":
        from pose_estimation_amd import mobilenet as mb

        net = mb.TrainablePoseMobileNetV2("cuda:0", setting=mb.MBV2_SETTING[:4])
        net.load_state(mb.init_state(net.blocks, 17, 0, seed=3))
        net.enable_data_parallel(bucket_bytes=1 << 18)
        stride, sigma, radius = 8.0, 1.0, 1
    else:
        net = TrainablePoseResNet("cuda:0", layers=(1, 1, 1, 1))
        net.load_state(pm.init_state(net.specs, seed=3))
        net.enable_data_parallel(bucket_bytes=4 << 20)
    log(rank, f"{which} model built, {len(net.bucketer.buckets)} buckets")
    B, H, W = 4, 64, 64
    img = torch.from_numpy(synthetic.images_u8(B, H, W, seed=10 + rank)).cuda()
    joints = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=20 + rank)).cuda()

    from pose_estimation_amd import ops

    def backward(grad):
        if which == "resnet":
            net.backward(grad)
        else:
            net.backward({"final" if which == "hrnet" else "heat": grad})

    if os.environ.get("REH_STEPS_ONLY") == "1":
        for i in range(3):
            loss, _ = net.train_step(img, joints, lr=1e-3)
            log(rank, f"step {i} loss {loss.item():.5f}")
        dist.destroy_process_group()
        return

    # local gradient without exchange, for the check
    bucketer, net.bucketer = net.bucketer, None
    hm = net.forward_train(img)
    _, _, g, _ = ops.mse_fwd_bwd(hm, joints=joints, stride=stride, sigma=sigma, radius=radius)
    backward(g)
    local = net.grads.clone()
    net.bucketer = bucketer
    log(rank, "local gradient done")

    hm = net.forward_train(img)
    _, _, g, _ = ops.mse_fwd_bwd(hm, joints=joints, stride=stride, sigma=sigma, radius=radius)
    backward(g)
    log(rank, "backward with bucketed all-reduce issued")
    net.bucketer.finish()
    torch.cuda.synchronize()
    log(rank, "all-reduce finished")
    local = local.cpu()   # gloo collectives on host tensors only (see parallel.py)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = torch.stack(gathered).sum(0)
    err = (net.grads.cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-30)
    log(rank, f"summed gradient rel err {err:.2e}")
    assert err < 1e-5, err   # the all-reduce adds rank sums in its own order

    for i in range(3):
        loss, _ = net.train_step(img, joints, lr=1e-3)
        log(rank, f"step {i} loss {loss.item():.5f}")
    mine = net.master.cpu()
    masters = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(masters, mine)
    assert torch.equal(masters[0], masters[1]), "ranks diverged"
    log(rank, "parameters identical on both ranks")
    dist.destroy_process_group()


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
    print("rehearsal ok", flush=True)


if __name__ == "__main__":
    main()
