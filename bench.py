#!/usr/bin/env python3
"""Benchmark of the pose-estimation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload decode|render|mse|...]

One process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this
process is one rank.  Started plainly with --gpus N > 1 it launches the N ranks itself, as child
processes of a parent that never touches a GPU, and exits non-zero if fewer than N GPUs are
visible: the line printed always reports the number of ranks that really ran.
Every workload shards by batch (weak scaling: each rank processes its own batch of the stated
size); the training workloads exchange gradients (bucketed all-reduce over RCCL), the others
use no data-path collective.  Rank 0 prints ONE JSON line.

Default workload "decode" is BASELINE.json configs[1]: heat-map peak decode of
1024 x 17 x 64 x 48 float32 maps per step, inputs resident in HBM, rotating over enough
distinct input buffers (> 512 MB) that the 256 MB Infinity Cache cannot serve the reads.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import pose_estimation_amd as pe  # noqa: E402
from pose_estimation_amd import synthetic  # noqa: E402

METRIC = "images/sec @256×192,17kp (infer+train, 1/2/4/8 GPU); decode GB/s vs HBM peak"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, GB/s

K_JOINTS, H, W = 17, 64, 48
MAP_BYTES = H * W * 4


def host_cores() -> int:
    """Host threads the CPU baselines use: the box's CPU share (OMP_NUM_THREADS where the
    harness sets it; os.cpu_count() reports the whole machine there), else every core."""
    n = os.environ.get("OMP_NUM_THREADS")
    if n and n.isdigit() and int(n) > 0:
        return int(n)
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


class Workload:
    name = ""
    batch = 0
    dtype = "f32"
    bytes_per_step = 0      # algorithmic HBM bytes of one step (= one launch of `kernel`)
    kernel = ""
    bound = "hbm"

    def setup(self, device): ...
    def step(self, i: int): ...
    def cpu_baseline(self): ...


class Decode(Workload):
    name = "decode_1024x17x64x48_f32"
    batch = 1024
    dtype = "f32"
    kernel = "decode_contig_kernel<12>"
    # read every map once + 16 B of results per map
    bytes_per_step = batch * K_JOINTS * (MAP_BYTES + 16)
    nbuf = 4  # 4 x 214 MB = 856 MB of distinct inputs

    def setup(self, device):
        # the stated generator: per map one Gaussian blob (sigma 2, amplitude U(0.3, 1)) at a
        # uniformly random sub-pixel centre plus N(0, 0.01) noise, NumPy PCG64, one seed per buffer
        rank = int(os.environ.get("RANK", "0"))
        self.bufs = [torch.from_numpy(synthetic.decode_heatmaps(self.batch, K_JOINTS, H, W,
                                                                seed=100 * rank + i)).to(device)
                     for i in range(self.nbuf)]
        self.out = (torch.empty(self.batch, K_JOINTS, 2, device=device),
                    torch.empty(self.batch, K_JOINTS, device=device),
                    torch.empty(self.batch, K_JOINTS, dtype=torch.int32, device=device))

    def step(self, i):
        pe.decode(self.bufs[i % self.nbuf], refine=True, out=self.out)

    def cpu_baseline(self):
        from concurrent.futures import ThreadPoolExecutor

        import oracle

        cores = host_cores()
        hm = synthetic.decode_heatmaps(self.batch, K_JOINTS, H, W, seed=0)
        # the batch is cut into one slice per core (NumPy releases the GIL inside its loops)
        cuts = np.linspace(0, self.batch, cores + 1).astype(int)
        parts = [hm[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        times = []
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(oracle.decode_heatmaps, parts))
            t_end = time.perf_counter() + 10.0
            while len(times) < 5 or (time.perf_counter() < t_end and len(times) < 40):
                t0 = time.perf_counter()
                list(ex.map(oracle.decode_heatmaps, parts))
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        return {"value": self.batch / med, "unit": "images/s", "cores": cores, "kind": "port",
                "sample": f"NumPy oracle, full {self.batch}x{K_JOINTS}x{H}x{W} batch in "
                          f"{len(parts)} slices on {cores} threads, median of {len(times)} runs "
                          f"({med * 1e3:.1f} ms each)"}


class Render(Workload):
    name = "render_1024x17x64x48_f32"
    batch = 1024
    kernel = "render_contig_kernel"
    bytes_per_step = batch * K_JOINTS * (MAP_BYTES + 12 + 4)

    def setup(self, device):
        self.j = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 256, 192)).to(device)
        self.outs = [torch.empty(self.batch, K_JOINTS, H, W, device=device) for _ in range(4)]

    def step(self, i):
        pe.render_targets(self.j, H, W, 4.0, 2.0, 6, out=self.outs[i % 4])

    def cpu_baseline(self):
        import oracle

        j = synthetic.joints(64, K_JOINTS, 256, 192)
        t0 = time.perf_counter()
        oracle.render_targets(j, H, W, 4.0, 2.0, 6)
        dt = time.perf_counter() - t0
        return {"value": 64 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                "sample": "NumPy oracle, 64-image batch, one run"}


class Mse(Workload):
    fused = False
    batch = 1024
    kernel = "mse_map_kernel"

    def __init__(self):
        self.name = ("mse_fused" if self.fused else "mse") + "_1024x17x64x48_f32"
        per_map = (2 if self.fused else 3) * MAP_BYTES
        self.bytes_per_step = self.batch * K_JOINTS * per_map

    def setup(self, device):
        g = torch.Generator(device=device).manual_seed(7)
        self.j = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 256, 192)).to(device)
        self.preds = [torch.randn(self.batch, K_JOINTS, H, W, device=device, generator=g)
                      for _ in range(3)]
        self.t, self.w = pe.render_targets(self.j, H, W, 4.0, 2.0, 6)
        self.ws = pe.mse_workspace(self.batch, K_JOINTS, device)

    def step(self, i):
        if self.fused:
            pe.mse_fwd_bwd(self.preds[i % 3], joints=self.j, stride=4.0, sigma=2.0, radius=6,
                           workspace=self.ws)
        else:
            pe.mse_fwd_bwd(self.preds[i % 3], self.t, self.w, workspace=self.ws)

    def cpu_baseline(self):
        import oracle

        rng = np.random.default_rng(0)
        p = rng.normal(size=(64, K_JOINTS, H, W)).astype(np.float32)
        j = synthetic.joints(64, K_JOINTS, 256, 192)
        t, w = oracle.render_targets(j, H, W, 4.0, 2.0, 6)
        t0 = time.perf_counter()
        oracle.mse_fwd_bwd(p, t, w)
        dt = time.perf_counter() - t0
        return {"value": 64 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                "sample": "NumPy oracle, 64-image batch, one run"}


class MseFused(Mse):
    fused = True


VEC_PAIRS = [(5, 7), (7, 9), (6, 8), (8, 10), (11, 13), (13, 15), (12, 14), (14, 16), (5, 6),
             (11, 12), (5, 11), (6, 12), (0, 5), (0, 6), (1, 2), (3, 4)]


class VecRender(Workload):
    """Vector-map labels: 256 images x 16 keypoint pairs = 64 channels of 64x48 per image."""
    name = "vecmap_render_256x64x64x48_f32"
    batch = 256
    kernel = "vec_render_contig_kernel"
    bytes_per_step = batch * 4 * len(VEC_PAIRS) * MAP_BYTES     # every map written in full

    def setup(self, device):
        from pose_estimation_amd import ops

        self.ops = ops
        self.j = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 256, 192)).to(device)
        self.outs = [torch.empty(self.batch, 4 * len(VEC_PAIRS), H, W, device=device)
                     for _ in range(4)]

    def step(self, i):
        self.ops.render_vecmaps(self.j, VEC_PAIRS, H, W, 4.0, 2, out=self.outs[i % 4])

    def cpu_baseline(self):
        import oracle

        j = synthetic.joints(16, K_JOINTS, 256, 192)
        t0 = time.perf_counter()
        oracle.render_vecmaps(j, VEC_PAIRS, H, W, 4.0, 2)
        dt = time.perf_counter() - t0
        return {"value": 16 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                "sample": "NumPy oracle, 16-image batch, one run"}


class VecLoss(VecRender):
    """Vector-map loss + gradient: the label is rebuilt in registers; pred is read only near
    the windows, the gradient is written in full."""
    name = "vecmap_loss_256x64x64x48_f32"
    kernel = "vec_loss_kernel"
    bytes_per_step = 256 * 4 * len(VEC_PAIRS) * MAP_BYTES       # the gradient write

    def setup(self, device):
        super().setup(device)
        g = torch.Generator(device=device).manual_seed(7)
        self.preds = [torch.randn(self.batch, 4 * len(VEC_PAIRS), H, W, device=device, generator=g)
                      for _ in range(3)]
        self.ws = torch.empty(1 << 20, dtype=torch.uint8, device=device)

    def step(self, i):
        self.ops.vecmap_loss_fwd_bwd(self.preds[i % 3], self.j, VEC_PAIRS, 4.0, 2, workspace=self.ws)

    def cpu_baseline(self):
        import oracle

        j = synthetic.joints(16, K_JOINTS, 256, 192)
        p = np.random.default_rng(0).normal(size=(16, 4 * len(VEC_PAIRS), H, W)).astype(np.float32)
        t0 = time.perf_counter()
        oracle.vecmap_loss_fwd_bwd(p, j, VEC_PAIRS, 4.0, 2)
        dt = time.perf_counter() - t0
        return {"value": 16 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                "sample": "NumPy oracle, 16-image batch, one run"}


class Augment(Workload):
    """Input pipeline: 640x480 uint8 frames -> scale jitter / crop / flip / normalise -> the
    320x320 bf16 space-to-depth image, plus the joint transform (the reference's configured
    augmentation, test.ipynb:995-1010)."""
    name = "augment_640x480_to_320x320_b256"
    batch = 256
    dtype = "f32"
    kernel = "augment_s2d_kernel"
    # reads the 640x480x3 frame at most once, writes 320x320 pixels x 8 bytes (16 bf16 per 2x2)
    bytes_per_step = batch * (480 * 640 * 3 + 320 * 320 * 8)

    def setup(self, device):
        from pose_estimation_amd import augment

        self.aug = augment
        self.imgs = [torch.from_numpy(synthetic.images_u8(self.batch, 480, 640, seed=1 + i)).to(device)
                     for i in range(3)]
        self.prm = torch.from_numpy(augment.random_params(self.batch, (480, 640), (320, 320),
                                                          seed=3)).to(device)
        self.j = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 480, 640)).to(device)
        self.out = torch.empty(self.batch, 160, 160, 16, dtype=torch.bfloat16, device=device)

    def step(self, i):
        self.aug.augment_u8(self.imgs[i % 3], self.prm, (320, 320), out=self.out)
        self.aug.transform_joints(self.j, self.prm, 320, self.aug.COCO_FLIP_PERM)

    def cpu_baseline(self):
        import oracle
        from pose_estimation_amd import augment
        from pose_estimation_amd import nn as pnn

        img = synthetic.images_u8(8, 480, 640, seed=1)
        prm = augment.random_params(8, (480, 640), (320, 320), seed=3)
        t0 = time.perf_counter()
        oracle.augment_u8(img, prm, (320, 320), pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
        dt = time.perf_counter() - t0
        return {"value": 8 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                "sample": "NumPy oracle, 8-image batch, one run"}


class Adam(Workload):
    name = "adam_34M_f32"
    batch = 1  # one optimizer step
    kernel = "adam_kernel"
    n = 34_000_000
    bytes_per_step = n * 30

    def setup(self, device):
        self.p = torch.randn(self.n, device=device)
        self.m = torch.zeros(self.n, device=device)
        self.v = torch.zeros(self.n, device=device)
        self.g = torch.randn(self.n, device=device)
        self.pb = torch.empty(self.n, dtype=torch.bfloat16, device=device)

    def step(self, i):
        pe.adam_step(self.p, self.m, self.v, self.g, i + 1, p_bf16=self.pb)

    def cpu_baseline(self):
        import oracle

        n = 4_000_000
        z = np.zeros(n, np.float32)
        g = np.ones(n, np.float32)
        t0 = time.perf_counter()
        oracle.adam_step(z, z, z, g, 1e-3, 0.9, 0.999, 1e-8, 1)
        dt = time.perf_counter() - t0
        return {"value": (n / self.n) / dt, "unit": "steps/s", "cores": 1, "kind": "port",
                "sample": "NumPy oracle, 4M-parameter slice, scaled to 34M"}


class Infer(Workload):
    """BASELINE configs[2]: ResNet-50 + 3-deconv inference, 256x192, batch 256, bf16, with the
    uint8 preprocessing in front and the heat-map decode behind (one step = images -> keypoints)."""
    name = "resnet50_pose_infer_256x192_b256_bf16"
    batch = 256
    dtype = "bf16"
    bound = "mfma"
    kernel = "igemm_kernel + stem_pool_kernel + conv3x3_c64_kernel (all conv/deconv launches of one forward)"
    peak = 2500.0  # dense bf16 MFMA, TFLOP/s

    def setup(self, device):
        from pose_estimation_amd import flops, model

        self.net = model.PoseResNet(device)
        self.net.load_state(model.init_state(self.net.specs, seed=3))
        macs, _, _ = flops.forward_macs(256, 192)
        self.flops_per_step = 2.0 * macs * self.batch
        self.imgs = [torch.from_numpy(synthetic.images_u8(self.batch, 256, 192, seed=1 + i)).to(device)
                     for i in range(2)]

    def step(self, i):
        hm = self.net.predict(self.imgs[i % 2])
        pe.decode(hm)

    def roofline(self, kern_ms):
        ach = self.flops_per_step / (kern_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": self.kernel, "achieved": ach, "peak": self.peak,
                "unit": "TFLOP/s", "frac": ach / self.peak,
                "flops_per_step": self.flops_per_step, "step_ms": kern_ms, "traffic": None}

    def cpu_baseline(self):
        from oracle import torch_oracle as to
        from pose_estimation_amd import model
        from pose_estimation_amd import nn as pnn

        specs = model.resnet50_pose_specs()
        state = model.init_state(specs, seed=3)
        img = synthetic.images_u8(8, 256, 192, seed=1)
        import oracle

        def once():
            hm = to.pose_resnet_forward(specs, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
            oracle.decode_heatmaps(np.asarray(hm, dtype=np.float32))

        to.pose_resnet_forward(specs, state, img[:1], pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
        times = []
        t_end = time.perf_counter() + 10.0
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 100):
            t0 = time.perf_counter()
            once()
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        return {"value": 8 / med, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU float32 oracle forward + NumPy decode, batch 8, median of "
                          f"{len(times)} runs ({med:.2f} s each)"}


class MobileNetInfer(Workload):
    """The reference's own backbone family and resolution: MobileNetV2 (output stride 8) +
    heat-map head, 320x320 images -> 40x40 maps, batch 256, bf16, preprocessing and decode
    included.  Mostly bandwidth-bound (depthwise 3x3 and thin pointwise layers)."""
    name = "mobilenetv2_pose_infer_320x320_b256_bf16"
    batch = 256
    heatmap = (40, 40)
    dtype = "bf16"
    bound = "mfma"
    kernel = "igemm_kernel + depthwise3x3_kernel (all launches of one forward)"
    peak = 2500.0

    def setup(self, device):
        from pose_estimation_amd import mobilenet as mb

        self.net = mb.PoseMobileNetV2(device)
        self.net.load_state(mb.init_state(self.net.blocks, seed=3))
        self.flops_per_step = 2.0 * mb.forward_macs(320, 320, K_JOINTS) * self.batch
        self.imgs = [torch.from_numpy(synthetic.images_u8(self.batch, 320, 320, seed=1 + i)).to(device)
                     for i in range(2)]

    def step(self, i):
        pe.decode(self.net.predict(self.imgs[i % 2]))

    def roofline(self, kern_ms):
        ach = self.flops_per_step / (kern_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": self.kernel, "achieved": ach, "peak": self.peak,
                "unit": "TFLOP/s", "frac": ach / self.peak,
                "flops_per_step": self.flops_per_step, "step_ms": kern_ms, "traffic": None}

    def cpu_baseline(self):
        from oracle import torch_oracle as to
        from pose_estimation_amd import mobilenet as mb
        from pose_estimation_amd import nn as pnn

        blocks = mb.mobilenet_v2_specs()
        state = mb.init_state(blocks, seed=3)
        img = synthetic.images_u8(8, 320, 320, seed=1)
        to.mobilenet_pose_forward(blocks, state, img[:1], pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            to.mobilenet_pose_forward(blocks, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        return {"value": 8 / med, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU float32 oracle forward, batch 8, median of 3 ({med:.2f} s)"}


class HRNetInfer(Workload):
    """BASELINE configs[4] (forward part): HRNet-W32-style multi-branch backbone at 384x288 ->
    96x72 heat maps, batch 64, bf16, preprocessing and decode included."""
    name = "hrnet_w32_pose_infer_384x288_b64_bf16"
    batch = 64
    heatmap = (96, 72)
    dtype = "bf16"
    bound = "mfma"
    kernel = "igemm_kernel + conv3x3_c32_kernel + conv3x3_c64_kernel + fuse_upsample_add_kernel (all launches of one forward)"
    peak = 2500.0

    def setup(self, device):
        from pose_estimation_amd import hrnet

        self.net = hrnet.PoseHRNet(device)
        self.net.load_state(hrnet.init_state(self.net.spec, seed=3))
        self.flops_per_step = 2.0 * hrnet.forward_macs(self.net.spec, 384, 288) * self.batch
        self.imgs = [torch.from_numpy(synthetic.images_u8(self.batch, 384, 288, seed=1 + i)).to(device)
                     for i in range(2)]

    def step(self, i):
        pe.decode(self.net.predict(self.imgs[i % 2]))

    def roofline(self, kern_ms):
        ach = self.flops_per_step / (kern_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": self.kernel, "achieved": ach, "peak": self.peak,
                "unit": "TFLOP/s", "frac": ach / self.peak,
                "flops_per_step": self.flops_per_step, "step_ms": kern_ms, "traffic": None}

    def cpu_baseline(self):
        from oracle import torch_oracle as to
        from pose_estimation_amd import hrnet
        from pose_estimation_amd import nn as pnn

        spec = hrnet.hrnet_specs()
        state = hrnet.init_state(spec, seed=3)
        img = synthetic.images_u8(4, 384, 288, seed=1)
        to.hrnet_pose_forward(spec, state, img[:1], pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
        t0 = time.perf_counter()
        to.hrnet_pose_forward(spec, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
        dt = time.perf_counter() - t0
        return {"value": 4 / dt, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU float32 oracle forward, batch 4, one run ({dt:.2f} s)"}


class Train(Workload):
    """BASELINE configs[3]: full training step (forward with batch statistics, fused target
    render + masked MSE, backward, gradient all-reduce when N > 1, fused Adam), batch 64 per GPU."""
    name = "resnet50_pose_train_256x192_b64_bf16"
    batch = 64
    dtype = "bf16"
    bound = "mfma"
    kernel = "igemm_kernel + wgrad_kernel (all launches of one step)"
    peak = 2500.0
    sharding = "data parallel: batch per rank; one rank, so no gradient exchange"

    def setup(self, device):
        import torch.distributed as dist

        from pose_estimation_amd import flops, model
        from pose_estimation_amd.training import TrainablePoseResNet

        self.net = TrainablePoseResNet(device)
        self.net.load_state(model.init_state(self.net.specs, seed=3))
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            # PE_DP_TRANSPORT=bf16: gradients cross the links as bf16 (half the bytes); the
            # default keeps them float32, as the reference's Horovod exchange does
            self.transport = os.environ.get("PE_DP_TRANSPORT", "f32")
            self.net.enable_data_parallel(32 << 20, transport=self.transport)
            self.sharding = (f"data parallel: batch per rank, bucketed {self.transport} gradient "
                             f"all-reduce ({backend_name(dist)}) behind backward")
        self.host_replay = not multi and os.environ.get("PE_HOST_REPLAY", "0") == "1"
        macs, _, _ = flops.forward_macs(256, 192)
        self.flops_per_step = 3 * 2.0 * macs * self.batch   # backward = 2 x forward
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.imgs = torch.from_numpy(synthetic.images_u8(self.batch, 256, 192, seed=1 + rank)).to(device)
        self.joints = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 256, 192,
                                                        seed=2 + rank)).to(device)
        self.mask = torch.from_numpy(synthetic.ignore_mask(self.batch, H, W, seed=4 + rank)).to(device)

    def step(self, i):
        self.net.train_step(self.imgs, self.joints, self.mask, lr=1e-3,
                            host_replay=self.host_replay)

    def dp_check(self, dist, on_host):
        """After the timed steps of an N > 1 run: every rank trained on its own images, so the
        ranks hold identical parameters only if every bucket of every step was summed over all
        of them.  Two checksums of the float32 master arena, MAX and MIN over the ranks."""
        return _arena_identical_across_ranks(self.net.master, dist, on_host)

    def roofline(self, kern_ms):
        ach = self.flops_per_step / (kern_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": self.kernel, "achieved": ach, "peak": self.peak,
                "unit": "TFLOP/s", "frac": ach / self.peak,
                "flops_per_step": self.flops_per_step, "step_ms": kern_ms, "traffic": None}

    def cpu_baseline(self):
        import oracle
        from oracle import torch_oracle as to
        from pose_estimation_amd import model
        from pose_estimation_amd import nn as pnn

        specs = model.resnet50_pose_specs()
        state = model.init_state(specs, seed=3)
        B = 8
        img = synthetic.images_u8(B, 256, 192, seed=1)
        j = synthetic.joints(B, K_JOINTS, 256, 192, seed=2)
        m = synthetic.ignore_mask(B, H, W, seed=4)
        times = []
        t_end = time.perf_counter() + 12.0
        while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 50):
            t0 = time.perf_counter()
            t, w = oracle.render_targets(j, H, W, 4.0, 2.0, 6)
            to.pose_resnet_train_reference(specs, state, img, t, w, m, pnn.IMAGENET_MEAN,
                                           pnn.IMAGENET_STD)
            times.append(time.perf_counter() - t0)
        dt = float(np.median(times))
        return {"value": B / dt, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU autograd oracle (render + loss + backward, no optimizer), "
                          f"batch {B}, median of {len(times)} runs ({dt:.2f} s each)"}


class HRNetTrain(Workload):
    """BASELINE configs[4]: full training step of the HRNet-W32-style multi-branch network at
    384x288 -> 96x72 heat maps (forward with batch statistics, fused target render + masked MSE,
    backward, gradient all-reduce when N > 1, fused Adam), batch 64 per GPU."""
    name = "hrnet_w32_pose_train_384x288_b64_bf16"
    batch = 64
    heatmap = (96, 72)
    dtype = "bf16"
    bound = "mfma"
    kernel = "igemm_kernel + wgrad_kernel (all launches of one step)"
    peak = 2500.0
    sharding = "data parallel: batch per rank; one rank, so no gradient exchange"

    def setup(self, device):
        import torch.distributed as dist

        from pose_estimation_amd import hrnet

        self.net = hrnet.TrainablePoseHRNet(device)
        self.net.load_state(hrnet.init_state(self.net.spec, seed=3))
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            self.net.enable_data_parallel(32 << 20)
            self.sharding = ("data parallel: batch per rank, bucketed f32 gradient all-reduce "
                             f"({backend_name(dist)}) behind backward")
        # PE_HIP_GRAPH=1 (one GPU only): replay the forward/loss/backward pass as a captured HIP
        # graph.  Measured slower than eager launches on this stack (44.5 vs 39.3 ms), so off.
        self.hip_graph = not multi and os.environ.get("PE_HIP_GRAPH") == "1"
        # PE_HOST_REPLAY (one GPU only): enqueue the pass from the call list recorded on the
        # first step instead of walking the Python graph
        self.host_replay = not multi and os.environ.get("PE_HOST_REPLAY", "0") == "1"
        self.flops_per_step = 3 * 2.0 * hrnet.forward_macs(self.net.spec, 384, 288) * self.batch
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.imgs = torch.from_numpy(synthetic.images_u8(self.batch, 384, 288, seed=1 + rank)).to(device)
        self.joints = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 384, 288,
                                                        seed=2 + rank)).to(device)
        self.mask = torch.from_numpy(synthetic.ignore_mask(self.batch, 96, 72, seed=4 + rank)).to(device)

    def step(self, i):
        self.net.train_step(self.imgs, self.joints, self.mask, lr=1e-3, hip_graph=self.hip_graph,
                            host_replay=self.host_replay)

    def roofline(self, kern_ms):
        ach = self.flops_per_step / (kern_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": self.kernel, "achieved": ach, "peak": self.peak,
                "unit": "TFLOP/s", "frac": ach / self.peak,
                "flops_per_step": self.flops_per_step, "step_ms": kern_ms, "traffic": None}

    def dp_check(self, dist, on_host):
        return _arena_identical_across_ranks(self.net.master, dist, on_host)

    def cpu_baseline(self):
        import oracle
        from oracle import torch_oracle as to
        from pose_estimation_amd import hrnet
        from pose_estimation_amd import nn as pnn

        spec = hrnet.hrnet_specs()
        state = hrnet.init_state(spec, seed=3)
        B = 2
        img = synthetic.images_u8(B, 384, 288, seed=1)
        j = synthetic.joints(B, K_JOINTS, 384, 288, seed=2)
        m = synthetic.ignore_mask(B, 96, 72, seed=4)
        t0 = time.perf_counter()
        t, w = oracle.render_targets(j, 96, 72, 4.0, 3.0, 9)
        to.hrnet_train_reference(spec, state, img, t, w, m, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
        dt = time.perf_counter() - t0
        return {"value": B / dt, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU autograd oracle (render + loss + backward, no optimizer), "
                          f"batch {B}, one run ({dt:.1f} s)"}


class MobileNetTrain(Workload):
    """The reference's own network family and label layout: MobileNetV2 (output stride 8) with
    heat-map and vector-map heads (19 keypoint pairs), 320x320 images -> 40x40 maps, full
    training step (forward with batch statistics, both label renders fused into their losses,
    backward, gradient all-reduce when N > 1, fused Adam), batch 64 per GPU."""
    name = "mobilenetv2_pose_train_320x320_b64_bf16"
    batch = 64
    heatmap = (40, 40)
    dtype = "bf16"
    bound = "hbm"
    kernel = "all launches of one step (depthwise, batch-norm and thin pointwise layers)"
    peak = 2500.0
    sharding = "data parallel: batch per rank; one rank, so no gradient exchange"

    def setup(self, device):
        import torch.distributed as dist

        from pose_estimation_amd import mobilenet as mb

        self.net = mb.TrainablePoseMobileNetV2(device, pairs=mb.COCO_PAIRS)
        self.net.load_state(mb.init_state(self.net.blocks, K_JOINTS, len(mb.COCO_PAIRS), seed=3))
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            self.net.enable_data_parallel(32 << 20)
            self.sharding = ("data parallel: batch per rank, bucketed f32 gradient all-reduce "
                             f"({backend_name(dist)}) behind backward")
        self.hip_graph = not multi and os.environ.get("PE_HIP_GRAPH") == "1"
        # PE_HOST_REPLAY (one GPU only): enqueue the pass from the call list recorded on the
        # first step instead of walking the Python graph
        self.host_replay = not multi and os.environ.get("PE_HOST_REPLAY", "0") == "1"
        self.flops_per_step = 3 * 2.0 * mb.forward_macs(320, 320, K_JOINTS, len(mb.COCO_PAIRS)) \
            * self.batch
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.imgs = torch.from_numpy(synthetic.images_u8(self.batch, 320, 320, seed=1 + rank)).to(device)
        self.joints = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 320, 320,
                                                        seed=2 + rank)).to(device)

    def step(self, i):
        self.net.train_step(self.imgs, self.joints, None, lr=1e-3, hip_graph=self.hip_graph,
                            host_replay=self.host_replay)

    def roofline(self, kern_ms):
        ach = self.flops_per_step / (kern_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": self.kernel, "achieved": ach, "peak": self.peak,
                "unit": "TFLOP/s", "frac": ach / self.peak,
                "flops_per_step": self.flops_per_step, "step_ms": kern_ms, "traffic": None}

    def dp_check(self, dist, on_host):
        return _arena_identical_across_ranks(self.net.master, dist, on_host)

    def cpu_baseline(self):
        from oracle import torch_oracle as to
        from pose_estimation_amd import mobilenet as mb
        from pose_estimation_amd import nn as pnn

        blocks = mb.mobilenet_v2_specs()
        state = mb.init_state(blocks, K_JOINTS, len(mb.COCO_PAIRS), seed=3)
        B = 4
        img = synthetic.images_u8(B, 320, 320, seed=1)
        dh = np.zeros((B, K_JOINTS, 40, 40), np.float32)
        dv = np.zeros((B, 4 * len(mb.COCO_PAIRS), 40, 40), np.float32)
        t0 = time.perf_counter()
        to.mobilenet_train_reference(blocks, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, dh, dv)
        dt = time.perf_counter() - t0
        return {"value": B / dt, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU autograd oracle (forward + backward from given head "
                          f"gradients, no loss, no optimizer), batch {B}, one run ({dt:.1f} s)"}


class MobileNetDetTrain(MobileNetTrain):
    """MobileNetTrain plus the reference's third label group (`labels['bboxes']`): two stride-2
    detection levels with box heads over 1500 anchors, labels assigned on the GPU from one
    person box per image, objectness + box loss beside the two dense losses."""
    name = "mobilenetv2_pose_det_train_320x320_b64_bf16"

    def setup(self, device):
        import torch.distributed as dist

        from pose_estimation_amd import detection
        from pose_estimation_amd import mobilenet as mb

        self.net = mb.TrainablePoseMobileNetV2(device, pairs=mb.COCO_PAIRS, detection=True)
        self.net.load_state(mb.init_state(self.net.blocks, K_JOINTS, len(mb.COCO_PAIRS), seed=3,
                                          det_blocks=self.net.det_blocks))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.net.enable_data_parallel(32 << 20)
            self.sharding = ("data parallel: batch per rank, bucketed f32 gradient all-reduce "
                             f"({backend_name(dist)}) behind backward")
        macs = mb.forward_macs(320, 320, K_JOINTS, len(mb.COCO_PAIRS)) + mb.detection_macs()
        self.flops_per_step = 3 * 2.0 * macs * self.batch
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.imgs = torch.from_numpy(synthetic.images_u8(self.batch, 320, 320, seed=1 + rank)).to(device)
        self.joints = torch.from_numpy(synthetic.joints(self.batch, K_JOINTS, 320, 320,
                                                        seed=2 + rank)).to(device)
        self.anchors = torch.from_numpy(detection.make_anchors()).to(device)
        self.detection = detection

    def step(self, i):
        # label assignment is part of the step: boxes -> per-anchor targets on the GPU
        boxes, cls = self.detection.boxes_from_joints(self.joints, 320, 320)
        classes, weights, regs, _ = self.detection.anchor_assign(self.anchors, boxes, cls)
        self.net.train_step(self.imgs, self.joints, None, lr=1e-3,
                            det_labels=(classes, weights, regs))

    def dp_check(self, dist, on_host):
        return _arena_identical_across_ranks(self.net.master, dist, on_host)

    def cpu_baseline(self):
        from oracle import torch_oracle as to
        from pose_estimation_amd import mobilenet as mb
        from pose_estimation_amd import nn as pnn

        blocks = mb.mobilenet_v2_specs()
        det = mb.detection_specs(blocks[-1].cout)
        state = mb.init_state(blocks, K_JOINTS, len(mb.COCO_PAIRS), seed=3, det_blocks=det)
        B = 4
        img = synthetic.images_u8(B, 320, 320, seed=1)
        dh = np.zeros((B, K_JOINTS, 40, 40), np.float32)
        dv = np.zeros((B, 4 * len(mb.COCO_PAIRS), 40, 40), np.float32)
        db = [torch.zeros(B, 15, 20, 20), torch.zeros(B, 15, 10, 10)]
        t0 = time.perf_counter()
        to.mobilenet_train_reference(blocks, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, dh, dv,
                                     det_blocks=det, dboxes=db)
        dt = time.perf_counter() - t0
        return {"value": B / dt, "unit": "images/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"torch-CPU autograd oracle (forward + backward from given head "
                          f"gradients, no loss, no optimizer), batch {B}, one run ({dt:.1f} s)"}


def collective_env() -> dict:
    """The environment a collective library reads, as this rank sees it."""
    keys = ("NCCL_", "RCCL_", "HSA_", "GPU_MAX_HW_QUEUES", "TORCH_NCCL_", "PE_DP_", "HIP_VISIBLE",
            "ROCR_VISIBLE", "CUDA_VISIBLE")
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(keys)}


def backend_name(dist) -> str:
    """'RCCL' for torch.distributed's nccl backend on ROCm, else the backend's own name."""
    if dist is None or not dist.is_initialized():
        return "none"
    b = dist.get_backend()
    return "RCCL" if b == "nccl" else str(b)


def _arena_identical_across_ranks(arena, dist, on_host):
    a = arena.detach().double()
    chk = torch.stack([a.sum(), a.abs().sum(), (a * a).sum()])
    if on_host:
        chk = chk.cpu()
    hi, lo = chk.clone(), chk.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    finite = bool(torch.isfinite(hi).all())
    return {"params_identical_across_ranks": finite and bool((hi == lo).all()),
            "ranks": dist.get_world_size(), "checksum": float(hi[1])}


WORKLOADS = {"decode": Decode, "render": Render, "mse": Mse, "mse_fused": MseFused, "adam": Adam,
             "vecmap": VecRender, "vecmap_loss": VecLoss, "mobilenet": MobileNetInfer,
             "augment": Augment, "hrnet": HRNetInfer,
             "infer": Infer, "train": Train, "hrnet_train": HRNetTrain,
             "mobilenet_train": MobileNetTrain, "mobilenet_det_train": MobileNetDetTrain}


def launch_ranks(args) -> int:
    """--gpus N > 1 without a launcher: start the N ranks as child processes.  This parent makes
    no GPU call (torch.cuda.device_count() does not initialise the GPU), so the children are
    fresh processes and nothing that has touched a GPU is ever re-executed."""
    import socket
    import subprocess

    rehearse = os.environ.get("PE_REHEARSE_ON_ONE_GPU") == "1"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not rehearse:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible "
              "(PE_REHEARSE_ON_ONE_GPU=1 rehearses N ranks on one GPU over gloo)", file=sys.stderr)
        return 2
    # The rendezvous port: bound here and handed to rank 0 still free but for the instant
    # between close and re-bind; a second try with a fresh port covers the case that somebody
    # took it in between (rank 0 then fails to bind and every rank exits within the init
    # timeout).
    rc = 0
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()),
                                           *sys.argv[1:]], env=env))
        rc = _wait_ranks(procs)
        if rc != EXIT_RENDEZVOUS:
            break
    return rc


EXIT_RENDEZVOUS = 5     # a rank could not join the process group (nothing was measured)


def _wait_ranks(procs, grace: float = 10.0) -> int:
    """Waits for every rank.  As soon as one exits non-zero the others are told to stop
    (SIGTERM, SIGKILL after `grace` seconds): ranks that lost a peer would otherwise sit in the
    rendezvous or in a collective until its timeout.  Returns the first failing status, else 0.
    Only the exact child processes started here are signalled."""
    rc = 0
    live = list(procs)
    while live:
        for pr in list(live):
            st = pr.poll()
            if st is None:
                continue
            live.remove(pr)
            if st != 0 and rc == 0:
                rc = abs(st)
                for other in live:
                    other.terminate()
                t_end = time.monotonic() + grace
                while any(o.poll() is None for o in live) and time.monotonic() < t_end:
                    time.sleep(0.1)
                for other in live:
                    if other.poll() is None:
                        other.kill()
        time.sleep(0.05)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="decode", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="default workload only: skip the inference and training legs")
    ap.add_argument("--also-timeout", type=float, default=240.0)
    ap.add_argument("--init-timeout", type=float, default=180.0,
                    help="seconds a rank waits for the others in the rendezvous")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would misreport "
                         "the number of GPUs")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    # PE_REHEARSE_ON_ONE_GPU=1: every rank uses cuda:0 and the collectives go through gloo, so
    # the N > 1 code path can be rehearsed on a one-GPU box (RCCL refuses two ranks per device).
    rehearse = os.environ.get("PE_REHEARSE_ON_ONE_GPU") == "1"
    if not rehearse and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU "
                         f"({torch.cuda.device_count()} visible)")
    device = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist

        from datetime import timedelta

        backend = "gloo" if rehearse else "nccl"
        step = f"init_process_group({backend!r})"
        try:
            if rehearse:
                dist.init_process_group("gloo", timeout=timedelta(seconds=args.init_timeout))
            else:
                dist.init_process_group("nccl", device_id=device,
                                        timeout=timedelta(seconds=args.init_timeout))
            # the first collective builds the communicator (rings, channels, peer mappings):
            # done here, by name, so that a failure is not mistaken for one of the workload's
            step = f"first all_reduce over {backend}"
            probe = torch.ones(1, device="cpu" if rehearse else device)
            dist.all_reduce(probe)
            if not rehearse:
                torch.cuda.synchronize(device)
            if int(probe.item()) != world:
                raise RuntimeError(f"sum of ones over {world} ranks is {probe.item()}")
        except Exception as exc:  # noqa: BLE001 - reported as the line, then a non-zero exit
            msg = f"{step} failed on rank {rank}: {type(exc).__name__}: {exc}"[:600]
            print(f"bench.py: {msg}", file=sys.stderr, flush=True)
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "n_gpus": world,
                                  "error": msg, "collective_env": collective_env()}), flush=True)
            os._exit(EXIT_RENDEZVOUS)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    progress = {"leg": args.workload, "phase": "setup", "step": -1}

    def measure(wl, steps, warmup, setup=True, spread=None):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides;
        (wall seconds, device milliseconds between HIP events on the launch stream), each the
        maximum over ranks.  spread: a dict that receives the fastest and the slowest rank's
        wall time per step (N > 1)."""
        if setup:
            progress.update(phase="setup", step=-1)
            wl.setup(device)
        progress.update(phase="warmup")
        for i in range(warmup):
            progress["step"] = i
            wl.step(i)
        barrier()
        progress.update(phase="timed")
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            progress["step"] = i
            wl.step(i)
        ev1.record()
        progress.update(phase="barrier")
        barrier()
        elapsed = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)  # same stream as the launches (torch's current stream)
        if dist is not None:
            t = torch.tensor([elapsed, dev_ms], dtype=torch.float64,
                             device="cpu" if rehearse else device)
            lo = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            if spread is not None:
                spread["rank_ms_per_step_min"] = float(lo[0]) * 1e3 / steps
                spread["rank_ms_per_step_max"] = float(t[0]) * 1e3 / steps
            elapsed, dev_ms = float(t[0]), float(t[1])
        return elapsed, dev_ms

    def exchange_report(wl, steps, warmup, ms_with):
        """N > 1, a workload with a gradient exchange: the same ranks time the same step with
        the exchange switched off (every rank keeps its local gradients), so the line states
        how much of the step the all-reduce leaves exposed behind backward.  Runs after the
        cross-rank parameter check: the ranks' parameters drift apart while it is off."""
        net = getattr(wl, "net", None)
        bk = getattr(net, "bucketer", None)
        if dist is None or bk is None:
            return None
        rep = dict(bk.describe())
        rep["collective_env"] = collective_env()
        try:
            bk.enabled = False
            progress.update(phase="no-exchange")
            e0, _ = measure(wl, steps, max(2, min(warmup, 5)), setup=False)
            rep["ms_per_step_without_exchange"] = e0 * 1e3 / steps
            rep["exchange_exposed_ms"] = ms_with - e0 * 1e3 / steps
            # bytes each rank sends per step under a ring all-reduce, over the exposed time and
            # over the whole step: lower bounds on what the links carried
            ring = 2.0 * (world - 1) / world * rep["bytes_per_step"]
            rep["ring_bytes_per_rank_per_step"] = int(ring)
            rep["bus_gbs_over_step"] = ring / (ms_with * 1e-3) / 1e9
        finally:
            bk.enabled = True
        return rep

    traffic_file = ROOT / "profiles" / "traffic.json"
    traffic = json.loads(traffic_file.read_text()) if traffic_file.exists() else {}

    def roofline_of(wl, kern_ms):
        if hasattr(wl, "roofline"):
            rl = wl.roofline(kern_ms)
        else:
            achieved = wl.bytes_per_step / (kern_ms * 1e-3) / 1e9
            rl = {"bound": wl.bound, "kernel": wl.kernel, "achieved": achieved,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                  "bytes_per_launch": wl.bytes_per_step, "launch_ms": kern_ms, "traffic": None}
        # HBM bytes per launch (per step for the network workloads) from the committed
        # rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, gfx950-corrected); null when this
        # workload has not been profiled
        rec = traffic.get(wl.name)
        if rec:
            rl["traffic"] = rec["traffic_bytes_per_launch"]
            if rl["bound"] == "mfma":
                # the network steps move enough bytes for HBM to matter as much as the matrix
                # cores: the measured traffic over this run's step time, against the HBM peak
                rl["hbm_frac"] = rl["traffic"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                alg = rec.get("algorithmic_bytes_per_launch")
                if alg:
                    # bytes the step must move if every tensor is touched once per pass that
                    # needs it (flops.train_step_bytes / the forward's tensor bytes): what
                    # the measured traffic is a multiple of, and the step's HBM floor
                    rl["algorithmic_bytes"] = alg
                    rl["traffic_over_algorithmic"] = rl["traffic"] / alg
                    rl["hbm_floor_ms"] = alg / (HBM_PEAK_GBS * 1e9) * 1e3
        return rl

    wl = WORKLOADS[args.workload]()
    spread0 = {}
    elapsed, dev_ms = measure(wl, args.steps, args.warmup, spread=spread0)
    out = None
    if rank == 0:
        out = {
            "metric": METRIC,
            # `value` counts images of the workload named in config.workload (for decode: batches
            # of 17 heat maps per image); the network images/s of the metric are under "also"
            "value_of": f"{wl.name}: images/s of this workload over all ranks",
            "value": world * wl.batch * args.steps / elapsed,
            "unit": "images/s" if args.workload != "adam" else "steps/s",
            "n_gpus": world,
            "ranks": dist.get_world_size() if dist is not None else 1,
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic",
            "config": {"workload": wl.name, "batch_per_gpu": wl.batch, "joints": K_JOINTS,
                       "heatmap": list(getattr(wl, "heatmap", (H, W))),
                       "sharding": getattr(wl, "sharding", "batch, no collective")},
            "roofline": roofline_of(wl, dev_ms / args.steps),
        }
    if out is not None and spread0:
        out.update(spread0)
    if dist is not None and hasattr(wl, "dp_check"):
        chk = wl.dp_check(dist, rehearse)       # a collective: every rank takes part
        rep = exchange_report(wl, max(1, min(args.steps, 30)), args.warmup,
                              elapsed * 1e3 / args.steps)
        if out is not None:
            out["dp_check"] = chk
            if rep is not None:
                out["exchange"] = rep

    # The metric also names network inference and the data-parallel training step.  A default
    # run measures them after the headline workload, with the same barrier-bracketed loop, and
    # reports them under "also" in the same JSON line (whole-job images/s over all ranks), so a
    # 1/2/4/8-GPU sweep of this script shows how the training step's gradient all-reduce scales.
    # They never change "value".  A failure or a stall in them must not lose the headline
    # number: errors are recorded in the line; a watchdog prints the line, says which leg and
    # step were in flight, and ends the process with a non-zero status.
    legs = []
    if args.workload == "decode" and not args.no_also:
        import threading

        also = {}
        if out is not None:
            out["also"] = also

        def give_up():
            msg = (f"secondary workloads exceeded {args.also_timeout} s in leg "
                   f"{progress['leg']!r}, phase {progress['phase']!r}, step {progress['step']}")
            if out is not None:
                also.setdefault("error", msg)
                print(json.dumps(out), flush=True)
            print(f"bench.py rank {rank}: {msg}", file=sys.stderr, flush=True)
            os._exit(3)

        dog = threading.Timer(args.also_timeout, give_up)
        dog.daemon = True
        dog.start()
        first = wl
        wl = None
        steps2, warm2 = max(1, min(args.steps, 50)), max(1, min(args.warmup, 10))
        failed = False
        # training first: its weight-gradient and projection streams are then the third and
        # fourth streams this process puts work on (after the default stream and, in an N > 1
        # run, the collective's), so with four hardware queues per process none of the streams
        # the data-parallel step overlaps on has to share a queue; the inference leg's side
        # stream comes last and only that leg can feel it
        for key in ("train", "infer"):
            progress.update(leg=key)
            torch.cuda.empty_cache()
            try:
                w2 = WORKLOADS[key]()
                spread2 = {}
                e2, d2 = measure(w2, steps2, warm2, spread=spread2)
                also[key] = {"value": world * w2.batch * steps2 / e2, "unit": "images/s",
                             "ms_per_step": e2 * 1e3 / steps2, "steps": steps2, "warmup": warm2,
                             "dtype": w2.dtype,
                             "config": {"workload": w2.name, "batch_per_gpu": w2.batch,
                                        "sharding": getattr(w2, "sharding",
                                                            "batch, no collective")},
                             "roofline": roofline_of(w2, d2 / steps2)}
                also[key].update(spread2)
                if dist is not None and hasattr(w2, "dp_check"):
                    also[key]["dp_check"] = w2.dp_check(dist, rehearse)
                    rep = exchange_report(w2, max(1, min(steps2, 30)), warm2, e2 * 1e3 / steps2)
                    if rep is not None:
                        also[key]["exchange"] = rep
                legs.append((key, w2))
            except Exception as exc:  # noqa: BLE001 - recorded in the line, headline kept
                also[key] = {"error": (f"{type(exc).__name__} in leg {key!r}, phase "
                                       f"{progress['phase']!r}, step {progress['step']} "
                                       f"(backend {backend_name(dist)}): {exc}")[:500]}
                failed = True
                break   # the ranks may no longer be in step: no further collectives
            finally:
                w2 = None
        dog.cancel()
        wl = first
        if failed and dist is not None:
            if out is not None:
                print(json.dumps(out), flush=True)
            os._exit(4)   # ranks out of step: no barrier can be trusted any more

    # CPU baselines (rank 0 of a one-GPU run only): the oracle of each measured leg timed on the
    # host cores, a bounded sample each; GPU work is over, so they do not disturb the timings
    if out is not None and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = wl.cpu_baseline()
        for key, w2 in legs:
            try:
                out["also"][key]["cpu_baseline"] = w2.cpu_baseline()
            except Exception as exc:  # noqa: BLE001
                out["also"][key]["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
