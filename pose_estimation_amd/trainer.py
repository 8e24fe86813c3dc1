"""`Trainer`: the reference's entry object (`Trainer('config.yaml')`,
`trainer.get_features_labels_data()`, test.ipynb:400-401) over the HIP training graph.

The reference's Trainer built a tf.data pipeline and (presumably) an Estimator loop; neither
survives in the snapshot.  This Trainer keeps the constructor and the `(features, labels)`
batch shape (`features['images']`, `labels['heatmaps']` NHWC, `labels['masks']`), and adds the
per-batch calls the hot path needs: `train_step`, `predict`, `decode`.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from . import augment, ops, synthetic
from .config import (flip_permutation, keypoint_names, learning_rate_at, parse_config,
                     select_keypoints)
from .model import init_state
from .training import TrainablePoseResNet


class Trainer:
    def __init__(self, config=None, device: Optional[str] = None):
        cfg = parse_config(config)
        self.data_cfg = cfg["data_config"]
        self.train_cfg = cfg["train_config"]
        self.model_cfg = cfg["model_config"]
        if device is None:
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        self.device = torch.device(device)
        # The reference names the joints it trains on (`train_keypoints`, test.ipynb:1013): the
        # names select rows of the data set's keypoint list, and their count is the number of
        # heat maps the model predicts (it overrides model_config.num_joints).
        self.keypoints = keypoint_names(self.data_cfg, self.model_cfg)
        self.train_index = select_keypoints(self.keypoints, self.train_cfg.get("train_keypoints"))
        if self.train_index is not None:
            self.model_cfg.num_joints = len(self.train_index)
            self.trained_keypoints = [self.keypoints[i] for i in self.train_index]
        else:
            self.trained_keypoints = list(self.keypoints)
            self.model_cfg.num_joints = len(self.keypoints)    # an explicit name list decides
        self._train_index_dev: Optional[torch.Tensor] = None
        self.net: Optional[TrainablePoseResNet] = None
        self.global_step = 0
        self._folded_at = -1     # global_step at which the inference constants were last folded

    @staticmethod
    def _rank() -> int:
        import torch.distributed as dist

        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    # ------------------------------------------------------------------ model
    def build_model(self, state: Optional[Dict[str, torch.Tensor]] = None, seed: int = 3):
        m = self.model_cfg
        if m.backbone == "resnet50":
            self.net = TrainablePoseResNet(self.device, num_joints=m.num_joints,
                                           layers=tuple(m.layers), head_ch=m.head_channels,
                                           bn_momentum=m.bn_momentum, sigma=m.sigma,
                                           heatmap_stride=float(m.heatmap_stride))
            self.net.load_state(state if state is not None else init_state(self.net.specs, seed))
        elif m.backbone in ("hrnet_w32", "hrnet"):
            from . import hrnet

            if int(m.heatmap_stride) != 4:
                raise ValueError("the HRNet-style network predicts at stride 4")
            self.net = hrnet.TrainablePoseHRNet(
                self.device, num_joints=m.num_joints, width=int(m.get("width", 32)),
                modules=tuple(m.get("modules", (1, 4, 3))), blocks=int(m.get("blocks", 4)),
                layer1_blocks=int(m.get("layer1_blocks", 4)), bn_momentum=m.bn_momentum,
                sigma=m.sigma, heatmap_stride=4.0)
            self.net.load_state(state if state is not None
                                else hrnet.init_state(self.net.spec, seed))
        elif m.backbone in ("mobilenet_v2", "mobilenet"):
            from . import mobilenet as mb

            stride = int(m.heatmap_stride)
            pairs = m.get("pairs")
            self.net = mb.TrainablePoseMobileNetV2(
                self.device, num_joints=m.num_joints,
                pairs=[tuple(p) for p in pairs] if pairs else None, output_stride=stride,
                bn_momentum=m.bn_momentum, sigma=m.sigma, radius=int(m.get("radius", 1)),
                detection=bool(m.get("detection", False)),
                box_loss_weight=float(m.get("box_loss_weight", 1.0)))
            self.net.load_state(state if state is not None else mb.init_state(
                self.net.blocks, m.num_joints, self.net.num_pairs, seed,
                det_blocks=self.net.det_blocks))
        else:
            raise NotImplementedError(f"backbone {m.backbone!r} is not built yet")
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.net.broadcast_parameters(0)
            self.net.enable_data_parallel(int(m.bucket_mb) << 20)
        return self.net

    def _need_model(self) -> TrainablePoseResNet:
        if self.net is None:
            self.build_model()
        return self.net

    # ------------------------------------------------------------------ keypoint subset
    def _trained(self, joints: torch.Tensor) -> torch.Tensor:
        """Rows of the trained keypoints out of joints [B, K_all, 3] (a gather; joints that
        already hold only the trained rows pass through)."""
        if self.train_index is None or joints.shape[1] == len(self.train_index):
            return joints
        if joints.shape[1] != len(self.keypoints):
            raise ValueError(f"joints carry {joints.shape[1]} keypoints; the data set has "
                             f"{len(self.keypoints)}, of which {len(self.train_index)} are trained")
        if self._train_index_dev is None or self._train_index_dev.device != joints.device:
            self._train_index_dev = torch.tensor(self.train_index, dtype=torch.int64,
                                                 device=joints.device)
        return joints.index_select(1, self._train_index_dev).contiguous()

    def _flip_perm(self, k: int):
        """Left/right relabelling for mirrored frames, by keypoint NAME: over the whole data
        set's list when the joints still carry every keypoint, else over the trained ones."""
        if k == len(self.keypoints):
            return flip_permutation(self.keypoints)
        if k == len(self.trained_keypoints):
            return flip_permutation(self.trained_keypoints)
        return None

    # ------------------------------------------------------------------ detection labels
    def _anchors(self, H: int, W: int) -> torch.Tensor:
        """Anchors of the two detection levels (strides 16 and 32: 20x20 and 10x10 cells, three
        per cell, at the reference's 320x320; test.ipynb:423-464)."""
        from . import detection

        key = (H, W)
        if getattr(self, "_anchor_key", None) != key:
            levels = ((H // 16, W // 16), (H // 32, W // 32))
            self._anchor_cache = torch.from_numpy(detection.make_anchors(levels)).to(self.device)
            self._anchor_key = key
        return self._anchor_cache

    def _bbox_labels(self, joints: torch.Tensor, H: int, W: int) -> dict:
        """`labels['bboxes']` in the reference's keys (`classes`, `weights`, `regs`,
        test.ipynb:504-507, 571-572), made on the GPU from one person box per image."""
        from . import detection

        boxes, cls = detection.boxes_from_joints(joints, H, W)
        classes, weights, regs, _ = detection.anchor_assign(self._anchors(H, W), boxes, cls)
        out = {"classes": classes, "weights": weights, "regs": regs, "boxes": boxes}
        # the reference also exposes the labels per feature level: `feat_1` = the 20x20x3
        # anchors of the stride-16 map, `feat_2` = the 10x10x3 of the stride-32 map, each with
        # its own classes / weights / regs (test.ipynb:538-541, 571-572).  Views, no copies.
        n1 = (H // 16) * (W // 16) * 3
        for name, sl in (("feat_1", slice(0, n1)), ("feat_2", slice(n1, classes.shape[1]))):
            out[name] = {"classes": classes[:, sl], "weights": weights[:, sl],
                         "regs": regs[:, sl]}
        return out

    # ------------------------------------------------------------------ data (synthetic)
    def get_features_labels_data(self, num_batches: Optional[int] = None
                                 ) -> Iterator[Tuple[dict, dict]]:
        """Yields (features, labels) in the reference's batch shape, labels rendered on the GPU
        straight into the NHWC layout the reference's notebook reads (test.ipynb:919-931)."""
        B = int(self.train_cfg.batch_size)
        H, W = self.train_cfg.preprocess["image_resize"]
        if H % 32 or W % 32:
            raise ValueError(f"image_resize {H}x{W}: the network needs multiples of 32 "
                             "(five stride-2 stages down, three up)")
        K = len(self.keypoints)          # every annotated joint; the trained ones are a subset
        s = int(self.model_cfg.heatmap_stride)
        sigma = float(self.model_cfg.sigma)
        i = 0
        rank = self._rank()      # every data-parallel rank draws its own shard of the batch
        while num_batches is None or i < num_batches:
            seed = int(self.data_cfg.seed) + 1000 * i + 100003 * rank
            img = torch.from_numpy(synthetic.images_u8(B, H, W, seed + 1)).to(self.device)
            joints = torch.from_numpy(synthetic.joints(B, K, H, W, seed + 2)).to(self.device)
            mask = torch.from_numpy(synthetic.ignore_mask(B, H // s, W // s, seed + 4)).to(self.device)
            radius = self.model_cfg.get("radius")
            tj = self._trained(joints)
            hm, w = ops.render_targets(tj, H // s, W // s, float(s), sigma,
                                       None if radius is None else int(radius), layout="NHWC")
            # the mask covers the frame it was drawn on; train_step warps it with the frame when
            # it resizes or augments
            labels = {"heatmaps": hm, "joints": joints, "weights": w, "masks": mask,
                      "keypoints": list(self.trained_keypoints)}
            pairs = self.model_cfg.get("pairs")
            if pairs:   # the reference's second dense label (test.ipynb:925-944); pair indices
                #         count the TRAINED keypoints, as the notebook's do (test.ipynb:926)
                labels["vecmaps"] = ops.render_vecmaps(tj, [tuple(p) for p in pairs], H // s,
                                                       W // s, float(s), layout="NHWC")
            if self.model_cfg.get("detection"):   # the third label group (test.ipynb:432)
                labels["bboxes"] = self._bbox_labels(joints, H, W)
            yield ({"images": img}, labels)
            i += 1

    # ------------------------------------------------------------------ per-batch calls
    def train_step(self, features, labels=None, mask=None):
        """train_step(features_dict, labels_dict) or train_step(images_u8, joints[, mask]).
        Returns the loss as a Python float is NOT done here (no sync): a [1] device tensor."""
        net = self._need_model()
        bb = None
        if isinstance(features, dict):
            images, joints, mask = features["images"], labels["joints"], labels.get("masks")
            bb = labels.get("bboxes")
        else:
            images, joints = features, labels
        t = self.train_cfg
        x_s2d = None
        H, W = t.preprocess["image_resize"]
        aug_cfg = getattr(t, "augmentation", None)
        if aug_cfg or tuple(images.shape[1:3]) != (H, W):
            # frames of any size: resize (+ the configured augmentation) on the GPU, straight
            # into the stem's input layout; joints and the ignore mask follow the same map
            B = images.shape[0]
            src_hw = (int(images.shape[1]), int(images.shape[2]))
            if aug_cfg:
                prm = augment.random_params(
                    B, src_hw, (H, W), tuple(aug_cfg.get("scale_range", (1.0, 1.0))),
                    bool(aug_cfg.get("flip_left_right", False)),
                    bool(aug_cfg.get("random_crop", False)),
                    seed=int(self.data_cfg.seed) + 7919 * self.global_step + 104729 * self._rank())
            else:
                prm = augment.resize_params(B, src_hw, (H, W))
            prm_t = torch.from_numpy(prm).to(images.device)
            x_s2d = augment.augment_u8(images, prm_t, (H, W))
            joints = augment.transform_joints(joints, prm_t, W, self._flip_perm(joints.shape[1]))
            if mask is not None:
                # the given mask covers the SOURCE frame in square cells; cells of the output
                # frame that fall outside it are trained as background (fill = 1)
                hs = int(self.model_cfg.heatmap_stride)
                if src_hw[0] % mask.shape[1] or src_hw[0] // mask.shape[1] * mask.shape[2] != src_hw[1]:
                    raise ValueError(f"mask {tuple(mask.shape[1:])} does not tile the "
                                     f"{src_hw[0]}x{src_hw[1]} frames in square cells")
                mask = augment.augment_mask(mask, prm_t, float(src_hw[0] // mask.shape[1]),
                                            (H // hs, W // hs), float(hs), fill=1)
            images = None
            bb = None     # boxes follow the transformed joints
        kw = {}
        if getattr(net, "det_blocks", None):
            if bb is None:
                bb = self._bbox_labels(joints, H, W)     # the person box spans every joint
            kw["det_labels"] = (bb["classes"], bb["weights"], bb["regs"])
        joints = self._trained(joints)
        lr = learning_rate_at(t, self.global_step)
        opt = t.optimizer
        params = opt.get("params", {})
        name = opt.get("name", "adam").lower()
        if name == "adam":
            loss, _ = net.train_step(images, joints, mask, lr=lr, optimizer="adam",
                                     beta1=float(params.get("beta1", 0.9)),
                                     beta2=float(params.get("beta2", 0.999)),
                                     eps=float(params.get("epsilon", 1e-8)), x_s2d=x_s2d, **kw)
        elif name in ("sgd", "momentum"):
            loss, _ = net.train_step(images, joints, mask, lr=lr, optimizer="sgd",
                                     momentum=float(params.get("momentum", 0.9)), x_s2d=x_s2d,
                                     **kw)
        else:
            raise ValueError(f"unknown optimizer {name!r}")
        self.global_step += 1
        return loss

    def predict(self, images_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] -> float32 heat maps [B,K,H/4,W/4] (running BN statistics)."""
        net = self._need_model()
        if self._folded_at != self.global_step:   # parameters moved since the last fold
            net.fold_bn()
            self._folded_at = self.global_step
        return net.predict(images_u8)

    def decode(self, heatmaps: torch.Tensor, layout: str = "NCHW"):
        """Heat maps -> (xy [B,K,2], conf [B,K], idx [B,K]) in heat-map cells."""
        return ops.decode(heatmaps, layout=layout)

    # ------------------------------------------------------------------ checkpoint
    def save(self, path: str) -> None:
        """Writes the checkpoint from rank 0 only (data-parallel ranks hold identical parameters;
        the running batch-norm statistics saved are rank 0's)."""
        net = self._need_model()
        if self._rank() != 0:
            return
        torch.save({"master": net.master.cpu(), "opt_m": net.opt_m.cpu(), "opt_v": net.opt_v.cpu(),
                    "bn_mean": {k: v.cpu() for k, v in net.bn_mean.items()},
                    "bn_var": {k: v.cpu() for k, v in net.bn_var.items()},
                    "step": net.step_count, "global_step": self.global_step}, path)

    def load(self, path: str) -> None:
        net = self._need_model()
        ck = torch.load(path, weights_only=True)
        net.master.copy_(ck["master"])
        net.opt_m.copy_(ck["opt_m"])
        net.opt_v.copy_(ck["opt_v"])
        for k, v in ck["bn_mean"].items():
            net.bn_mean[k].copy_(v)
        for k, v in ck["bn_var"].items():
            net.bn_var[k].copy_(v)
        net.step_count = int(ck["step"])
        self.global_step = int(ck["global_step"])
        net.work.copy_(net.master)
        net.refresh_transposes()
        net.fold_bn()
        self._folded_at = self.global_step
