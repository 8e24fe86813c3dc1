"""Configuration in the reference's shape: a YAML file (or dict) with three sections,
`data_config`, `train_config`, `model_config`, each exposed as an object with attributes
(`parse_config('config.yaml')` at test.ipynb:381-386; keys at test.ipynb:962-974, 995-1014).

The reference's own config.yaml is absent; the defaults below are the values its notebook
prints, plus the model keys this build needs.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Union

import yaml


class Section:
    """Attribute view of one config section (the reference's objects expose __dict__)."""

    def __init__(self, d: Dict[str, Any]):
        self.__dict__.update(copy.deepcopy(d))

    def get(self, key: str, default=None):
        return self.__dict__.get(key, default)

    def __repr__(self) -> str:
        return f"Section({self.__dict__!r})"


# The data set's keypoint vocabulary, in annotation order.  The reference selects the joints it
# trains on BY NAME (`train_keypoints: ['left_wrist', 'right_wrist']`, test.ipynb:1013); its
# annotation files are COCO-style (test.ipynb:41-44), whose person keypoints are these 17.
COCO_KEYPOINTS = ["nose", "left_eye", "right_eye", "left_ear", "right_ear", "left_shoulder",
                  "right_shoulder", "left_elbow", "right_elbow", "left_wrist", "right_wrist",
                  "left_hip", "right_hip", "left_knee", "right_knee", "left_ankle", "right_ankle"]


DEFAULTS: Dict[str, Dict[str, Any]] = {
    "data_config": {
        "synthetic": True,          # the reference's TFRecords are private; batches are synthetic
        "seed": 0,
        "keypoints": None,          # names of the annotated joints; None = the 17 COCO names
    },
    "train_config": {               # test.ipynb:995-1014
        "batch_size": 2,
        "is_training": True,
        "learning_rate": 0.001,
        "learning_rate_decay": {"decay_rate": 0.95, "decay_steps": 10000},
        "optimizer": {"name": "adam",
                      "params": {"beta1": 0.9, "beta2": 0.999, "epsilon": "1e-8"}},
        "preprocess": {"image_resize": [256, 192]},
        # the reference trains with {"flip_left_right": True, "random_crop": True,
        # "scale_range": [0.5, 2.0]} (test.ipynb:995-997); None = frames are fed as they are
        "augmentation": None,
        "model_dir": "./models",
        "num_epochs": 0,
        "train_keypoints": None,
    },
    "model_config": {
        "backbone": "resnet50",
        "num_joints": 17,
        "layers": [3, 4, 6, 3],
        "head_channels": 256,
        "heatmap_stride": 4,
        "sigma": 2.0,
        "bn_momentum": 0.1,
        "bucket_mb": 32,
    },
}


def _merge(base: Dict[str, Any], over: Dict[str, Any]) -> Dict[str, Any]:
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def parse_config(source: Union[str, Dict[str, Any], None] = None) -> Dict[str, Section]:
    """YAML path or dict -> {'data_config', 'train_config', 'model_config'} of Section objects.
    Unknown keys are kept; missing ones take the defaults."""
    if source is None:
        raw: Dict[str, Any] = {}
    elif isinstance(source, dict):
        raw = source
    else:
        with open(source) as f:
            raw = yaml.safe_load(f) or {}
    unknown = set(raw) - set(DEFAULTS)
    if unknown:
        raise KeyError(f"unknown config sections: {sorted(unknown)}")
    return {name: Section(_merge(DEFAULTS[name], raw.get(name) or {})) for name in DEFAULTS}


def keypoint_names(data_cfg: Section, model_cfg: Section):
    """The data set's keypoint names: `data_config.keypoints`, else the COCO names when the
    model has 17 joints, else 'kp0', 'kp1', ..."""
    names = data_cfg.get("keypoints")
    if names:
        return [str(n) for n in names]
    k = int(model_cfg.get("num_joints", 17))
    return list(COCO_KEYPOINTS) if k == 17 else [f"kp{i}" for i in range(k)]


def select_keypoints(names, train_keypoints):
    """`train_config.train_keypoints` (names, test.ipynb:1013) -> indices into `names`, in the
    order given; None selects everything.  Unknown or repeated names are an error that lists
    the vocabulary."""
    if not train_keypoints:
        return None
    idx = []
    for n in train_keypoints:
        if n not in names:
            raise KeyError(f"train_keypoints: {n!r} is not a keypoint of the data set {names}")
        if names.index(n) in idx:
            raise ValueError(f"train_keypoints: {n!r} is listed twice")
        idx.append(names.index(n))
    return idx


def flip_permutation(names):
    """For a left-right mirrored image: joint i of the output is joint perm[i] of the input.
    'left_X' and 'right_X' swap when both are in `names`; everything else stays.  A name whose
    partner is missing keeps its own row (there is no label to take over)."""
    perm = list(range(len(names)))
    for i, n in enumerate(names):
        for a, b in (("left_", "right_"), ("right_", "left_")):
            if n.startswith(a) and (b + n[len(a):]) in names:
                perm[i] = names.index(b + n[len(a):])
    return perm


def learning_rate_at(train_cfg: Section, step: int) -> float:
    """Exponential decay lr * rate^(step/steps) (continuous, as tf.train.exponential_decay)."""
    lr = float(train_cfg.learning_rate)
    dec = train_cfg.get("learning_rate_decay")
    if dec:
        lr *= float(dec["decay_rate"]) ** (step / float(dec["decay_steps"]))
    return lr
