"""MI355X-native (gfx950) pose-estimation hot path.

Python host code over libpe_hip.so, a C-ABI library of hand-written HIP kernels.
torch tensors are buffers only.  Importing the package does not load the library;
the first op does, and fails loudly if it has not been built
(`python -m pose_estimation_amd.build`).
"""
from . import _ffi  # noqa: F401
from .ops import (  # noqa: F401
    adam_scalars,
    adam_step,
    decode,
    gaussian_table,
    mse_fwd_bwd,
    mse_workspace,
    render_targets,
    render_vecmaps,
    vecmap_loss_fwd_bwd,
    sgd_step,
)

from .config import parse_config  # noqa: F401
from .trainer import Trainer  # noqa: F401

__version__ = "0.1.0"
