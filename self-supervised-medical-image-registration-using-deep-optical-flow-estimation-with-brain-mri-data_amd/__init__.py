"""mireg -- MI355X-native registration hot path.

Public surface mirrors the reference's Python API for the hot path:
  opticalFlowReg, FlowNetS, FlowNetC, PWCDCNet, Correlation, OFEloss, stn, dice_average
Everything computes through libmireg_hip.so (hand-written gfx950 kernels); nothing here falls
back to ATen/MIOpen or to the CPU oracle.
"""
from . import _lib  # noqa: F401
from .ops import OFEloss, dice_average, dice_batch, resize_bilinear, seg_round, stn  # noqa: F401
from .flownets import FlowNetS  # noqa: F401
from .flownetc import FlowNetC  # noqa: F401
from .pwcnet import PWCDCNet  # noqa: F401
from .correlation import Correlation  # noqa: F401
from .affine3d import Affloss, affmodel  # noqa: F401
from .models import generate_grid, grid_generator, opticalFlowReg  # noqa: F401
from .trainer import RegistrationTrainer  # noqa: F401

__all__ = ["affmodel", "Affloss", "FlowNetS", "FlowNetC", "PWCDCNet", "Correlation", "opticalFlowReg", "RegistrationTrainer", "generate_grid", "grid_generator", "OFEloss", "dice_average", "dice_batch", "resize_bilinear", "seg_round", "stn"]
