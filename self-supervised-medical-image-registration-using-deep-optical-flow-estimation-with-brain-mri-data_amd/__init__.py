"""mireg -- MI355X-native registration hot path.

Public surface mirrors the reference's Python API for the hot path:
  opticalFlowReg, FlowNetS, FlowNetC, PWCDCNet, Correlation, OFEloss, stn, dice_average
Everything computes through libmireg_hip.so (hand-written gfx950 kernels); nothing here falls
back to ATen/MIOpen or to the CPU oracle.
"""
from . import _lib  # noqa: F401
from .ops import (OFEloss, correlation_loss, dice_average, dice_batch, photometric_loss, resize_bilinear, seg_round,  # noqa: F401
                  smoothness_loss, stn)
from .flownets import FlowNetS  # noqa: F401
from .flownetc import FlowNetC  # noqa: F401
from .pwcnet import PWCDCNet  # noqa: F401
from .correlation import Correlation  # noqa: F401
from .affine3d import Affloss, affmodel, correlation_loss_3d, photometric_loss_3d  # noqa: F401
from .volume import OFEloss3d, resize_trilinear, smoothness_loss_3d, stn3d  # noqa: F401
from .flownets3d import FlowNetS3D, opticalFlowReg3d  # noqa: F401
from .models import generate_grid, grid_generator, opticalFlowReg  # noqa: F401
from .trainer import RegistrationTrainer  # noqa: F401
from .optim import Adam  # noqa: F401
from .tuning import autotune  # noqa: F401
from . import checkpoint  # noqa: F401
from .flownet2_ops import ChannelNorm, Resample2d, Upsample  # noqa: F401
from .flownet2 import FlowNet2, FlowNet2S, FlowNetFusion, FlowNetSD  # noqa: F401
from .metrics import (CORR, MI, MSE, PSNR, dist_hausdorff, extract_boundary_points, modified_hausdorff, pair_metrics, ssim_batch,  # noqa: F401
                      structural_similarity)

__all__ = ["autotune", "dist_hausdorff", "extract_boundary_points", "FlowNet2", "FlowNet2S", "FlowNetSD", "FlowNetFusion", "modified_hausdorff", "structural_similarity", "ssim_batch", "Resample2d", "ChannelNorm", "Upsample", "MSE", "PSNR", "CORR", "MI", "pair_metrics", "Adam", "FlowNetS3D", "opticalFlowReg3d", "OFEloss3d", "stn3d", "resize_trilinear", "smoothness_loss_3d", "affmodel", "Affloss", "FlowNetS", "FlowNetC", "PWCDCNet", "Correlation", "opticalFlowReg", "RegistrationTrainer", "generate_grid", "grid_generator", "OFEloss", "photometric_loss", "correlation_loss", "smoothness_loss", "photometric_loss_3d", "correlation_loss_3d", "dice_average", "dice_batch", "resize_bilinear", "seg_round", "stn"]
