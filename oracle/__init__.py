"""CPU oracle for the mireg hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32, CPU) restatement of the reference
algorithm for the registration hot path.  It exists so
that the hand-written HIP kernels can be checked against something that was
itself pinned to the reference's behaviour (``oracle/gen_golden.py`` imports the
reference from /root/reference in the build container, compares it with this
restatement and writes small fixtures under ``tests/golden/``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here.  The product package
(``mireg``) never does: it fails loudly when its HIP library is missing.

Parity status
  * FlowNetS / stn / losses / Dice / generate_grid / affine 3-D ops: pinned by
    running the reference's own Python on CPU (fixtures G1-G4, G7, G8).
  * Correlation / PWC warp arithmetic: the reference imports these from
    un-vendored third-party CUDA packages (NVIDIA/flownet2-pytorch
    ``correlation_package``, no version pinned; ClementPinard
    ``spatial_correlation_sampler``).  PARITY UNPINNED for the correlation
    arithmetic itself: the restatement follows the published definition
    (mean over channels of f1 . shifted f2, zero outside) and is anchored on
    the reference's structural pins (441 / 81 output channels, same HxW,
    1/C normalisation) only.  PWCDCNet.warp is plain torch in the reference
    (PWC/models/PWCNet.py:143-179) and is pinned by running it (G6).
  * FlowNet2 stack (FlowNetSD, FlowNetFusion, the 6-channel FlowNetS, the
    FlowNet2 chain): pinned by running the reference's classes (G9) with this
    package's Correlation / Resample2d / ChannelNorm in place of the three
    external CUDA layers; Resample2d and ChannelNorm themselves are PARITY
    UNPINNED (NVIDIA/flownet2-pytorch custom layers, absent and unversioned:
    published definitions restated).
"""
from . import ops, nets  # noqa: F401
