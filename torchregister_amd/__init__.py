"""MI355X-native TorchRegister hot path (drop-in for AgamChopra/TorchRegister's Register API).

`import torchregister_amd as tr` (or `import TorchRegister as tr` through the alias package)
gives `tr.Register(mode=...).optim(moving, target, ...)` / `reg(moving)` backed by hand-written
HIP kernels in lib/libtrx.so (C ABI: include/trx.h).  There is no CPU fallback.
"""
__version__ = "0.2.3"   # = the reference package version this host layer mirrors (ref:src/TorchRegister/__init__.py:4); C ABI: trx_version()

from ._engine import AffineSolver, FlowSolver, LossSpec, SlabFlowSolver, SlabPeers, run_slabs_lockstep  # noqa: F401
from .sharding import register_sharded  # noqa: F401
from .torchregister import Register  # noqa: F401
from .utils import (EPSILON, Attention_UNet, K_gauss, LocalNCCLoss, NCCLoss, NMI, NMILoss, PDF, PDF_xis, Regressor, SpatialTransformer, SSDLoss,  # noqa: F401
                    Theta, attention_grid, get_pdf, norm, padNd)
from .warpings import affine_register, compose_theta, flow_register, get_affine_warp, rigid_register  # noqa: F401
