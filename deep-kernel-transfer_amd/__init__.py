"""dkt_amd -- MI355X-native (gfx950) implementation of the DKT hot path: deep-kernel Gram build and
exact-GP marginal likelihood (jittered Cholesky, log-det, solves, posterior mean) as hand-written HIP
kernels behind a C ABI (include/dkt_abi.h), with the reference's Python method surface on top.

Import as `import dkt_amd` (see dkt_amd.py at the repository root)."""
from . import _lib, backbone, configs, distributed, gp, ops  # noqa: F401
from .dkt import DKT  # noqa: F401
from .dkt_regression import DKT as DKTRegression  # noqa: F401

__all__ = ["DKT", "DKTRegression", "ops", "gp", "backbone", "configs", "distributed", "_lib"]
