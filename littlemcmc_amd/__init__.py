"""littlemcmc_amd -- MI355X-native many-chain HMC/NUTS engine behind littlemcmc's API.

The public names are the ones /root/reference/littlemcmc/__init__.py:19-29 exports, plus the GPU-side pieces
(``targets``, ``Engine``, ``diagnostics``, ``distributed``). Numerics live in liblmc_hip.so (HIP, gfx950);
importing the package needs no GPU, using it does -- there is no CPU fallback."""

__version__ = "0.1.0"

from . import diagnostics, distributed, quadpotential as _qp, targets
from .base_hmc import StepRandUniform
from .engine import Engine
from .hmc import HamiltonianMC
from .nuts import NUTS
from .sampling import init_nuts, sample

# mass matrices: diagonal (the hot path) and dense (model_ndim <= 256), all on the device
quad_potential = _qp.quad_potential
QuadPotentialDiag, QuadPotentialDiagAdapt = _qp.QuadPotentialDiag, _qp.QuadPotentialDiagAdapt
QuadPotentialFull, QuadPotentialFullInv, QuadPotentialFullAdapt = (
    _qp.QuadPotentialFull, _qp.QuadPotentialFullInv, _qp.QuadPotentialFullAdapt)

__all__ = ["sample", "init_nuts", "HamiltonianMC", "NUTS", "quad_potential", "QuadPotentialDiag", "QuadPotentialFull",
           "QuadPotentialFullInv", "QuadPotentialDiagAdapt", "QuadPotentialFullAdapt", "Engine", "StepRandUniform", "targets",
           "diagnostics", "distributed"]
