"""littlemcmc_amd -- MI355X-native many-chain HMC/NUTS engine behind littlemcmc's API.

Export list mirrors /root/reference/littlemcmc/__init__.py:19-29. Numerics live in
liblmc_hip.so (HIP, gfx950); importing the package does not need a GPU, using it does."""

__version__ = "0.1.0"

from . import diagnostics, distributed, targets
from .engine import Engine
from .hmc import HamiltonianMC
from .nuts import NUTS
from .quadpotential import (
    QuadPotentialDiag,
    QuadPotentialDiagAdapt,
    QuadPotentialFull,
    QuadPotentialFullAdapt,
    QuadPotentialFullInv,
    quad_potential,
)
from .sampling import init_nuts, sample

__all__ = [
    "sample", "init_nuts", "HamiltonianMC", "NUTS", "quad_potential", "QuadPotentialDiag",
    "QuadPotentialFull", "QuadPotentialFullInv", "QuadPotentialDiagAdapt", "QuadPotentialFullAdapt",
    "Engine", "targets", "diagnostics", "distributed",
]
