"""Sampler warnings (host-only record types; /root/reference/littlemcmc/report.py:20-37)."""
import enum
from collections import namedtuple

SamplerWarning = namedtuple("SamplerWarning", "kind, message, level, step, exec_info, extra")


@enum.unique
class WarningType(enum.Enum):
    DIVERGENCE = 1
    TUNING_DIVERGENCE = 2
    DIVERGENCES = 3
    TREEDEPTH = 4
    BAD_PARAMS = 5
    CONVERGENCE = 6
    BAD_ACCEPTANCE = 7
    BAD_ENERGY = 8
