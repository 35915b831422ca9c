"""Host-only record types for sampler warnings.

Mirrors the *interface* of /root/reference/littlemcmc/report.py:20-37 (field names of ``SamplerWarning`` and the
member names / values of ``WarningType`` are part of the API surface: ``step.warnings()`` returns them)."""
import collections
import enum

_WARNING_FIELDS = ("kind", "message", "level", "step", "exec_info", "extra")
SamplerWarning = collections.namedtuple("SamplerWarning", _WARNING_FIELDS)

# member order fixes the values 1..8, as in the reference
_WARNING_KINDS = ("DIVERGENCE TUNING_DIVERGENCE DIVERGENCES TREEDEPTH "   # HMC / NUTS
                  "BAD_PARAMS CONVERGENCE BAD_ACCEPTANCE BAD_ENERGY")       # sampler parameters / convergence
WarningType = enum.unique(enum.Enum("WarningType", _WARNING_KINDS.split(), start=1, module=__name__))
