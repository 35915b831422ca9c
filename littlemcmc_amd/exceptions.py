"""Exception types of the package (reference: /root/reference/littlemcmc/exceptions.py:22 for SamplingError)."""
from ._abi import HipLibraryError  # noqa: F401  (no-GPU / no-library failures are loud, never a fallback)


class SamplingError(RuntimeError):
    """Raised when sampling cannot proceed."""


__all__ = ["SamplingError", "HipLibraryError"]
