"""Exceptions (/root/reference/littlemcmc/exceptions.py:22)."""

__all__ = ["SamplingError"]


class SamplingError(RuntimeError):
    """Error while sampling."""
