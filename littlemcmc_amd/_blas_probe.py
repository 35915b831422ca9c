"""Which float32 dot-product rounding does THIS host's numpy use?

The reference's start-state kinetic energy is ``0.5 * p.dot(v)`` on float32 arrays
(/root/reference/littlemcmc/quadpotential.py:210-214 via integration.py:63-64), i.e. the host BLAS's
``sdot``. The device reproduces OpenBLAS's two x86-64 summation orders (csrc/lmc_sampler.hpp:
sdot_openblas); this module decides which one the local numpy matches so that a run on this box
tracks the reference *as it would run on this box*. Pure host logic, a few dozen tiny dot products at
import time of an Engine; falls back to the SkylakeX order (what AVX-512 hosts use).
"""
import numpy as np

from . import _abi

_f32 = np.float32
_cached = None


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(_f32)


def _emulate(x, y, mode):
    n = len(x)
    n1 = n & ~31
    simd = _f32(0)
    if n1:
        if mode == _abi.SDOT_OPENBLAS_SKYLAKEX:
            n64 = n1 & ~63
            acc = np.zeros(64, _f32)
            for b in range(0, n64, 64):
                acc = _fma(x[b:b + 64], y[b:b + 64], acc)
            a = np.stack([acc[16 * k:16 * k + 8] + acc[16 * k + 8:16 * k + 16] for k in range(4)]).astype(_f32)
            if n1 > n64:
                for k in range(4):
                    a[k] = _fma(x[n64 + 8 * k:n64 + 8 * k + 8], y[n64 + 8 * k:n64 + 8 * k + 8], a[k])
            s = ((a[0] + a[1]).astype(_f32) + a[2]).astype(_f32) + a[3]
            h = (s[:4] + s[4:]).astype(_f32)
        else:
            acc = np.zeros(32, _f32)
            for b in range(0, n1, 32):
                acc = _fma(x[b:b + 32], y[b:b + 32], acc)
            hk = np.stack([acc[8 * k:8 * k + 4] + acc[8 * k + 4:8 * k + 8] for k in range(4)]).astype(_f32)
            h = ((hk[0] + hk[1]).astype(_f32) + (hk[2] + hk[3]).astype(_f32)).astype(_f32)
        simd = _f32(_f32(h[0] + h[1]) + _f32(h[2] + h[3]))
    tail = np.float64(0)
    for i in range(n1, n):
        tail = tail + np.float64(_f32(x[i] * y[i]))
    return _f32(tail + np.float64(simd))


def emulate_sdot(x, y, mode):
    """numpy emulation of the device's sdot_openblas (used by the probe and by tests)."""
    return _emulate(np.asarray(x, _f32), np.asarray(y, _f32), mode)


def detect_sdot_mode():
    """LMC_SDOT_* constant whose rounding matches ``np.dot`` on float32 vectors on this host."""
    global _cached
    if _cached is not None:
        return _cached
    rs = np.random.RandomState(12345)
    score = {_abi.SDOT_OPENBLAS_SKYLAKEX: 0, _abi.SDOT_OPENBLAS_HASWELL: 0}
    trials = 0
    for n in (96, 128, 200, 75):
        for _ in range(6):
            x = rs.randn(n).astype(_f32)
            y = (x * (0.5 + rs.rand(n))).astype(_f32)
            ref = np.dot(x, y)
            trials += 1
            for mode in score:
                score[mode] += int(_emulate(x, y, mode) == ref)
    best = max(score, key=score.get)
    _cached = best if score[best] >= 0.9 * trials else _abi.SDOT_OPENBLAS_SKYLAKEX
    return _cached
