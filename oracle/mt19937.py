"""Restatement of numpy's legacy ``RandomState`` stream (MT19937 + polar gaussian).

TEST INFRASTRUCTURE ONLY (see oracle/README.md). The reference draws every random number
from the global legacy ``np.random`` generator:

* ``np.random.seed(s)``            /root/reference/littlemcmc/sampling.py:133,496-497,576
* ``np.random.randint(2**30)``     /root/reference/littlemcmc/sampling.py:134
* ``np.random.rand(d)``            /root/reference/littlemcmc/sampling.py:584 ; hmc.py:141,166
* ``np.random.uniform()``          /root/reference/littlemcmc/math.py:25
* ``numpy.random.normal(size=d)``  /root/reference/littlemcmc/quadpotential.py:223,376

numpy (third-party dependency of the reference, numpy>=1.17 ``_legacy`` distributions; the
version in this image is 2.2.6) implements these as documented below; this file restates
that published algorithm in plain Python integers so the device RNG
(littlemcmc_amd/csrc/lmc_rng.hpp) can be checked word-for-word, and is itself pinned against
``np.random.RandomState`` in tests/test_oracle_rng.py.
"""
import math

N = 624
M = 397
MATRIX_A = 0x9908B0DF
UPPER = 0x80000000
LOWER = 0x7FFFFFFF
MASK32 = 0xFFFFFFFF


class MT19937:
    """State = 624 words + position + cached gaussian, exactly numpy's legacy layout."""

    def __init__(self, seed=None):
        self.mt = [0] * N
        self.pos = N
        self.has_gauss = 0
        self.gauss = 0.0
        if seed is not None:
            self.seed(seed)

    # np.random.seed(int): init_genrand (Knuth LCG), pos = 624, gaussian cache cleared.
    def seed(self, s):
        s &= MASK32
        mt = self.mt
        mt[0] = s
        for i in range(1, N):
            s = (1812433253 * (s ^ (s >> 30)) + i) & MASK32
            mt[i] = s
        self.pos = N
        self.has_gauss = 0
        self.gauss = 0.0

    def _regen(self):
        mt = self.mt
        for i in range(N - M):
            y = (mt[i] & UPPER) | (mt[i + 1] & LOWER)
            mt[i] = mt[i + M] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
        for i in range(N - M, N - 1):
            y = (mt[i] & UPPER) | (mt[i + 1] & LOWER)
            mt[i] = mt[i + (M - N)] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
        y = (mt[N - 1] & UPPER) | (mt[0] & LOWER)
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
        self.pos = 0

    def u32(self):
        if self.pos == N:
            self._regen()
        y = self.mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & MASK32

    # random_sample / rand / uniform(): 53-bit double from two words.
    def double(self):
        a = self.u32() >> 5
        b = self.u32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    # randint(2**k): masked draw, no rejection for power-of-two ranges.
    def randint_pow2(self, k):
        return self.u32() & ((1 << k) - 1)

    # legacy_gauss: Marsaglia polar method; the second variate is cached ACROSS calls.
    def normal(self):
        if self.has_gauss:
            self.has_gauss = 0
            g = self.gauss
            self.gauss = 0.0
            return g
        while True:
            x1 = 2.0 * self.double() - 1.0
            x2 = 2.0 * self.double() - 1.0
            r2 = x1 * x1 + x2 * x2
            if 0.0 < r2 < 1.0:
                break
        f = math.sqrt(-2.0 * math.log(r2) / r2)
        self.gauss = f * x1
        self.has_gauss = 1
        return f * x2

    def get_state(self):
        """Same tuple layout as ``np.random.RandomState.get_state()`` (legacy=True)."""
        return ("MT19937", list(self.mt), self.pos, self.has_gauss, self.gauss)

    def set_state(self, state):
        _, key, pos, has_gauss, gauss = state
        self.mt = [int(k) for k in key]
        self.pos = int(pos)
        self.has_gauss = int(has_gauss)
        self.gauss = float(gauss)
