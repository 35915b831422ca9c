"""Plain-numpy restatement of littlemcmc_amd/diagnostics.py (split R-hat, Geyer multi-chain ESS).

TEST INFRASTRUCTURE ONLY. The reference has no diagnostics: parity UNPINNED against
the reference; this file pins the device/torch implementation against a direct, loop-level statement of the
Stan reference-manual formulas."""
import numpy as np


def split(x):
    c, n, d = x.shape
    h = n // 2
    return np.concatenate([x[:, :h], x[:, n - h:]], axis=0)


def rank_normalize(x):
    """Vehtari et al. (2021) eq. 14: z = Phi^-1((r - 3/8) / (S + 1/4)), r = rank among ALL S draws of the dimension."""
    from scipy.stats import norm, rankdata

    c, n, d = x.shape
    z = np.empty_like(x, dtype="d")
    for j in range(d):
        r = rankdata(x[:, :, j].ravel(), method="average")
        z[:, :, j] = norm.ppf((r - 0.375) / (c * n + 0.25)).reshape(c, n)
    return z


def rhat_ess(x, do_split=True, rank_normalized=False):
    if rank_normalized:
        x = rank_normalize(x)
    if do_split:
        x = split(x)
    m, n, d = x.shape
    rhat = np.zeros(d)
    ess = np.zeros(d)
    for j in range(d):
        y = x[:, :, j]
        means = y.mean(axis=1)
        acov = np.zeros((m, n))
        for c in range(m):
            z = y[c] - means[c]
            for t in range(n):
                acov[c, t] = np.dot(z[: n - t], z[t:]) / n
        chain_var = acov[:, 0] * n / (n - 1.0)
        w = chain_var.mean()
        b_over_n = means.var(ddof=1) if m > 1 else 0.0
        var_plus = w * (n - 1.0) / n + b_over_n
        rhat[j] = np.sqrt(var_plus / w)
        rho = 1.0 - (w - acov.mean(axis=0) * n / (n - 1.0)) / var_plus
        rho[0] = 1.0
        tau = -1.0
        prev = np.inf
        t = 0
        while t + 1 < n:
            p = rho[t] + rho[t + 1]
            if p <= 0:
                break
            p = min(p, prev)
            prev = p
            tau += 2.0 * p
            t += 2
        tau = max(tau, 1.0 / np.log10(max(m * n, 10.0)))
        ess[j] = m * n / tau
    return rhat, ess


def torch_chain_stats(x, t0, n, lag0, lags_per_pass=16):
    """Restatement of ONE pass of the device kernel ``lmc_diag_chain_stats`` (littlemcmc_amd/csrc/lmc_diag.hip) for CPU
    tensors, by FFT: out[0] = sum of chain means, out[1] = sum of squared chain means, out[2] = sum of unbiased
    within-chain variances (lag0 == 0 only), out[3 + k] = sum of biased autocovariances at lag lag0 + k of the
    sub-series [t0, t0 + n) of every chain of x[chains, draws, d].

    TEST INFRASTRUCTURE ONLY: the CPU tests of the multi-rank reduction logic (gloo, world_size 2) inject it through
    ``diagnostics.summarize(..., stats_fn=torch_chain_stats)``; the -m gpu tests compare the HIP kernel with it."""
    import torch

    blk = x[:, t0:t0 + n].to(torch.float64)
    d = blk.shape[2]
    out = torch.zeros((3 + lags_per_pass, d), dtype=torch.float64, device=x.device)
    if blk.shape[0] == 0:
        return out
    mean = blk.mean(dim=1)
    cen = blk - mean[:, None, :]
    nfft = 1 << (2 * n - 1).bit_length()
    f = torch.fft.rfft(cen, n=nfft, dim=1)
    acov = torch.fft.irfft(f.real ** 2 + f.imag ** 2, n=nfft, dim=1)[:, :n] / n       # [c, n, d], biased
    out[0] = mean.sum(dim=0)
    out[1] = (mean ** 2).sum(dim=0)
    if lag0 == 0:
        out[2] = acov[:, 0].sum(dim=0) * (n / (n - 1.0))
    hi = min(lag0 + lags_per_pass, n)
    if hi > lag0:
        out[3:3 + hi - lag0] = acov[:, lag0:hi].sum(dim=0)
    return out
