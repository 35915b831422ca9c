"""-m gpu: device RNG, leapfrog and adaptation against golden fixtures captured from the reference
and against the oracle. Everything goes through the C ABI (liblmc_hip.so via ctypes)."""
import os

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import targets as T
from oracle.mt19937 import MT19937
from tests._gpu_util import device_target

pytestmark = pytest.mark.gpu


def test_rng_golden_stream(golden_dir):
    g = np.load(os.path.join(golden_dir, "seeds.npz"))
    with lmc.Engine(T.StdNormal(3), chains=2) as eng:
        eng.seed([int(g["stream_seed"])] * 2)
        out = eng.rng_draw([-700, 1001, -1, 10])
        for c in range(2):
            row = out[c]
            np.testing.assert_array_equal(row[:700], g["stream_doubles"])            # bit-exact words
            np.testing.assert_allclose(row[700:1701], g["stream_normals"], rtol=5e-16, atol=0)
            assert row[1701] == float(g["stream_after_uniform"])
            np.testing.assert_allclose(row[1702:1712], g["stream_normals2"], rtol=5e-16, atol=0)
        st = eng.get_rng_state(0)
        assert st[2] == int(g["stream_state_pos"])


@pytest.mark.parametrize("n", [1, 2, 7, 64, 127, 128, 129, 311, 1000])
def test_rng_normals_vs_numpy_many_seeds(n):
    seeds = [0, 1, 42, 4242, 2 ** 30 - 1, 123456789, 77, 99]
    with lmc.Engine(T.StdNormal(3), chains=len(seeds)) as eng:
        eng.seed(seeds)
        out = eng.rng_draw([n, -3, n, n, -1, n])   # repeated calls: cache carry-over for odd n, twists
        for c, s in enumerate(seeds):
            rs = np.random.RandomState(s)
            want = np.concatenate([rs.normal(size=n), rs.random_sample(3), rs.normal(size=n), rs.normal(size=n),
                                   rs.random_sample(1), rs.normal(size=n)])
            np.testing.assert_allclose(out[c], want, rtol=5e-16, atol=0)
            st, ws = eng.get_rng_state(c), rs.get_state()
            assert st[2] == ws[2] and st[3] == ws[3]
            np.testing.assert_array_equal(st[1], ws[1])
            if ws[3]:
                np.testing.assert_allclose(st[4], ws[4], rtol=5e-16)


def test_rng_state_roundtrip_with_numpy_global():
    np.random.seed(2024)
    np.random.normal(size=5)   # leaves a cached gaussian
    with lmc.Engine(T.StdNormal(3), chains=1) as eng:
        eng.set_rng_state(0, np.random.get_state())
        got = eng.rng_draw([6, -2])[0]
        want = np.concatenate([np.random.normal(size=6), np.random.random_sample(2)])
        np.testing.assert_allclose(got, want, rtol=5e-16, atol=0)
        st = eng.get_rng_state(0)
        assert st[2] == np.random.get_state()[2]


def test_mt_seed_matches_oracle():
    with lmc.Engine(T.StdNormal(3), chains=3) as eng:
        eng.seed([5, 6, 2 ** 32 - 1])
        for c, s in enumerate([5, 6, 2 ** 32 - 1]):
            mt = MT19937(s)
            st = eng.get_rng_state(c)
            assert [int(x) for x in st[1]] == mt.mt and st[2] == 624 and st[3] == 0


def test_leapfrog_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "leapfrog.npz"))
    for ci in range(int(g["n_cases"])):
        k = "c%d_" % ci
        d = int(g[k + "d"])
        tgt = device_target(g[k + "family"], d, g[k + "params"])
        adapt = str(g[k + "pot"]) == "adapt"
        with lmc.Engine(tgt, chains=2, potential="diag_adapt" if adapt else "diag", sdot="skylakex") as eng:
            eng.set_potential(np.zeros(d), g[k + "var"].astype("d"), 10.0)
            n, eps = int(g[k + "n"]), float(g[k + "eps"])
            out = eng.trajectory(g[k + "q"][0], g[k + "p"][0], eps, n, n, p0_is_f32=adapt)
            for c in range(2):
                for name in ("q", "p", "v", "g"):
                    np.testing.assert_allclose(out[name][c], g[k + name], rtol=1e-12, atol=1e-14,
                                               err_msg="%s%s" % (k, name))
                # energies: start state is a float32 kinetic energy in the adapt case
                np.testing.assert_allclose(out["energy"][c][1:], g[k + "energy"][1:], rtol=1e-12, atol=1e-12)
                # ... rounded like the capture host's BLAS (sdot_k_SKYLAKEX): agrees to the last float32 bit
                np.testing.assert_allclose(out["energy"][c][0], g[k + "energy"][0], rtol=1e-13, atol=1e-13)
                np.testing.assert_allclose(out["logp"][c], g[k + "logp"], rtol=1e-12, atol=1e-13)
                # reversibility (reference tests/test_hmc.py:23-40, rtol 1e-5)
                np.testing.assert_allclose(out["q"][c][-1], out["q"][c][0], rtol=1e-5, atol=1e-12)
                np.testing.assert_allclose(out["p"][c][-1], out["p"][c][0], rtol=1e-5, atol=1e-12)


def test_logp_dlogp_matches_oracle_targets():
    from oracle import targets as OT

    rs = np.random.RandomState(3)
    for fam, d in [("std_normal", 5), ("std_normal", 130), ("ar1", 2), ("ar1", 65), ("ar1", 200),
                   ("funnel", 3), ("funnel", 256), ("diag_gaussian", 70), ("diag_gaussian", 1000)]:
        f = OT.make(fam, d)
        tgt = device_target(fam, d, f.params())
        q = rs.randn(4, d)
        with lmc.Engine(tgt, chains=4) as eng:
            logp, grad = eng.logp_dlogp(q)
        for c in range(4):
            wl, wg = f(q[c])
            np.testing.assert_allclose(logp[c], wl, rtol=1e-12, atol=1e-12, err_msg=fam)
            np.testing.assert_allclose(grad[c], wg, rtol=1e-13, atol=1e-13, err_msg=fam)
        # reference plug-in signature: callable target object
        l1, g1 = tgt(q[0])
        np.testing.assert_allclose(l1, f(q[0])[0], rtol=1e-12, atol=1e-12)
        assert g1.shape == (d,)


@pytest.mark.parametrize("mode", ["skylakex", "haswell"])
def test_start_energy_float32_blas_order(mode):
    """0.5f * sdot(p, v) of the start state: device == numpy emulation of the OpenBLAS kernel, bit for bit,
    for every length class (pure tail, 32-block, 64-blocks + 32-block + tail)."""
    from littlemcmc_amd import _abi
    from littlemcmc_amd._blas_probe import emulate_sdot

    rs = np.random.RandomState(11)
    code = {"skylakex": _abi.SDOT_OPENBLAS_SKYLAKEX, "haswell": _abi.SDOT_OPENBLAS_HASWELL}[mode]
    for d in [1, 2, 3, 10, 31, 32, 33, 63, 64, 65, 96, 100, 127, 128, 129, 200, 256, 257, 1000, 1024]:
        chains = 3
        with lmc.Engine(T.StdNormal(d), chains=chains, sdot=mode) as eng:
            var = (0.5 + rs.rand(d))
            eng.set_potential(np.zeros(d), var, 10.0)
            p = rs.randn(chains, d).astype(np.float32)
            out = eng.trajectory(np.zeros((chains, d)), p.astype("d"), 0.1, 0, 0, p0_is_f32=True)
            for c in range(chains):
                v = var.astype(np.float32) * p[c]
                want = np.float32(0.5) * emulate_sdot(p[c], v, code)
                assert out["energy"][c, 0] == np.float64(want), (mode, d, out["energy"][c, 0], want)
                np.testing.assert_array_equal(out["v"][c, 0], v.astype("d"))


def test_start_energy_matches_host_numpy_when_detected():
    """With sdot="auto" the device start energy equals this host's float32 numpy dot (the reference's
    arithmetic on this box) whenever the probe recognises the BLAS kernel."""
    from littlemcmc_amd._blas_probe import detect_sdot_mode, emulate_sdot

    mode = detect_sdot_mode()
    rs = np.random.RandomState(5)
    x = rs.randn(200).astype(np.float32)
    y = rs.randn(200).astype(np.float32)
    if emulate_sdot(x, y, mode) != np.dot(x, y):
        pytest.skip("host BLAS sdot order not recognised; device uses the SkylakeX order")
    d = 128
    with lmc.Engine(T.StdNormal(d), chains=4) as eng:
        p = rs.randn(4, d).astype(np.float32)
        out = eng.trajectory(np.zeros((4, d)), p.astype("d"), 0.1, 0, 0, p0_is_f32=True)
        for c in range(4):
            want = 0.5 * p[c].dot(np.ones(d, np.float32) * p[c])
            assert out["energy"][c, 0] == np.float64(want)
