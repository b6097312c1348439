"""RNG contract: Philox known answers, and Python (oracle/rng_contract.py, injected into the reference) ==
C (oracle/pedn_oracle.c) for binomial / normal draws."""
import ctypes as C
import os
import sys

import numpy as np

import oracle_driver as od

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import rng_contract as rc  # noqa: E402

# Random123 known-answer vectors for philox4x32-10
KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers_python_and_c():
    L = od.lib()
    for ctr, key, want in KAT:
        assert rc.philox4x32_10(ctr, key) == want
        buf = (C.c_uint32 * 4)(*ctr)
        L.pedn_oracle_philox(buf, key[0], key[1])
        assert tuple(buf) == want


def test_binomial_and_normal_python_equals_c():
    L = od.lib()
    rng = np.random.default_rng(5)
    for _ in range(3000):
        n = int(rng.choice([0, 1, 2, 5, 15, 16, 17, 40, 333, 40000]))
        p = float(np.float32(rng.random())) if rng.random() < 0.8 else float(rng.choice([0.0, 1.0, 0.9, 0.5]))
        seed, rep, link, t, site = int(rng.integers(0, 2**40)), int(rng.integers(0, 4096)), int(rng.integers(0, 2000)), int(rng.integers(0, 700)), int(rng.integers(0, 3))
        a = rc.binomial(n, p, rc.Stream(seed, rep, link, t, site))
        b = L.pedn_oracle_binomial(n, p, seed, rep, link, t, site)
        assert a == b, (n, p, a, b)
        assert 0 <= a <= n
        x = rc.normal(0.05, rc.Stream(seed, rep, link, t, rc.SITE_NOISE))
        y = L.pedn_oracle_normal(0.05, seed, rep, link, t)
        assert x == y


def test_binomial_moments_are_sane():
    L = od.lib()
    for n, p in ((12, 0.7), (200, 0.9), (3000, 0.75)):
        draws = np.array([L.pedn_oracle_binomial(n, p, 99, 0, k, 7, 0) for k in range(20000)], dtype=float)
        assert abs(draws.mean() - n * p) < 4 * np.sqrt(n * p * (1 - p) / len(draws)) + 0.5
        assert abs(draws.std() - np.sqrt(n * p * (1 - p))) < 0.05 * np.sqrt(n * p * (1 - p)) + 0.1


def test_powf_restatement_matches_numpy_float32_scalar_power():
    """numpy's float32 scalar power is libm powf (glibc 2.35 here); the restatement must agree bit for bit."""
    L = od.lib()
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.random(20000).astype(np.float32), np.float32([0.0, 1.0, 0.5, 1e-30, 2.0 ** -24])])
    for y in (0.8, 2, 3):
        for x in xs:
            assert np.float32(L.pedn_oracle_powf(float(x), float(np.float32(y)))) == x ** y


def test_exp_restatement_matches_libm_exp():
    """np.exp on a double array reaches libm's exp when numpy's AVX512F loop is disabled (oracle/ref_harness.py); the
    oracle restates glibc 2.35's FMA build of it and must agree bit for bit on this machine's libm."""
    import ctypes
    import ctypes.util

    L = od.lib()
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.exp.restype = ctypes.c_double
    libm.exp.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(2)
    xs = np.concatenate([np.linspace(-40, 3, 20001), -rng.random(20000) * 100, [0.0, -0.0, 1e-300, -1e-20, 1.0, -1.0]])
    mism = sum(1 for x in xs if L.pedn_oracle_exp(float(x)) != libm.exp(float(x)))
    import platform
    if platform.libc_ver()[1] == "2.35":
        assert mism == 0
    else:       # another glibc may round a few arguments differently; the restatement itself is < 1 ulp
        assert all(abs(L.pedn_oracle_exp(float(x)) - np.exp(x)) <= np.spacing(np.exp(x)) for x in xs[::50])
    assert L.pedn_oracle_exp(0.0) == 1.0
