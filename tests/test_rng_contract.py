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


def test_exp_is_within_one_ulp_of_numpy():
    L = od.lib()
    xs = np.linspace(-40, 3, 20001)
    mine = np.array([L.pedn_oracle_exp(float(x)) for x in xs])
    ref = np.exp(xs)
    assert np.all(np.abs(mine - ref) <= np.spacing(ref))
    assert L.pedn_oracle_exp(0.0) == 1.0
