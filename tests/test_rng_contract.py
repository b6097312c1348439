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


def test_device_randomiser_contract_draws_the_reference_ranges():
    """oracle/rand_contract.py (the CPU restatement the device randomiser is pinned against, tests/test_gpu_scenarios.py) on its own:
    exactly int(P * fraction) corridors per replica, both links of a chosen corridor scaled from their own base values within the
    reference's ranges and floors (env_loader.py:363-424), OD weights in [1, 10) (:224-259), demand parameters in their ranges (:183-222),
    a constant pattern's series constant, the others non-negative integers that stop at T."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import rand_contract as rc

    P = 37
    corridors = [(2 * p, 2 * p + 1) for p in range(P)]
    base = {l: (2.0 + 0.1 * (l % 3), 6.0 + 0.5 * (l % 2), 1.1 + 0.01 * l) for l in range(2 * P)}
    seen = np.zeros(P, dtype=int)
    for g in range(200):
        got = rc.link_params(99, g, corridors, base)
        touched = 0
        for p, (a, b) in enumerate(corridors):
            for l in (a, b):
                kc, kj, vf = got[l]
                assert 0.6 * base[l][2] <= vf <= base[l][2] and kj >= 2.0 * kc - 1e-12 and kc >= 0.5
                assert max(0.5, 0.6 * base[l][0]) - 1e-12 <= kc <= 1.2 * base[l][0] + 1e-12
            # a corridor's two links share the decision, and a chosen corridor may still draw "neither part" (1/4 of the time)
            assert (got[a] != base[a]) == (got[b] != base[b])
            touched += got[a] != base[a]
            seen[p] += got[a] != base[a]
        assert touched <= int(P * 0.2)
    assert seen.min() > 0 and seen.max() < 80                      # every corridor is picked sometimes, none always
    w = [rc.od_weight(5, g, od) for g in range(50) for od in range(6)]
    assert 1.0 <= min(w) and max(w) < 10.0 and len(set(w)) == len(w)
    T, patterns = 500, set()
    for g in range(60):
        pat, lo, pk, start, length, height = rc.demand_params(7, g, 12, T)
        patterns.add(pat)
        assert 2.0 <= lo < 10.0 and pk >= lo + 5.0 and 10 <= length < 20 and 0 <= start < T - length and 20.0 <= height < 50.0
        s = rc.demand_series(7, g, 12, T, pat, lo, pk, start, length, height)
        assert len(s) == T + 1 and min(s) >= 0.0
        if pat == 1:
            assert set(s) == {lo}
        else:
            assert s[T] == 0.0 and all(x == int(x) for x in s) and 0.3 * lo * T < sum(s) < (lo + 2 * pk + 60) * T
    assert patterns == {0, 1, 2}
