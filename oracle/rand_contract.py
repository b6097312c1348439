"""Contract of the DEVICE randomiser -- TEST INFRASTRUCTURE, pure Python.

`pedn_randomize_scenarios` / `pedn_draw_demand` (include/pedn.h) draw, for every replica at once and on the device, what the reference's
`NetworkEnvGenerator.randomize_network` draws per env on the host from numpy's global stream
(/root/reference/src/utils/env_loader.py:160-181: `generate_random_link_params` :363-424, `generate_random_od_flows` :224-259,
`generate_random_demand_params` :183-222, and the series of /root/reference/src/LTM/od_manager.py:92-155).  A sequential numpy stream
cannot be replayed by a kernel, so the device has a contract of its own -- the same DISTRIBUTIONS (tests/test_gpu_scenarios.py checks them
against a numpy generator), its own Philox-keyed numbers.  This file restates that contract step by step so that the kernels
(pednstream_amd/csrc/pedn_kernels.hpp: rand_links_kernel, rand_od_kernel, rand_demand_kernel, draw_demand_kernel) are pinned BIT FOR BIT:

  key            = (seed & 0xffffffff, seed >> 32); g = replica_offset + replica (the GLOBAL replica id)
  u01(w)         = w * 2^-32
  link params    corridor p = 0 .. P-1 in order: c = Philox(p, 0x61, 0, g); taken so far = n; k = int(P * link_fraction)
                 chosen = c0 * (P - p) < (k - n) << 32          (selection sampling: every k-subset equally likely, exactly k chosen)
                 density part  = chosen and (c1 & 0xffff) < 0x8000      speed part = chosen and (c1 >> 16) < 0x8000
                 F = 0.6 + (1.2 - 0.6) * u01(c2)                G = 0.6 + (0.9 - 0.6) * u01(c3)
                 both links of the corridor from their OWN base parameters: k_c = max(0.5, k_c * F), k_j = max(2 k_c, k_j * F); v_f = v_f * G
  OD weights     w[od] = 1 + (10 - 1) * u01(Philox(od, 0x63, 0, g)[0])
  demand params  a = Philox(node, 0x51, 0, g), b = Philox(node, 0x52, 0, g): pattern = (a0 * 3) >> 32 (0 gaussian_peaks, 1 constant,
                 2 sudden_demand); base = 2 + 8 u01(a1); peak = max(10 + 20 u01(a2), base + 5); spike length = 10 + ((a3 * 10) >> 32);
                 spike start = (b0 * max(T - length, 1)) >> 32; spike height = 20 + ((b1 * 30) >> 32)
  demand series  constant: base for every t <= T.  Otherwise, t < T: lambda = base + peak * (exp(-(t - T/4)^2 / (2 (T/20)^2)) +
                 exp(-(t - 3T/4)^2 / (2 (T/20)^2))) [glibc exp, summed left to right]; Poisson(lambda) by inversion with
                 u = ((c0 << 32 | c1) >> 11) * 2^-53, c = Philox(t, node, 0x50, g): the smallest k with u <= sum_{j<=k} p_j,
                 p_0 = exp(-lambda), p_j = p_{j-1} * (lambda / j), k <= 1000; sudden_demand adds the height inside its spike; t = T: 0
"""
import math

from rng_contract import philox4x32_10

_MASK = 0xFFFFFFFF


def _key(seed):
    return (seed & _MASK, (seed >> 32) & _MASK)


def _u01(w):
    return w * 2.0 ** -32


def link_params(seed, g, corridors, base, link_fraction=0.2):
    """corridors: [(link a, link b)] in corridor order; base: {link: (k_critical, k_jam, free_flow_speed)} -> {link: (kc, kj, vf)}."""
    P = len(corridors)
    k = int(P * link_fraction)
    taken, out = 0, {}
    for p, ab in enumerate(corridors):
        c = philox4x32_10((p, 0x61, 0, g & _MASK), _key(seed))
        chosen = c[0] * (P - p) < ((k - taken) << 32)
        taken += 1 if chosen else 0
        dens = chosen and (c[1] & 0xFFFF) < 0x8000
        spd = chosen and (c[1] >> 16) < 0x8000
        F = 0.6 + (1.2 - 0.6) * _u01(c[2])
        G = 0.6 + (0.9 - 0.6) * _u01(c[3])
        for l in ab:
            kc, kj, vf = base[l]
            if dens:
                kc = max(0.5, base[l][0] * F)
                kj = max(kc * 2.0, base[l][1] * F)
            if spd:
                vf = base[l][2] * G
            out[l] = (kc, kj, vf)
    return out


def od_weight(seed, g, od):
    return 1.0 + (10.0 - 1.0) * _u01(philox4x32_10((od, 0x63, 0, g & _MASK), _key(seed))[0])


def demand_params(seed, g, node, T):
    a = philox4x32_10((node, 0x51, 0, g & _MASK), _key(seed))
    b = philox4x32_10((node, 0x52, 0, g & _MASK), _key(seed))
    pattern = (a[0] * 3) >> 32
    base = 2.0 + 8.0 * _u01(a[1])
    peak = 10.0 + 20.0 * _u01(a[2])
    if peak < base + 5.0:
        peak = base + 5.0
    length = 10 + ((a[3] * 10) >> 32)
    span = T - length if T - length > 1 else 1
    start = (b[0] * span) >> 32
    height = float(20 + ((b[1] * 30) >> 32))
    return pattern, base, peak, start, length, height


def demand_series(seed, g, node, T, pattern, base, peak, start=0, length=0, height=0.0):
    """The T + 1 entries of the origin's demand row (pedn_draw_demand with explicit parameters, or those of demand_params)."""
    key = _key(seed)
    out = []
    w = T / 20.0
    for t in range(T + 1):
        val = 0.0
        if pattern == 1:
            val = base
        elif t < T:
            x, y = t - T / 4.0, t - 3.0 * T / 4.0
            lam = base + peak * math.exp(-(x * x) / (2.0 * w * w)) + peak * math.exp(-(y * y) / (2.0 * w * w))
            c = philox4x32_10((t, node, 0x50, g & _MASK), key)
            u = float(((c[0] << 32) | c[1]) >> 11) * 2.0 ** -53
            p = math.exp(-lam)
            cdf, k = p, 0
            while u > cdf and k < 1000:
                k += 1
                p *= lam / float(k)
                cdf += p
            val = float(k)
            if pattern == 2 and start <= t < start + length:
                val += height
        out.append(val)
    return out
