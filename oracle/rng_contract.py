"""RNG contract -- TEST INFRASTRUCTURE (oracle side), pure Python.

The reference draws from numpy's global MT19937 stream inside the hot path
(/root/reference/src/LTM/link.py:337,343,356,382 and
/root/reference/src/utils/functions.py:133).  A sequential global stream cannot be
reproduced by a parallel kernel, so parity is defined with an injected,
counter-based generator shared by the three implementations:

  * this file           -- monkey-patched into the real reference by oracle/ref_harness.py
  * oracle/pedn_oracle.c -- the C restatement
  * pednstream_amd/csrc  -- the HIP kernels

Contract (all integer / IEEE-754 basic operations, no transcendental functions, no FMA):

  words(seed, replica, link, t, site)[i]  = Philox4x32-10(counter=(t, link, site | (i//4)<<8, replica),
                                                         key=(seed & 0xffffffff, seed >> 32))[i % 4]
  half(...)[i]  = (words[i//2] >> 16*(i%2)) & 0xffff                    # eight 16-bit uniforms per Philox call
  z(...)        = (sum_{i<8} half[i] - 262140) * C,  C = 0x1.3988e1409212ep-16 = sqrt(1.5) * 2^-16
                                                                        # Irwin-Hall(8) from ONE Philox call, exact in binary64
  normal(sigma) = sigma * z
  binomial(n,p) = 0 if n <= 0 or p <= 0;  n if p >= 1
                  #{ i < n : half[i] < floor(p * 2^16) }                 if n <= 16   (one call for n <= 8, two otherwise)
                  clamp(floor(n*p + sqrt((n*p)*(1-p)) * z + 0.5), 0, n)  otherwise

(Contract v2.  v1 summed twelve 32-bit words = three Philox calls per draw; on the GPU the 32x32->64 multiplies of Philox
were the largest ALU cost of busy networks and of the speed-noise draw, so a draw now costs one call.)

Sites: 0 = sending-flow release draw (link.py:337/343), 1 = activity draw (link.py:356),
       2 = receiving-flow reverse-pedestrian draw (link.py:382), 3 = speed noise (functions.py:133).
"""
import math

SITE_RELEASE, SITE_ACTIVITY, SITE_REVERSE, SITE_NOISE = 0, 1, 2, 3

_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """Philox4x32 with 10 rounds (Salmon et al., SC'11).  counter: 4 u32, key: 2 u32."""
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


class Stream:
    """Word stream for one (seed, replica, link, t, site) key."""

    __slots__ = ("key", "t", "link", "site", "replica", "_buf", "_call")

    def __init__(self, seed, replica, link, t, site):
        self.key = (seed & _MASK, (seed >> 32) & _MASK)
        self.t, self.link, self.site, self.replica = t & _MASK, link & _MASK, site, replica & _MASK
        self._buf = ()
        self._call = 0

    def half(self, i):
        w = self.word(i >> 1)
        return (w >> (16 * (i & 1))) & 0xFFFF

    def word(self, i):
        call = i >> 2
        if call != self._call or not self._buf:
            self._buf = philox4x32_10((self.t, self.link, self.site | (call << 8), self.replica), self.key)
            self._call = call
        return self._buf[i & 3]


Z_SCALE = float.fromhex("0x1.3988e1409212ep-16")      # sqrt(1.5) * 2^-16: unit variance for the sum of 8 u16


def z_irwin_hall(stream):
    s = 0
    for i in range(8):
        s += stream.half(i)
    return float(s - 262140) * Z_SCALE


def normal(sigma, stream):
    return float(sigma) * z_irwin_hall(stream)


def binomial(n, p, stream):
    n = int(n)
    p = float(p)
    if n < 0:
        raise ValueError("n < 0")
    if n == 0 or p <= 0.0:
        return 0
    if p >= 1.0:
        return n
    if n <= 16:
        thr = int(math.floor(p * 65536.0))
        return sum(1 for i in range(n) if stream.half(i) < thr)
    mean = n * p
    sd = math.sqrt(mean * (1.0 - p))
    k = math.floor(mean + sd * z_irwin_hall(stream) + 0.5)
    return int(min(max(k, 0), n))


def binomial_meanfield(n, p):
    """RNG-free debugging mode: binomial -> floor(n*p), normal -> 0."""
    n = int(n)
    if n <= 0:
        return 0
    return int(math.floor(n * float(p)))
