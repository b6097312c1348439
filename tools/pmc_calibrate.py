"""Known-byte-count streaming launch for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: other access widths than 16 B/lane are uncalibrated).
device_math_kernel op 7 reads two arrays of n doubles and writes one, 8 bytes per lane, fully coalesced --
the access shape of the engine's f64 history rows.  n = 2^26 -> 1 GiB read, 0.5 GiB written (beyond the 256 MiB MALL)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pednstream_amd import engine  # noqa: E402

n = 1 << 26
a = np.ones(n)
b = np.full(n, 2.0)
out = np.empty(n)
P = C.POINTER(C.c_double)
rc = engine.lib().pedn_device_math(0, 7, n, a.ctypes.data_as(P), b.ctypes.data_as(P), 0, out.ctypes.data_as(P))
assert rc == 0 and out[12345] == 3.0
print(f"calibration launch done: read {2 * n * 8} B, wrote {n * 8} B")
