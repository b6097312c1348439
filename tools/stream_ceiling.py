"""Practical HBM ceiling for the engine's access shapes: a fully coalesced a[i] + b[i] -> out[i] over 2^26 doubles
(1 GiB read, 0.5 GiB written, beyond the 256 MiB MALL) with 8-byte lanes (op 7, the shape of the f64 history rows in
node_kernel) and 16-byte lanes (op 8, the shape of link_kernel).  Run under `rocprofv3 --kernel-trace --stats` and divide
1.5 GiB by the device_math_kernel durations, or read the wall times printed here (they include the host copies)."""
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
for op in (7, 8, 7, 8, 7, 8):
    rc = engine.lib().pedn_device_math(0, op, n, a.ctypes.data_as(P), b.ctypes.data_as(P), 0, out.ctypes.data_as(P))
    assert rc == 0 and out[12345] == 3.0 and out[-1] == 3.0
# op 9: the same bytes in chunks of 256 B ... 32 KB visited in a scrambled order (the history rows of one wave are 512-byte
# segments scattered over the arrays): measured 5.1-5.5 TB/s for every chunk size, i.e. the segment size is not a limit
for chunk in (32, 64, 128, 256, 512, 4096):
    rc = engine.lib().pedn_device_math(0, 9, n, a.ctypes.data_as(P), b.ctypes.data_as(P), chunk, out.ctypes.data_as(P))
    assert rc == 0 and (out == 3.0).all()
print("done")
