"""The exp table used by both the oracle and the HIP engine is glibc 2.35's __exp_data.tab (N = 128).

It was read out of this image's /lib/x86_64-linux-gnu/libm.so.6 (the struct is located by its first member
invln2N = 0x1.71547652b82fep+7; the 256-word table follows 0x70 bytes later) and is validated against libm's exp by
oracle/check_powf.c.  The two copies (oracle/exp_table.inc, pednstream_amd/csrc/exp_table.inc) are data, kept separate so
that the product never includes anything from oracle/.  Re-run only on a machine with the same glibc:"""
import re
import struct

data = open("/lib/x86_64-linux-gnu/libm.so.6", "rb").read()
inv = struct.pack("<d", float.fromhex("0x1.71547652b82fep+7"))
base = [m.start() for m in re.finditer(re.escape(inv), data)][0]
tab = struct.unpack("<256Q", data[base + 0x70:base + 0x70 + 2048])
assert tab[1] == 0x3ff0000000000000
lines = ["/* glibc 2.35 __exp_data.tab (sysdeps/ieee754/dbl-64/e_exp_data.c, N = 128): {tail bits, scale bits - (k << 45)} */"]
lines += ["{0x%016xULL, 0x%016xULL}," % (tab[2 * k], tab[2 * k + 1]) for k in range(128)]
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("oracle", os.path.join("pednstream_amd", "csrc")):
    with open(os.path.join(root, d, "exp_table.inc"), "w") as f:
        f.write("\n".join(lines) + "\n")
print("wrote exp_table.inc (x2)")
