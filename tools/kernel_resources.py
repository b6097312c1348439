#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in a built engine library, read from its gfx950 code object (no GPU needed).

    python tools/kernel_resources.py [path/to/libpedn_hip.so] [name-filter]

Columns: VGPRs, AGPRs, SGPRs, scalar spills, vector spills, scratch bytes per lane (private_segment_fixed_size), static LDS bytes."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def kernel_metadata(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"(\s*)(- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        if m.group(2) and len(m.group(1)) <= 2:
            cur = {}
        key, val = m.group(3), m.group(4).strip().strip("'")
        if cur is None:
            continue
        if key == "name" and "symbol" not in cur and not line.startswith(" " * 8):
            kernels[val] = cur
        elif key in KEYS:
            cur[key] = int(val)
    names = list(kernels)
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", d).replace("void ", ""): kernels[n] for n, d in zip(names, out)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "pednstream_amd", "csrc", "libpedn_hip.so")
    filt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    k = kernel_metadata(lib)
    print(f"{len(k)} kernels in {lib}")
    print(f"{'kernel':70s} vgpr agpr sgpr s_spill v_spill scratch   lds")
    for name in sorted(k):
        if filt in name:
            r = k[name]
            print(f"{name:70s} " + " ".join(f"{r.get(key, 0):{w}d}" for key, w in zip(KEYS, (4, 4, 4, 7, 7, 7, 5))))
