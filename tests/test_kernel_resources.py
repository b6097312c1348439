"""Register budget of the step kernels, read from the code object inside the built libpedn_hip.so (no GPU needed).

node_kernel is compiled for 8 waves per SIMD and sits at 63 of its 64 vector registers: small edits have flipped the allocator
from a few scalar spills (into VGPR lanes, cheap) to 16 vector spills into scratch, which costs +11 us per launch on
melbourne x 1024.  A build like that passes every parity test, so it is caught here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pednstream_amd", "csrc", "libpedn_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_metadata(tmp_path):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools) or not os.path.exists(LIB):
        pytest.skip("ROCm LLVM tools or the built library are not here")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([tools[0], "--dump-section", f".hip_fatbin={fat}", LIB], check=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'")
        if key == "name":
            cur = kernels.setdefault(val, {})
        elif cur is not None and key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "sgpr_count", "group_segment_fixed_size"):
            cur[key] = int(val)
    demangle = shutil.which("c++filt")
    if demangle:
        names = list(kernels)
        out = subprocess.run([demangle], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        kernels = {re.sub(r"\(.*", "", d).replace("void ", ""): kernels[n] for n, d in zip(names, out)}
    return kernels


def test_hot_kernels_have_no_vector_spills_and_keep_their_occupancy(tmp_path):
    k = kernel_metadata(tmp_path)
    assert any(n.startswith("node_kernel<") for n in k), sorted(k)[:5]
    for name, r in k.items():
        # the instantiations the benchmarked configurations run: shared link parameters, classic node model, 8 waves per SIMD
        if re.match(r"node_kernel<false, 8, false, (true|false), (true|false), \d>", name):
            assert r["vgpr_spill_count"] == 0, (name, r)
            assert r["vgpr_count"] <= 64, (name, r)          # 8 waves per SIMD
        if re.match(r"link_kernel<1, (true|false)>", name) or name.startswith("link_kernel_pr<") or name.startswith("rl_observe_kernel<"):
            assert r["vgpr_spill_count"] == 0, (name, r)
            assert r["vgpr_count"] <= 72, (name, r)          # 7 waves per SIMD
        if name.startswith("link_turn_kernel<") or name.startswith("turn_frac_kernel<"):
            assert r["vgpr_spill_count"] == 0, (name, r)
            assert r["vgpr_count"] <= 168, (name, r)         # 3 waves per SIMD at least
