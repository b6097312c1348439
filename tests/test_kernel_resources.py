"""Register budget of the step kernels, read from the code object inside the built libpedn_hip.so (no GPU needed).

node_kernel is compiled for 8 waves per SIMD and sits at 63 of its 64 vector registers: small edits have flipped the allocator
from a few scalar spills (into VGPR lanes) to 16 vector spills into scratch, which costs +11 us per launch on
melbourne x 1024; and scalar spills that find no VGPR lane left go to scratch too.  Builds like that pass
every parity test, so they are caught here: no vector spill and NO scratch (private_segment_fixed_size) in any step kernel."""
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
    # one YAML list entry per kernel ("  - .agpr_count: ..."), keys in alphabetical order: .group_segment_fixed_size comes before .name
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"(\s*)(- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        if m.group(2) and len(m.group(1)) <= 2:        # a new kernel entry (nested "- .offset" items of .args are indented deeper)
            cur = {}
        key, val = m.group(3), m.group(4).strip().strip("'")
        if cur is None:
            continue
        if key == "name" and "symbol" not in cur and not line.startswith(" " * 8):
            kernels[val] = cur
        elif key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
            cur[key] = int(val)
    # scratch ACCESSES per kernel, from the disassembly: private_segment_fixed_size alone also counts frame slots the compiler reserved
    # and never touches (a spilled 8-register tuple whose spill was then lowered to VGPR lanes leaves its 32-byte slot behind)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    cur_sym = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur_sym = m.group(1)
            if cur_sym in kernels:
                kernels[cur_sym]["scratch_instructions"] = 0
        elif cur_sym in kernels and ("scratch_" in line or re.search(r"buffer_(load|store).*s\[0:3\]", line)):
            kernels[cur_sym]["scratch_instructions"] += 1
    demangle = shutil.which("c++filt")
    if demangle:
        names = list(kernels)
        out = subprocess.run([demangle], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        kernels = {re.sub(r"\(.*", "", d).replace("void ", ""): kernels[n] for n, d in zip(names, out)}
    return kernels


# Every instantiation the benchmarked configurations launch (melbourne / delft x 1024 plain stepping, the batched RL step on
# 45_intersections x 2048 with shared and with per-replica randomised scenarios, eager and clocked), with the budget each one is built for:
#   pattern -> (max VGPRs = waves per SIMD it must keep, max LDS bytes = workgroups per CU, max scalar spills (None: any),
#               max scratch INSTRUCTIONS in the kernel's code)
# Template arguments of node_kernel: <PR, LP, HIST, MD, LU, TF, CLK> (pedn_kernels.hpp: node_kernel_waves gives the register budget).
# Scalar spills go to VGPR lanes (v_writelane / v_readlane).  What must never happen in a step kernel is a scratch ACCESS: checked on
# the disassembly for every kernel below (node_kernel<LU> ships with private_segment_fixed_size 36 -- the slot of a spilled
# 8-register tuple whose spill was lowered to VGPR lanes afterwards; the kernel contains no scratch instruction, which is what this test
# now pins instead of trusting the metadata either way).
BUDGETS = {
    # shared link parameters, classic node model, at most 6 corridors per node: 8 waves per SIMD
    r"node_kernel<false, false, (true|false), 6, false, false, false>": (64, None, None, 0),
    # the same with the link update of the previous step performed by the slot waves (pedn_run's owner-wave plan, the headline kernel)
    r"node_kernel<false, false, (true|false), 6, true, false, false>": (64, None, None, 0),
    # a node of 7 or 8 corridors (loops unrolled for 8): 6 waves per SIMD
    r"node_kernel<false, false, (true|false), 8, (true|false), false, false>": (80, None, None, 0),
    # per-replica link parameters (randomised RL resets, ensembles): 6 waves per SIMD, no spill of either kind without the link update
    r"node_kernel<true, false, (true|false), \d, false, false, (true|false)>": (80, None, 0, 0),
    # ... with it (randomised ensembles of a model without device-computed rows): one 8-byte value goes through scratch
    r"node_kernel<true, false, (true|false), \d, true, false, false>": (80, None, None, 6),
    # the clocked env step (step index from the device clock): 6 waves per SIMD, no spill
    r"node_kernel<false, false, (true|false), 6, false, false, true>": (80, None, 0, 0),
    r"node_kernel<false, false, (true|false), 8, false, false, true>": (80, None, None, 6),
    # the single-launch plan of small batches (the slot waves compute their own rows of turning fractions): one block per CU, no scratch
    r"node_kernel<(true|false), false, (true|false), \d, true, true, false>": (256, None, None, 0),
    # ... with helper waves: sixteen waves per workgroup = 4 per SIMD, no scratch
    r"node_kernel_h<(true|false), (true|false), \d>": (128, None, None, 0),
    # the node LP (assign_flows_type 'optimal'): 6 waves per SIMD
    r"node_kernel<(true|false), true, (true|false), \d, false, false, false>": (80, None, None, 0),
    # second launch of a step with dynamic turning fractions and / or observations: 4 waves per SIMD, 4 workgroups per CU by LDS
    # (the OBS instantiation was at 131-133 VGPRs / 42.5 KB = 3 until the parts shared one LDS buffer)
    r"link_turn_kernel<(true|false), (true|false), (true|false), (true|false)>": (128, 40960, None, 0),
    r"turn_frac_kernel<(true|false), (true|false)>": (128, 40960, None, 0),
    r"link_kernel_1r<(true|false), (true|false)>": (64, None, 0, 0),    # one replica per lane: the link update as a launch of its own
    r"rl_observe_kernel<(true|false)>": (64, 8192, 0, 0),
    r"rl_apply_kernel": (64, None, 0, 0),
}


def test_hot_kernels_have_no_vector_spills_and_keep_their_occupancy(tmp_path):
    k = kernel_metadata(tmp_path)
    assert any(n.startswith("node_kernel<") for n in k), sorted(k)[:5]
    seen = {pat: 0 for pat in BUDGETS}
    for name, r in k.items():
        for pat, (max_vgpr, max_lds, max_sspill, max_scratch) in BUDGETS.items():
            if re.fullmatch(pat, name):
                seen[pat] += 1
                assert r["vgpr_spill_count"] == 0 or max_scratch > 0, (name, r)      # (the few rare instantiations allowed a scratch access)
                assert r["vgpr_count"] <= max_vgpr, (name, r)
                if max_lds is not None:
                    assert r["group_segment_fixed_size"] <= max_lds, (name, r)
                if max_sspill is not None:
                    assert r["sgpr_spill_count"] <= max_sspill, (name, r)
                assert r.get("scratch_instructions", 0) <= max_scratch, (name, r)
                assert r.get("private_segment_fixed_size", 0) <= 64, (name, r)
    assert all(seen.values()), {p: n for p, n in seen.items() if n == 0}      # every pattern still names a kernel of the library
