"""The C-ABI library loads without a GPU and exports every symbol include/pedn.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from pednstream_amd import engine

    return engine


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pedn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pedn_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    names = declared_functions()
    assert "pedn_step" in names and "pedn_rl_step" in names and len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/pedn.h but not exported"
    assert sorted(built.EXPORTS) == names, "engine.EXPORTS out of sync with the header"
    assert lib.pedn_abi_version() == built.ABI_VERSION == 4


def test_ctypes_struct_matches_header_layout(built):
    """ModelDesc / RlDesc mirror the C structs: compile a tiny C program printing sizeof/offsetof and compare."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "pedn.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(pedn_model_desc), offsetof(pedn_model_desc, dt), offsetof(pedn_model_desc, link_tt0),
         offsetof(pedn_model_desc, pf_temp), offsetof(pedn_model_desc, pair_upod), sizeof(pedn_rl_desc));
  return 0;
}'''
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        c = os.path.join(tmp, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(tmp, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, c])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    M, R = built.ModelDesc, built.RlDesc
    want = [ctypes.sizeof(M), M.dt.offset, M.link_tt0.offset, M.pf_temp.offset, M.pair_upod.offset, ctypes.sizeof(R)]
    assert got == want


def test_engine_refuses_to_run_without_a_gpu(built):
    """No CPU fallback: on a machine without a HIP device pedn_create must fail loudly (skipped where a GPU exists)."""
    import numpy as np

    from golden_util import Golden, build_network

    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    net = build_network(Golden("six_node_full"))
    with pytest.raises(RuntimeError):
        net.network_loading(1)
    assert np.asarray(net.nodes[1].demand).sum() > 0        # host-side objects stay usable
