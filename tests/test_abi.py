"""The C-ABI library loads (without a GPU) and exports every symbol that
include/dagcon.h declares; no compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dagcon.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(dagcon_[a-z_0-9]+)\s*\(", src)
    return sorted(set(names))


def test_header_and_binding_agree():
    from pbdagcon_amd import capi
    assert sorted(capi.EXPORTS) == declared_functions()


def test_library_exports_every_declared_symbol():
    from pbdagcon_amd import capi
    lib = capi.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/dagcon.h but not exported"
    assert lib.dagcon_abi_version() == 2


def test_struct_layouts_match_header():
    """sizes the C compiler gives the ABI structs == the ctypes mirrors."""
    import subprocess
    import tempfile
    from pbdagcon_amd import capi
    prog = r'''
#include <stdio.h>
#include "dagcon.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu\n", sizeof(dagcon_opts), sizeof(dagcon_batch),
 sizeof(dagcon_results), sizeof(dagcon_timings), sizeof(dagcon_graph_dump), sizeof(dagcon_pre_batch)); return 0;}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "s"),
                               os.path.join(d, "s.c")])
        out = subprocess.check_output([os.path.join(d, "s")]).split()
    got = [ctypes.sizeof(x) for x in (capi.Opts, capi.Batch, capi.Results, capi.Timings, capi.GraphDump, capi.PreBatch)]
    assert got == [int(x) for x in out]


def test_default_opts_are_pbdagcon_defaults():
    """main.cpp:181-211: -c 6 -m 500 -t 50; min_weight follows min_cov (quirk Q1)."""
    from pbdagcon_amd import capi
    o = capi.default_opts()
    assert (o.min_cov, o.min_len, o.trim, o.min_weight) == (6, 500, 50, -1)


def test_no_device_fails_loudly():
    """Without a GPU the product refuses to run: there is no CPU fallback."""
    import torch
    from pbdagcon_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.DagconError) as e:
        capi.Context()
    assert e.value.code == -2


def test_product_does_not_import_oracle():
    """The product package never references oracle/ (checker only)."""
    pkg = os.path.join(ROOT, "pbdagcon_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn
                assert "dagcon_oracle" not in txt and "liboracle" not in txt, fn
