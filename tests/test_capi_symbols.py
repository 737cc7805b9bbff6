"""The C-ABI library loads without a GPU and exports every symbol include/stgraph_hip.h declares."""
import ctypes
import os
import re

import pytest

from tests.util import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "stgraph_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ("stg_gcn_agg", "stg_gat_fwd_k0", "stg_gat_fwd_k1", "stg_gat_bwd", "stg_gat_bwd_er",
                 "stg_graph_build_device", "stg_graph_build_host", "stg_csr_ctor_host", "stg_last_error_string"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from stgraph_amd import _C
    lib = ctypes.CDLL(_C.LIB_PATH)
    declared = _declared()
    assert sorted(_C.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert _C.lib.stg_abi_version() == _C.ABI_VERSION


def test_error_convention_without_gpu():
    from stgraph_amd import _C
    assert _C.lib.stg_set_tuning(b"no_such_key", 1) == _C.STG_ERR_INVALID_ARGUMENT
    assert b"unknown key" in _C.lib.stg_last_error_string()
    with pytest.raises(_C.StgError) as ei:
        _C.set_tuning("gcn_unroll", 3)
    assert ei.value.code == _C.STG_ERR_INVALID_ARGUMENT
    # argument validation happens on the host, before any launch
    rc = _C.lib.stg_gcn_agg(None, None, None, None, None, None, None, None, None, 4, 0, 0, None)
    assert rc == _C.STG_ERR_INVALID_ARGUMENT
    assert _C.lib.stg_graph_build_device_workspace_bytes(1000, 100) > 0


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under stgraph_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "stgraph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "stg_oracle" not in text and "oracle/" not in text and "import oracle" not in text, f
