"""kernels.knobs(): ONE table of every switch (Python globals, environment variables, native tuning keys) -- complete and at its
defaults in a fresh process.  No GPU."""
import importlib
import inspect
import os
import re


def test_every_setter_and_environment_variable_is_in_the_table():
    from stgraph_amd import _C, kernels
    table = kernels.knobs()
    names = {n for n, v in table.items() if isinstance(v, dict) and "default" in v}
    for v in table.values():
        if isinstance(v, dict) and "default" in v:
            assert v["value"] == v["default"], v                      # nothing is switched at import
    modules = {v["module"] for v in table.values() if isinstance(v, dict) and "module" in v}
    for module in sorted(modules | {"stgraph_amd.kernels", "stgraph_amd.nn.functional", "stgraph_amd.temporal"}):
        mod = importlib.import_module(module)
        for fn, obj in inspect.getmembers(mod, inspect.isfunction):
            if fn.startswith("set_") and obj.__module__ == module and fn not in ("set_tuning",):
                assert fn[4:] in names, f"{module}.{fn} has no row in kernels._KNOBS"
    # every STGRAPH_AMD_* variable the package reads is listed
    root = os.path.dirname(os.path.abspath(kernels.__file__))
    seen = set()
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                seen |= set(re.findall(r"STGRAPH_AMD_[A-Z0-9_]+", open(os.path.join(dirpath, f)).read()))
    assert seen <= set(kernels.ENVIRONMENT), seen - set(kernels.ENVIRONMENT)
    assert table["library"]["abi"] == _C.ABI_VERSION == _C.lib.stg_abi_version()


def test_native_keys_match_the_library():
    """Every key of _C.TUNING_KEYS is accepted by stg_set_tuning (value 0 = auto), an unknown or retired one is refused."""
    import pytest
    from stgraph_amd import _C
    for k in _C.TUNING_KEYS:
        _C.set_tuning(k, 0)
    for k in ("step_impl", "step_fold", "no_such_knob"):
        with pytest.raises(_C.StgError):
            _C.set_tuning(k, 0)
    src = open(os.path.join(os.path.dirname(os.path.abspath(_C.__file__)), "csrc", "stg_common.hip")).read()
    assert set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', src)) == set(_C.TUNING_KEYS)
