"""The C-ABI shared library loads (no GPU needed) and exports exactly the symbols include/nkbhip.h declares."""
import ctypes
import re
from pathlib import Path

from nkb_classification import hip

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "nkbhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nkb_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_header_symbols():
    lib = hip.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nkbhip.h but not exported by libnkbhip.so"
    assert sorted(hip.exported_symbols()) == declared, "ctypes binding table and header disagree"
    assert lib.nkb_version() >= 100
    assert lib.nkb_last_error() is not None
    assert lib.nkb_conv_gemm_stat_tiles(1, 802816, 64) == 3136 and lib.nkb_conv_gemm_stat_tiles(0, 12544, 2048) == 98
    assert lib.nkb_kernel_name(0) == b"conv_igemm_fwd"


def test_argument_validation_without_gpu():
    """Entry points validate geometry on the host before any launch (error text through nkb_last_error)."""
    lib = hip.load()
    rc = lib.nkb_conv_gemm(1, 0, None, None, None, None, None, None, 1, 8, 8, 48, 48, 8, 8, 64, 64, 0, 1, 1, 1, 0, 0, 0, 0, 0, None, None)
    assert rc != 0 and b"Cin=48" in lib.nkb_last_error()
    rc = lib.nkb_conv_gemm(1, 0, None, None, None, None, None, None, 1, 8, 8, 64, 64, 8, 8, 64, 64, 0, 3, 3, 3, 1, 0, 0, 0, 0, None, None)
    assert rc != 0 and b"stride" in lib.nkb_last_error()
    rc = lib.nkb_bn_apply(1, None, None, None, None, None, 10, 12, 0, None, None, None, None)
    assert rc != 0 and b"C=12" in lib.nkb_last_error()


def test_binding_arity_matches_header():
    """Every ctypes signature in hip._SIGS has as many arguments as the prototype in include/nkbhip.h (a changed entry point
    whose binding was not updated would otherwise corrupt the call's argument registers silently)."""
    text = (ROOT / "include" / "nkbhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = dict(re.findall(r"\b(nkb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert len(protos) >= 50
    for name, (_, argtypes) in hip._SIGS.items():
        params = protos[name].strip()
        n = 0 if params in ("", "void") else len([a for a in params.split(",") if a.strip()])
        assert n == len(argtypes), f"{name}: header has {n} parameters, binding {len(argtypes)}"


def test_plan_dispatch_table_is_current_and_validates_entries():
    """csrc/plan_dispatch.inc (the typed call table behind nkb_plan_run) is what scripts/gen_plan_dispatch.py generates from the
    binding's signature table; the library's table agrees with the binding name by name; a table walk on the host rejects a
    bad function id / argument count before calling anything, validates an entry's geometry through the entry point itself and
    reports the failing index."""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_plan_dispatch.py"), "--check"]).returncode == 0, \
        "plan_dispatch.inc is stale: run python scripts/gen_plan_dispatch.py and rebuild"
    lib = hip.load()
    ids = hip._plan_fn_ids()
    recordable = {n for n, (res, _) in hip._SIGS.items() if res is ctypes.c_int and n not in hip._PURE}
    assert set(ids) == recordable and lib.nkb_plan_fn_count() == len(ids)
    failed = ctypes.c_int(-1)
    assert lib.nkb_plan_run(None, 0, ctypes.byref(failed)) == 0
    tab = (hip._PlanEntry * 3)()
    # entry 0: a well-formed nkb_bn_apply whose C = 12 the entry point itself refuses (host-side validation, no launch)
    tab[0].fn, tab[0].nargs = ids["nkb_bn_apply"], len(hip._SIGS["nkb_bn_apply"][1])
    tab[0].a[0].i, tab[0].a[6].i, tab[0].a[7].i = 1, 10, 12
    assert lib.nkb_plan_run(tab, 1, ctypes.byref(failed)) != 0 and failed.value == 0 and b"C=12" in lib.nkb_last_error()
    tab[0].fn, tab[0].nargs = lib.nkb_plan_fn_count() + 5, 2
    assert lib.nkb_plan_run(tab, 1, ctypes.byref(failed)) != 0 and b"bad function id" in lib.nkb_last_error()
    tab[0].fn, tab[0].nargs = ids["nkb_bn_apply"], 3            # wrong argument count for that function
    assert lib.nkb_plan_run(tab, 1, ctypes.byref(failed)) != 0 and failed.value == 0
    tab[0].fn = -77
    assert lib.nkb_plan_run(tab, 1, ctypes.byref(failed)) != 0 and b"unknown operation" in lib.nkb_last_error()
