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
