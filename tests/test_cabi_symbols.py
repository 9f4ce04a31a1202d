"""The C-ABI library loads on a machine without a GPU and exports every symbol
that include/beat_detector_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "beat_detector_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(bd_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("bd_detector_create", "bd_detector_is_beat", "bd_detector_destroy", "bd_analyze_batch",
                 "bd_analyze_batch_host", "bd_analyze_batch_multi", "bd_kind_name", "bd_kind_description",
                 "bd_kind_min_duration_between_beats_ms", "bd_u16_data_to_i16", "bd_f32_data_to_i16"):
        assert must in names


def test_library_exports_every_declared_symbol(bd):
    L = ctypes.CDLL(bd.lib_path())
    missing = [n for n in _declared_functions() if not hasattr(L, n)]
    assert not missing, missing


def test_python_binding_covers_the_header(bd):
    from beat_detector_amd.api import EXPORTED_SYMBOLS
    assert sorted(EXPORTED_SYMBOLS) == _declared_functions()


def test_library_has_gfx950_code_object(bd):
    data = open(bd.lib_path(), "rb").read()
    assert b"gfx950" in data


def test_rust_bindings_name_only_declared_symbols(bd):
    """hip/src/lib.rs (source only: no Rust toolchain here) must bind symbols the header declares
    and the library exports, and its #[repr(C)] parameter structs must list the header's fields in order."""
    import os, re
    from beat_detector_amd.api import EXPORTED_SYMBOLS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = open(os.path.join(root, "hip", "src", "lib.rs")).read()
    header = open(os.path.join(root, "include", "beat_detector_hip.h")).read()
    bound = set(re.findall(r"pub fn (bd_\w+)\s*\(", rs))
    assert len(bound) >= 15
    for name in bound:
        assert re.search(r"\b%s\s*\(" % name, header), name
        assert name in EXPORTED_SYMBOLS, name
    for struct in ("bd_params", "bd_onset_params", "bd_biquad_params"):
        rs_fields = re.findall(r"pub (\w+):", re.search(r"pub struct %s \{(.*?)\}" % struct, rs, re.S).group(1))
        h_body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s;" % (struct, struct), header, re.S).group(1)
        h_body = re.sub(r"/\*.*?\*/", "", h_body, flags=re.S)
        h_fields = re.findall(r"(\w+)\s*;", h_body)
        assert [f.rstrip("_") for f in rs_fields] == h_fields, struct
