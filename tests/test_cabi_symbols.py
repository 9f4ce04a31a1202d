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


_C_TO_RUST = {"int": "c_int", "void": "c_void", "char": "c_char", "uint8_t": "u8", "int16_t": "i16", "uint16_t": "u16",
              "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32"}


def _c_type_to_rust(c: str) -> str:
    """`const int16_t *` -> `*const i16`, `bd_detector **` -> `*mut *mut bd_detector`, `uint32_t` -> `u32`."""
    c = c.replace("volatile", " ").strip()
    stars = c.count("*")
    base = c.replace("*", " ").split()
    const = "const" in base
    base = [w for w in base if w not in ("const", "struct")]
    assert len(base) == 1, c
    t = _C_TO_RUST.get(base[0], base[0])
    for level in range(stars):
        # only the innermost pointee can be const in the header's declarations
        t = ("*const " if (const and level == 0) else "*mut ") + t
    return t


def _header_functions():
    """{name: (return type, [argument types])} of the header, in Rust spelling."""
    text = open(os.path.join(ROOT, "include", "beat_detector_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \t\*]*?)\b(bd_[a-z0-9_]+)\s*\(([^;{}()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        arg_types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (the last identifier)
                a = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", a).strip()
                arg_types.append(_c_type_to_rust(a))
        out[name] = ("" if ret == "void" else _c_type_to_rust(ret), arg_types)
    return out


def _rust_functions():
    rs = open(os.path.join(ROOT, "hip", "src", "ffi.rs")).read()
    rs = re.sub(r"//.*$", "", rs, flags=re.M)
    out = {}
    for m in re.finditer(r"pub fn (bd_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", rs, re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip()
        types = [a.split(":", 1)[1].strip() for a in args.split(",") if ":" in a]
        out[name] = (ret, types)
    return out


def test_rust_declarations_equal_the_header(bd):
    """hip/src/ffi.rs cannot be compiled here (no Rust toolchain), so every extern declaration is compared
    with the header: same set of functions, same number of arguments, same C types in the same order, same
    return type; #[repr(C)] structs list the header's fields in order."""
    from beat_detector_amd.api import EXPORTED_SYMBOLS
    header = _header_functions()
    rust = _rust_functions()
    assert sorted(header) == _declared_functions()
    assert sorted(rust) == sorted(header), set(rust) ^ set(header)
    assert sorted(rust) == sorted(EXPORTED_SYMBOLS)
    for name, (ret, args) in header.items():
        r_ret, r_args = rust[name]
        assert r_ret == ret, (name, r_ret, ret)
        assert r_args == args, (name, r_args, args)
    rs = open(os.path.join(ROOT, "hip", "src", "ffi.rs")).read()
    htext = open(os.path.join(ROOT, "include", "beat_detector_hip.h")).read()
    for struct in ("bd_params", "bd_onset_params", "bd_biquad_params", "bd_synth_source", "bd_input_source", "bd_biquad_design"):
        rs_fields = re.findall(r"pub (\w+):", re.search(r"pub struct %s \{(.*?)\n\}" % struct, rs, re.S).group(1))
        h_body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s;" % (struct, struct), htext, re.S).group(1)
        h_body = re.sub(r"/\*.*?\*/", "", h_body, flags=re.S)
        h_fields = []
        for decl in h_body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fp = re.search(r"\(\s*\*\s*(\w+)\s*\)", decl)        # function pointer member
            if fp:
                h_fields.append(fp.group(1))
                continue
            decl = re.sub(r"\[[^\]]*\]", "", decl)                  # array extents
            names = decl.split(None, 1)[1] if " " in decl else decl
            names = re.sub(r"^(const\s+)?\w+\s+", "", decl) if "," not in decl else decl.split(None, 1)[1]
            for nm in names.split(","):
                h_fields.append(nm.replace("*", "").strip().split()[-1])
        assert rs_fields == h_fields, (struct, rs_fields, h_fields)


def test_rust_crate_mirrors_the_reference_interface():
    """The safe layer names what the parent crate's public interface names (src/lib.rs:51-185, src/record.rs:39-44)."""
    lib_rs = open(os.path.join(ROOT, "hip", "src", "lib.rs")).read()
    rec_rs = open(os.path.join(ROOT, "hip", "src", "record.rs")).read()
    for needle in ("pub struct BeatInfo", "pub const fn relative_ms(&self) -> u32", "pub trait Strategy",
                   "fn is_beat(&self, samples: &[i16]) -> Option<BeatInfo>;", "fn kind(&self) -> StrategyKind;",
                   "fn min_duration_between_beats_ms() -> u32", "pub enum StrategyKind", "#[non_exhaustive]",
                   "pub fn detector(&self, sampling_rate: u32) -> Box<dyn Strategy + Send>",
                   "pub fn name(&self) -> &'static str", "pub fn description(&self) -> &'static str",
                   "pub fn values() -> Vec<Self>", "vec![Self::LPF, Self::Spectrum]",
                   "impl Strategy for HipLpfDetector", "impl Strategy for HipSpectrumDetector"):
        assert needle in lib_rs, needle
    assert "unimplemented!" not in lib_rs and "todo!" not in lib_rs
    for needle in ("pub fn start_listening", "on_beat_cb: impl Fn(BeatInfo) + Send + 'static", "strategy: StrategyKind",
                   "keep_recording: Arc<AtomicBool>", "-> Result<JoinHandle<()>, String>",
                   "Variable keep_recording is false from the beginning!?", "Must have input device!",
                   "pub fn u16_data_to_i16(data: &[u16]) -> Vec<i16>", "pub fn f32_data_to_i16(data: &[f32]) -> Vec<i16>",
                   "ffi::bd_start_listening", "ffi::bd_listener_join"):
        assert needle in rec_rs, needle
