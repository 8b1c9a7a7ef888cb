"""The C-ABI library loads on a CPU-only box and exports every symbol include/atoma_hip.h
declares (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "atoma_hip.h")
LIB = os.path.join(ROOT, "atoma-infer_amd", "lib", "libatoma_hip.so")

# the reference's own FFI (csrc/src/ffi.rs:3-102): these four must exist under these names
REFERENCE_SYMBOLS = ["run_mha", "copy_blocks_f16", "copy_blocks_bf16", "reshape_and_cache_flash"]


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b([A-Za-z_]\w*)\s*\(", text, flags=re.M)
    return sorted({n for n in names if n.startswith("atoma_") or n in REFERENCE_SYMBOLS or n == "run_mha_stream"})


def test_library_is_built():
    assert os.path.exists(LIB), "libatoma_hip.so missing: run __graft_entry__.build()"


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(LIB)
    names = declared_functions()
    assert len(names) >= 20, names
    for n in REFERENCE_SYMBOLS:
        assert n in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/atoma_hip.h but not exported: {missing}"


def test_error_channel_and_host_heuristics_without_gpu():
    lib = C.CDLL(LIB)
    lib.atoma_last_error.restype = C.c_char_p
    assert lib.atoma_last_error() == b""
    lib.atoma_num_splits_heuristic.argtypes = [C.c_int64] * 4
    assert lib.atoma_num_splits_heuristic(48, 108, 64, 128) == 2     # csrc/src/lib.rs:2116-2121 doc example
    lib.atoma_compute_num_splits.argtypes = [C.c_int64] * 5 + [C.c_int]
    assert lib.atoma_compute_num_splits(256, 32, 128, 4096, 1, 256) == 1
    assert lib.atoma_compute_num_splits(1, 32, 128, 4096, 1, 256) > 1
    # swap_blocks with an invalid src/dst device combination: the reference's error string
    lib.atoma_swap_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    assert lib.atoma_swap_blocks(None, None, None, 0, 0, 7, None) != 0
    assert lib.atoma_last_error().startswith(b"swap_blocks: Either src and dst are on the same cuda device")


def test_options_are_known_by_name_without_gpu():
    """atoma_set_option: every documented knob is accepted (host-side state only), an unknown name is an error, not a silent no-op"""
    lib = C.CDLL(LIB)
    lib.atoma_last_error.restype = C.c_char_p
    lib.atoma_set_option.argtypes = [C.c_char_p, C.c_int]
    for name, default in ((b"generic_prefill_tile", 64), (b"generic_prefill_kt", 0), (b"generic_prefill_rq", 0), (b"generic_decode_stream", 2), (b"generic_decode_waves", 0),
                          (b"linear_tile", 1), (b"decode_line_merge", 1)):
        assert lib.atoma_set_option(name, default) == 0, name
    assert lib.atoma_set_option(b"no_such_option", 1) == -1
    assert b"unknown option no_such_option" in lib.atoma_last_error()


def test_static_archive_defines_the_same_symbols():
    """lib/libatoma_hip.a (the reference links `static=flashattention`, csrc/build.rs:105-113): every declared symbol is defined in it"""
    import subprocess
    ar = os.path.join(ROOT, "atoma-infer_amd", "lib", "libatoma_hip.a")
    assert os.path.exists(ar), "libatoma_hip.a missing: make -C atoma-infer_amd"
    out = subprocess.run(["nm", "--defined-only", "-g", ar], capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TtWw"}
    missing = [n for n in declared_functions() if n not in defined]
    assert not missing, missing
