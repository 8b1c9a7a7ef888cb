"""SURVEY 8.f3 (the Rust-side shim): there is no Rust toolchain in the image, so the `-sys` crate under integration/ cannot be
compiled here.  What CAN be checked is that it says what the header says: generated from include/atoma_hip.h, up to date,
one `pub fn` per exported symbol with the C prototype's argument count, the reference's four FFI names present, and the
#[repr(C)] structs field for field."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as G  # noqa: E402

RS = os.path.join(ROOT, "integration", "atoma-hip-sys", "src", "lib.rs")
LIB = os.path.join(ROOT, "atoma-infer_amd", "lib", "libatoma_hip.so")


def test_generated_crate_is_up_to_date():
    text, _ = G.generate()
    assert open(RS).read() == text, "integration/atoma-hip-sys/src/lib.rs is stale: run python tools/gen_rust_ffi.py"


def test_one_rust_declaration_per_exported_symbol_with_the_same_arity():
    _, funcs = G.generate()
    rs = open(RS).read()
    lib = C.CDLL(LIB)
    assert len(funcs) >= 60
    for ret, name, params in funcs:
        assert hasattr(lib, name), f"{name} is declared but not exported"
        m = re.findall(rf"pub fn {name}\((.*?)\)(?: -> ([^;]+))?;", rs)
        assert len(m) == 1, name
        args = [a for a in m[0][0].split(", ") if a]
        assert len(args) == len(params), (name, len(args), len(params))
        assert (m[0][1] == "") == (ret == "void"), name
    for n in ("run_mha", "copy_blocks_f16", "copy_blocks_bf16", "reshape_and_cache_flash"):   # csrc/src/ffi.rs:3-102
        assert f"pub fn {n}(" in rs
    # run_mha: 47 arguments, as counted against ffi.rs in round 1
    assert [len(p) for _, n, p in funcs if n == "run_mha"] == [47]


def test_type_mapping_and_struct_layouts():
    consts, structs, _ = G.parse(open(G.HEADER).read())
    assert dict(consts)["ATOMA_BF16"] == 1 and dict(consts)["ATOMA_SWAP_GPU_TO_CPU"] == 2
    rt = lambda t: G.rust_type(t, structs)
    assert rt("const void *") == "*const c_void" and rt("void *") == "*mut c_void"
    assert rt("const void *const *") == "*const *const c_void" and rt("void *const *") == "*const *mut c_void"
    assert rt("const int64_t *") == "*const i64" and rt("float *") == "*mut f32" and rt("const atoma_tensor *") == "*const atoma_tensor"
    assert rt("uint32_t") == "u32" and rt("bool") == "bool" and rt("const char *") == "*const c_char"
    hdr = structs["atoma_kv_block_header"]
    size = sum({"char": 1, "uint32_t": 4, "uint64_t": 8, "uint8_t": 1}[t.strip()] * (n or 1) for t, _, n in hdr)
    assert size == 128                                      # the header comment's "128 bytes": no padding between the fields
    assert [f for _, f, _ in structs["atoma_seq_desc"]] == ["is_prompt", "no_block_tables", "length", "num_computed_tokens", "token_chunk_size",
                                                            "token_ids", "block_table", "block_table_len"]
    assert "(this: *mut atoma_flash_attention, " in open(RS).read()   # `self` is a Rust keyword: renamed
