"""Host mirror of the reference's Candle wrappers: argument checks and error strings (the ones the
reference's tests assert, csrc/tests/cache_manager_tests.rs:189-191,457,476,492,508,524, plus the
bail! texts of csrc/src/lib.rs and models/src/flash_attention.rs).  All checks fire before any
launch, so this runs without a GPU."""
import ctypes as C

import pytest

import atoma_hip as ah
from atoma_hip import F16, BF16, F32, I64, U32, tensor, ref, tensor_array

FAKE = 0x1000  # never dereferenced: every case fails its checks first


def err(rc):
    assert rc != 0
    return ah.last_error()


def test_copy_blocks_errors():
    """cache_manager_tests.rs:456-538."""
    kc = [tensor(FAKE, (4, 64, 2, 8), F16) for _ in range(2)]
    vc = [tensor(FAKE, (4, 64, 2, 8), F16) for _ in range(1)]
    mp = tensor(FAKE, (2, 2), I64)
    assert err(ah.lib.atoma_copy_blocks(tensor_array(kc), 2, tensor_array(vc), 1, ref(mp))) == \
        "key_caches and value_caches must have the same length"
    cpu = [tensor(FAKE, (4, 64, 2, 8), F16, device=-1) for _ in range(2)]
    assert err(ah.lib.atoma_copy_blocks(tensor_array(cpu), 2, tensor_array(cpu), 2, ref(mp))) == "device must be a cuda device"
    vb = [tensor(FAKE, (4, 64, 2, 8), BF16) for _ in range(2)]
    assert err(ah.lib.atoma_copy_blocks(tensor_array(kc), 2, tensor_array(vb), 2, ref(mp))) == \
        "Only support f16/bf16 dtypes and src and dst must have same dtype"
    f32 = [tensor(FAKE, (4, 64, 2, 8), F32) for _ in range(2)]
    assert err(ah.lib.atoma_copy_blocks(tensor_array(f32), 2, tensor_array(f32), 2, ref(mp))) == \
        "Only support f16/bf16 dtypes and src and dst must have same dtype"
    vc2 = [tensor(FAKE, (4, 64, 2, 8), F16) for _ in range(2)]
    bad = tensor(FAKE, (2, 3), I64)
    assert err(ah.lib.atoma_copy_blocks(tensor_array(kc), 2, tensor_array(vc2), 2, ref(bad))) == \
        "block_mapping must have shape [num_pairs, 2]"


def test_swap_blocks_errors():
    """cache_manager_tests.rs:188-202: cpu <-> cpu is rejected with this exact text."""
    a = tensor(FAKE, (3, 16, 2, 8), F16, device=-1)
    b = tensor(FAKE, (3, 16, 2, 8), F16, device=-1)
    pairs = (C.c_uint32 * 2)(0, 1)
    assert err(ah.lib.atoma_swap_blocks_tensor(ref(a), ref(b), pairs, 1)) == \
        ("swap_blocks: Either src and dst are on the same cuda device, or src and dst are on cpu and cuda devices, "
         "alternately")
    g0, g1 = tensor(FAKE, (3, 16, 2, 8), F16, device=0), tensor(FAKE, (3, 16, 2, 8), F16, device=1)
    assert err(ah.lib.atoma_swap_blocks_tensor(ref(g0), ref(g1), pairs, 1)) == \
        "swap_blocks: Both src and dst tensors should be on the same device to swap"


def test_reshape_and_cache_flash_errors():
    """cache_manager_tests.rs:686-769: dtype, shape and slot_mapping mismatches are errors."""
    key, val = tensor(FAKE, (10, 4, 64), F32), tensor(FAKE, (10, 4, 64), F32)
    kc, vc = tensor(FAKE, (2, 4, 64, 8), F32), tensor(FAKE, (2, 4, 64, 8), F32)
    sm = tensor(FAKE, (10,), I64)
    assert "Only support f16/bf16 dtypes" in err(ah.lib.atoma_reshape_and_cache_flash(ref(key), ref(val), ref(kc), ref(vc), ref(sm)))
    key, val = tensor(FAKE, (10, 4, 64), F16), tensor(FAKE, (10, 4, 64), F16)
    kc, vc = tensor(FAKE, (2, 4, 64, 8), F16), tensor(FAKE, (2, 4, 64, 7), F16)
    assert "Only support" in err(ah.lib.atoma_reshape_and_cache_flash(ref(key), ref(val), ref(kc), ref(vc), ref(sm)))
    kc, vc = tensor(FAKE, (2, 8, 4, 64), F16), tensor(FAKE, (2, 8, 4, 64), F16)
    sm9 = tensor(FAKE, (9,), I64)
    assert err(ah.lib.atoma_reshape_and_cache_flash(ref(key), ref(val), ref(kc), ref(vc), ref(sm9))) == \
        "Only support slot_mapping with shape [10] (got [9])"


def test_attention_wrapper_errors():
    """csrc/src/lib.rs bail! texts."""
    q = tensor(FAKE, (1, 4, 7, 64), F16)
    k = tensor(FAKE, (1, 4, 4, 64), F16)
    o = tensor(FAKE, (1, 4, 7, 64), F16)
    assert err(ah.lib.atoma_flash_attn(ref(q), ref(k), ref(k), 1.0, 0, ref(o))) == \
        "number of k/v heads 4 must divide number of heads in query 7"
    kb = tensor(FAKE, (1, 4, 4, 64), BF16)
    assert err(ah.lib.atoma_flash_attn(ref(q), ref(kb), ref(kb), 1.0, 0, ref(o))) == "query and key must have the same dtype"
    q12 = tensor(FAKE, (1, 4, 4, 12), F16)
    assert err(ah.lib.atoma_flash_attn(ref(q12), ref(q12), ref(q12), 1.0, 0, ref(q12))) == \
        "only supports head sizes that are a multiple of 8 (got 12)"
    q264 = tensor(FAKE, (1, 4, 4, 264), F16)
    assert err(ah.lib.atoma_flash_attn(ref(q264), ref(q264), ref(q264), 1.0, 0, ref(q264))) == \
        "only supports head dimension at most 256 (got 264)"
    qc = tensor(FAKE, (1, 4, 4, 64), F16, device=-1)
    assert err(ah.lib.atoma_flash_attn(ref(qc), ref(qc), ref(qc), 1.0, 0, ref(qc))) == "no cpu support for flash-attn"
    # paged entry points: page size must be a multiple of 16 (lib.rs:778-785, 1604-1608)
    q3 = tensor(FAKE, (8, 4, 64), F16)
    kc = tensor(FAKE, (4, 8, 4, 64), F16)
    cu = tensor(FAKE, (3,), U32)
    bt = tensor(FAKE, (2, 2), U32)
    assert err(ah.lib.atoma_flash_attn_varlen_with_block_table(ref(q3), ref(kc), ref(kc), None, ref(cu), ref(cu), 4, 16, 1.0,
                                                               -1, -1, ref(bt), ref(q3))) == \
        "page_block_size must be a multiple of 16, got 8"
    q4 = tensor(FAKE, (2, 1, 4, 64), F16)
    assert err(ah.lib.atoma_flash_attn_kv_cache_full(ref(q4), ref(kc), ref(kc), None, 1.0, ref(bt), None, 1, ref(q4))) == \
        "page_block_size must be a multiple of 16 when block_table is provided"
    bt3 = tensor(FAKE, (3, 2), U32)
    kc16 = tensor(FAKE, (4, 16, 4, 64), F16)
    assert "shape mismatch of block_table" in err(
        ah.lib.atoma_flash_attn_kv_cache_full(ref(q4), ref(kc16), ref(kc16), None, 1.0, ref(bt3), None, 1, ref(q4)))


def test_flash_attention_new_checks():
    """models/src/flash_attention.rs:474-546 (test_new*, test_supported_head_sizes)."""
    fa = ah.FlashAttention()
    assert ah.lib.atoma_flash_attention_new(C.byref(fa), 8, 4, 64, 1.0, None, -1, F32, 0) == 0
    assert (fa.num_heads, fa.num_kv_heads, fa.head_dim) == (8, 4, 64) and fa.softmax_scale == 1.0
    assert ah.lib.atoma_flash_attention_new(C.byref(fa), 7, 4, 64, 1.0, None, -1, F32, 0) != 0
    assert "must divide" in ah.last_error()
    assert err(ah.lib.atoma_flash_attention_new(C.byref(fa), 8, 4, 65, 1.0, None, -1, F32, 0)) == "head_dim 65 is not supported"
    for d in (64, 80, 96, 112, 128, 192, 256):
        assert ah.lib.atoma_flash_attention_new(C.byref(fa), 8, 4, d, 1.0, None, -1, F16, 0) == 0


def test_forward_shape_errors():
    """models/src/flash_attention.rs:334-372 and split_kv_cache :247-279 (test_split_kv_cache_invalid_shape)."""
    fa = ah.FlashAttention()
    assert ah.lib.atoma_flash_attention_new(C.byref(fa), 8, 4, 64, 1.0, None, -1, F16, 0) == 0
    q, k = tensor(FAKE, (5, 8, 64), F16), tensor(FAKE, (5, 4, 64), F16)
    meta = ah.AttnMetadata()
    out = tensor(FAKE, (5, 512), F16)
    kv4 = tensor(FAKE, (2, 10, 32, 4), F16)
    assert err(ah.lib.atoma_flash_attention_forward(C.byref(fa), ref(q), ref(k), ref(k), ref(kv4), C.byref(meta), ref(out))) == \
        "KV cache must have rank 5 (got 4)"
    k6 = tensor(FAKE, (6, 4, 64), F16)
    kv = tensor(FAKE, (2, 10, 32, 4, 64), F16)
    assert "must have the same number of tokens (got 5, 6, 6)" in err(
        ah.lib.atoma_flash_attention_forward(C.byref(fa), ref(q), ref(k6), ref(k6), ref(kv), C.byref(meta), ref(out)))
    q7 = tensor(FAKE, (5, 7, 64), F16)
    assert err(ah.lib.atoma_flash_attention_forward(C.byref(fa), ref(q7), ref(k), ref(k), ref(kv), C.byref(meta), ref(out))) == \
        "query must have [num_head, hidden_dim] = [8, 64] (got [7, 64])"
