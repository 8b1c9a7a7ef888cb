"""Shared helpers for the parity tests (numpy side)."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle.halfs import F16, BF16, from_f32, to_f32  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_half(rng, shape, dtype, scale=1.0):
    """N(0, scale) rounded to the storage dtype, as uint16 bit patterns."""
    return from_f32((rng.standard_normal(shape) * scale).astype(np.float32), dtype)


def ulp_tol(ref_f32, dtype, atol=1e-3):
    """|got - ref| <= atol + eps*|ref|, eps = one unit in the last place of the storage dtype
    (2^-7 bf16, 2^-10 f16): the '1e-3 bf16 tolerance' of BASELINE.json made scale-aware
    (SURVEY 8c)."""
    eps = 2.0 ** -7 if dtype == BF16 else 2.0 ** -10
    return atol + eps * np.abs(ref_f32)


# Absolute tolerance of an attention output against the fa_acausal (pure f32) oracle.
# BASELINE.json asks for 1e-3 (plus one unit in the last place of the storage dtype).
# That holds whenever the softmax mass is spread over many keys.  The reference's CUDA kernel --
# and ours -- rounds P to the storage dtype before P.V (softmax.h + the bf16 MMA): every term
# moves by up to 2^-9 (bf16) / 2^-12 (f16) relative, so when a row sees only a few keys (p close
# to 1) the result moves by up to 2^-9 * |v| ~ 4e-3 for N(0,1) values.  The same bound applies
# between two kernels that round P against different running maxima (split-KV, per-lane-group
# partial softmax), so the kernel-faithful oracle mode is checked with the same tolerance.
ATOL_FEW_KEYS = {BF16: 4e-3, F16: 1e-3}
ATOL_VS_F32 = ATOL_FEW_KEYS            # name used by tests that mix short and long rows


def attn_atol(dtype, visible_keys):
    """1e-3 once a row attends over >= 512 keys (the BASELINE configs use 2048-4096), the
    P-rounding bound below that."""
    return 1e-3 if visible_keys >= 512 else ATOL_FEW_KEYS[dtype]


def assert_close(got_bits, ref_bits, dtype, atol=1e-3, what=""):
    got, ref = to_f32(got_bits, dtype), to_f32(ref_bits, dtype)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite output"
    err = np.abs(got - ref)
    tol = ulp_tol(ref, dtype, atol)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} beyond tolerance; max err {err.max():.3e} "
                           f"at {np.unravel_index(err.argmax(), err.shape)} (ref {ref.flat[err.argmax()]:.5f})")


def poison_unwritten_slots(kc, vc, bt, lens, pattern):
    """The slots of every sequence's last page behind its length were never written by reshape_and_cache: whatever they hold must
    not reach an output (the reference zero-fills out-of-range V rows, flash_fwd_kernel.h:903).  Fills them -- and every page no
    sequence owns -- with `pattern` (a NaN code of the cache's element type); returns poisoned copies."""
    page = kc.shape[1]
    kc2, vc2 = kc.copy(), vc.copy()
    owned = set()
    for b, L in enumerate(lens):
        n = (int(L) + page - 1) // page
        owned.update(int(x) for x in bt[b, :n])
        if n and int(L) % page:
            kc2[bt[b, n - 1], int(L) % page:] = pattern
            vc2[bt[b, n - 1], int(L) % page:] = pattern
    for pg in range(kc.shape[0]):
        if pg not in owned:
            kc2[pg] = pattern
            vc2[pg] = pattern
    return kc2, vc2


_oracle_lib = None


def oracle_c():
    """ctypes handle of oracle/_build/liboracle.so (built on demand with gcc)."""
    global _oracle_lib
    if _oracle_lib is None:
        so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        src = os.path.join(ROOT, "oracle", "c", "oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        _oracle_lib = C.CDLL(so)
    return _oracle_lib


def c_attention(q, k, v, *, b, h, h_k, d, seqlen_q, seqlen_k, scale, is_bf16, q_strides, k_strides, v_strides,
                o_shape, o_strides, causal=0, cu_q=None, cu_k=None, k_cumulative=True, block_table=None,
                page=0, threads=0):
    """oracle_attention() of oracle/c/oracle.c on numpy uint16 arrays; strides = (batch,row,head)."""
    lib = oracle_c()
    o = np.zeros(o_shape, np.uint16)
    P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    cu_q = None if cu_q is None else np.ascontiguousarray(cu_q, np.int32)
    cu_k = None if cu_k is None else np.ascontiguousarray(cu_k, np.int32)
    bt = None if block_table is None else np.ascontiguousarray(block_table, np.int32)
    i64 = C.c_int64
    lib.oracle_attention.argtypes = [C.c_void_p] * 6 + [C.c_int] + [i64] * 12 + [C.c_int] * 4 + [C.c_float,
                                     C.c_void_p, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.oracle_attention(P(q), P(k), P(v), P(o), P(cu_q), P(cu_k), int(k_cumulative),
                         q_strides[0], k_strides[0], v_strides[0], o_strides[0],
                         q_strides[1], k_strides[1], v_strides[1], o_strides[1],
                         q_strides[2], k_strides[2], v_strides[2], o_strides[2],
                         b, h, h_k, d, scale, P(bt), 0 if bt is None else bt.shape[1], page,
                         seqlen_q, seqlen_k, int(is_bf16), int(causal), threads)
    return o


def make_paged_cache(rng, num_blocks, page, h_k, d, dtype, lens, shuffle=True):
    """K,V caches [nb,page,hk,d] filled with N(0,1) and a block table [B,max_blocks] whose rows
    own disjoint, randomly placed physical pages (unused entries = 0, as the worker pads them:
    backends/vllm/src/worker.rs:410-412)."""
    B = len(lens)
    need = [(int(L) + page - 1) // page for L in lens]
    max_blocks = max(1, max(need))
    assert sum(need) <= num_blocks
    perm = rng.permutation(num_blocks) if shuffle else np.arange(num_blocks)
    bt = np.zeros((B, max_blocks), np.int32)
    pos = 0
    for i, n in enumerate(need):
        bt[i, :n] = perm[pos:pos + n]
        pos += n
    kc = rand_half(rng, (num_blocks, page, h_k, d), dtype)
    vc = rand_half(rng, (num_blocks, page, h_k, d), dtype)
    return kc, vc, bt
