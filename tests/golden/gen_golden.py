"""Generates the committed fixtures under tests/golden/ (run from the repo root:
``python tests/golden/gen_golden.py``).

* reference_tables.npz -- the two golden tables the reference's own tests hold for this path,
  typed from /root/reference/csrc/tests/flash_attn_tests.rs:53-89 (G1, non-causal; also
  :119-135 and :216-233) and /root/reference/models/src/flash_attention.rs:688-704 (G2, causal),
  together with the inputs those tests build (arange(48) in f16 scaled by f16(1/30), f16(1/40),
  f16(1/50) -- Candle's f16 ``tensor / scalar``; softmax scale 0.5).
* oracle_cases.npz -- seeded inputs and the numpy oracle's outputs (f32-mode and kernel-mode)
  for small paged-decode / varlen-prefill / cache-op cases.  The reference itself cannot be
  run anywhere in this environment (Rust + CUDA), so these vectors pin the *oracle*, which in
  turn is pinned by the reference tables above.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import attn_oracle as A, cache_oracle as CO, norm_rope_oracle as NR  # noqa: E402
from oracle.halfs import F16, BF16  # noqa: E402
from util import rand_half, make_paged_cache  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

G1 = np.array([
    [[0.0837, 0.1038, 0.1238, 0.1438, 0.1637, 0.1837, 0.2037, 0.2238],
     [0.0922, 0.1122, 0.1322, 0.1522, 0.1721, 0.1921, 0.2122, 0.2322]],
    [[0.4204, 0.4404, 0.4604, 0.4805, 0.5005, 0.5205, 0.5405, 0.5605],
     [0.428, 0.448, 0.468, 0.488, 0.5083, 0.5283, 0.5483, 0.5684]],
    [[0.7554, 0.7754, 0.7954, 0.8154, 0.8354, 0.8555, 0.8755, 0.8955],
     [0.7622, 0.7822, 0.8022, 0.8223, 0.8423, 0.8623, 0.8823, 0.9023]]], np.float32)
G2 = np.array([
    [[0.0, 0.02, 0.04, 0.06, 0.08, 0.1, 0.12, 0.14],
     [0.0922, 0.1122, 0.1322, 0.1522, 0.1721, 0.1921, 0.2122, 0.2322]],
    [[0.3201, 0.3401, 0.3601, 0.3801, 0.4001, 0.4202, 0.4402, 0.4602],
     [0.428, 0.448, 0.468, 0.488, 0.5083, 0.5283, 0.5483, 0.5684]],
    [[0.6401, 0.6602, 0.6802, 0.7002, 0.7202, 0.7402, 0.7603, 0.7803],
     [0.7622, 0.7822, 0.8022, 0.8223, 0.8423, 0.8623, 0.8823, 0.9023]]], np.float32)


def golden_inputs(n=48, shape=(3, 2, 8)):
    base = np.arange(n, dtype=np.float32).astype(np.float16).reshape(shape)
    mk = lambda c: (base * np.float16(1.0 / c)).astype(np.float16).view(np.uint16)
    return mk(30), mk(40), mk(50)  # q, k, v as [heads=3, seq=2, d=8]


def main():
    q, k, v = golden_inputs()
    np.savez(os.path.join(OUT, "reference_tables.npz"), G1=G1, G2=G2, q=q, k=k, v=v,
             softmax_scale=np.float32(0.5))

    rng = np.random.default_rng(20260928)
    cases = {}
    # paged decode, bf16, d=128, GQA 4 (Llama-3.1-8B head shape), ragged lengths incl. 0 and 1
    lens = np.array([0, 1, 15, 16, 17, 77], np.int32)
    kc, vc, bt = make_paged_cache(rng, 24, 16, 2, 128, BF16, lens)
    qd = rand_half(rng, (len(lens), 1, 8, 128), BF16)
    sc = np.float32(1.0 / np.sqrt(128))
    cases.update(d1_q=qd, d1_kc=kc, d1_vc=vc, d1_bt=bt, d1_lens=lens, d1_scale=sc,
                 d1_out_f32=A.flash_attn_kv_cache(qd, kc, vc, sc, BF16, bt, lens, mode="f32"),
                 d1_out_kernel=A.flash_attn_kv_cache(qd, kc, vc, sc, BF16, bt, lens, mode="kernel"))
    # paged decode, f16, d=64, GQA 2, page 32
    lens = np.array([33, 64, 5], np.int32)
    kc, vc, bt = make_paged_cache(rng, 8, 32, 2, 64, F16, lens)
    qd = rand_half(rng, (len(lens), 1, 4, 64), F16)
    sc = np.float32(1.0 / np.sqrt(64))
    cases.update(d2_q=qd, d2_kc=kc, d2_vc=vc, d2_bt=bt, d2_lens=lens, d2_scale=sc,
                 d2_out_f32=A.flash_attn_kv_cache(qd, kc, vc, sc, F16, bt, lens, mode="f32"),
                 d2_out_kernel=A.flash_attn_kv_cache(qd, kc, vc, sc, F16, bt, lens, mode="kernel"))
    # causal varlen prefill, bf16, d=64, GQA 2, three sequences (one of length 1)
    cu = np.array([0, 37, 38, 90], np.int32)
    qp = rand_half(rng, (90, 4, 64), BF16)
    kp = rand_half(rng, (90, 2, 64), BF16)
    vp = rand_half(rng, (90, 2, 64), BF16)
    cases.update(p1_q=qp, p1_k=kp, p1_v=vp, p1_cu=cu, p1_scale=sc,
                 p1_out_f32=A.flash_attn_varlen(qp, kp, vp, cu, cu, sc, True, BF16, mode="f32"),
                 p1_out_kernel=A.flash_attn_varlen(qp, kp, vp, cu, cu, sc, True, BF16, mode="kernel"))
    # reshape_and_cache_flash: 10 tokens -> slots 0..9 of 2 pages of 8 (cache_manager_tests.rs:553-616)
    key = rand_half(rng, (10, 4, 64), F16)
    val = rand_half(rng, (10, 4, 64), F16)
    kcache = np.zeros((2, 8, 4, 64), np.uint16)
    vcache = np.zeros((2, 8, 4, 64), np.uint16)
    CO.reshape_and_cache_flash(key, val, kcache, vcache, np.arange(10))
    cases.update(r1_key=key, r1_val=val, r1_kcache=kcache, r1_vcache=vcache)
    # rmsnorm / rope (parity unpinned by the reference: oracle restates Candle)
    x = rand_half(rng, (5, 256), BF16)
    w = rand_half(rng, (256,), BF16)
    cases.update(n1_x=x, n1_w=w, n1_y=NR.rms_norm(x, w, 1e-5, BF16))
    cos, sin = NR.rope_table(64, 64, 500000.0, BF16,
                             dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                  original_max_position_embeddings=8192))
    xr = rand_half(rng, (7, 3, 64), BF16)
    pos = np.array([0, 1, 2, 3, 40, 41, 63], np.int64)
    cases.update(o1_x=xr, o1_cos=cos, o1_sin=sin, o1_pos=pos, o1_y=NR.rope(xr, cos, sin, pos, BF16))
    np.savez_compressed(os.path.join(OUT, "oracle_cases.npz"), **cases)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
