"""Differential fuzzing of the projection entry points (llama.rs:269-271,311,364-365) and of RMSNorm / RoPE: random batch sizes 1 .. 256
(every kernel family of csrc/linear_*.hip: VALU / MFMA weight streaming, 17..64-row tiles with in-launch K-split merges, the 65..256-row wide tiles, the sliced
fallback), random in / out features and padded strides.  Test infrastructure: oracle/ is the checker.

  linear     atoma_linear_decode against the exactly accumulated product (<= 1 ulp + 3e-5, < 1 % of the outputs off), padding untouched, twice to the bit
             (a K-split merged by its last arriver must not depend on who came last);
             _residual / _silu_mul / _rmsnorm / _rmsnorm_silu_mul against projection + separate op, bit for bit (the rounding points are part of the contract)
  qkv_rope   atoma_linear_decode_qkv_rope_cache against atoma_linear_decode + atoma_rope_qk_cache, bit for bit (q/k/v output and both caches)
  swap       atoma_swap_blocks_multi (cache_manager.rs:196-402 / worker.rs:602-632 loop): 1 .. 16 tensors, pages of 16 B .. 17 MiB (odd sizes too), gpu -> gpu,
             cpu -> gpu and gpu -> cpu with pageable host memory at odd byte offsets (the pinned bounce ring and its copy threads) or pinned, byte for byte
  norm_rope  atoma_rms_norm (<= 1 ulp vs oracle), atoma_add_rms_norm == atoma_add + atoma_rms_norm bit for bit, atoma_rope per-op mode bit-exact vs oracle

    python tests/fuzz_ops.py --seconds 300 [--seed 0] [--kinds linear,qkv_rope,norm_rope]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import linear_oracle as LO, norm_rope_oracle as NR  # noqa: E402
from oracle.halfs import F16, BF16, to_f32, from_f32  # noqa: E402
from util import rand_half  # noqa: E402

KINDS = ("linear", "linear", "qkv_rope", "norm_rope")
FP8_BASE = 2 * 10 ** 6    # ... decode over the fp8 (e4m3fn) KV cache: atoma_paged_decode_fp8 against the definition over the dequantised cache
SWAP_BASE = 10 ** 6       # seeds from here on: swap_blocks / swap_blocks_multi (added after the first campaign; earlier seeds keep their cases)
BATCHES = [1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 66, 96, 127, 128, 129, 160, 191, 192, 193, 224, 255, 256]
K_UNITS = [1, 2, 3, 4, 8, 12, 16, 24, 32, 40, 56, 64, 112, 128]                       # x 128
N_UNITS = [1, 2, 3, 4, 7, 8, 9, 16, 24, 32, 48, 64, 80, 96, 128, 256, 257, 384, 512, 896, 1792]      # x 16
MAX_WEIGHTS = 24 << 20


def draw(seed, kinds=KINDS):
    rng = np.random.default_rng(seed)
    kind = kinds[int(rng.integers(len(kinds)))]
    if seed >= FP8_BASE:
        hk, g = (int(x) for x in rng.choice([(8, 4), (8, 1), (2, 3), (1, 8), (1, 13), (2, 20), (4, 16), (8, 2), (1, 5)]))
        B = int(rng.choice([1, 2, 7, 16, 40, 64, 130, 256, 320]))
        top = max(20, min(int(rng.choice([40, 300, 1200, 4000])), (1 << 22) // (B * hk * g)))
        lens = rng.integers(0, top + 1, B)
        lens[rng.integers(0, B)] = top
        if rng.integers(4) == 0:
            lens[:] = top
        return dict(seed=int(seed), kind="fp8_decode", dtype=int(rng.choice([BF16, BF16, F16])), B=B, hk=hk, h=hk * g, page=int(rng.choice([16, 16, 32, 64])),
                    lens=[int(x) for x in lens], mqk=int(rng.integers(4) != 0))
    if seed >= SWAP_BASE:
        block = int(rng.choice([16, 48, 2048, 4104, 32768, 32768, 131072, 1 << 20, (17 << 20) + 16]))
        nt = int(rng.choice([1, 2, 5, 16]))
        nb = int(max(2, min(int(rng.choice([4, 17, 60])), (192 << 20) // (block * nt))))
        return dict(seed=int(seed), kind="swap", block=block, nt=nt, nb=nb, pairs=int(rng.integers(1, nb + 1)), dir=int(rng.choice([0, 1, 1, 2, 2])),
                    pinned=bool(rng.integers(3) == 0), offset=int(rng.choice([0, 0, 8, 2])))
    c = dict(seed=int(seed), kind=kind, dtype=int(rng.choice([BF16, BF16, F16])), B=int(rng.choice(BATCHES)) if rng.integers(3) else int(rng.integers(1, 257)))
    if kind == "linear":
        K = 128 * int(rng.choice(K_UNITS))
        N = 16 * int(rng.choice(N_UNITS))
        ep = int(rng.choice([0, 0, 1, 2, 3, 4]))
        if ep in (2, 4):
            N = max(32, N // 32 * 32)                    # intermediate size; the stacked matrix has 2 N rows
        while N * (2 if ep in (2, 4) else 1) * K > MAX_WEIGHTS:
            N = max(32, N // 64 * 32)
        c.update(K=K, N=N, ep=ep, xpad=int(rng.choice([0, 0, 8, 64])), ypad=int(rng.choice([0, 0, 4, 64])), wpad=int(rng.choice([0, 0, 8])))
    elif kind == "qkv_rope":
        d = int(rng.choice([64, 128, 128]))
        hk = int(rng.choice([1, 2, 4, 8]))
        c.update(K=128 * int(rng.choice([4, 8, 16, 32, 64])), d=d, hk=hk, h=hk * int(rng.choice([1, 2, 4, 8])), page=int(rng.choice([16, 32])))
    else:
        c.update(hidden=8 * int(rng.choice([1, 5, 16, 96, 128, 512, 640, 1024, 2048])), heads=int(rng.choice([1, 3, 8, 32])), d=int(rng.choice([32, 64, 96, 128, 256])),
                 pad=int(rng.choice([0, 8, 64])), eps=float(rng.choice([1e-5, 1e-6])))
    return c


def _ulp_check(got, ref, dtype, what):
    g, r = to_f32(got, dtype), to_f32(ref, dtype)
    if not np.isfinite(g).all():
        return f"{what}: non-finite output"
    ulp = 2.0 ** -7 if dtype == BF16 else 2.0 ** -10
    err = np.abs(g - r)
    if not (err <= ulp * np.abs(r) + 3e-5).all():
        return f"{what}: max err {err.max():.3e} beyond 1 ulp + 3e-5"
    if got.size >= 256 and (got != ref).mean() >= 0.01:
        return f"{what}: {(got != ref).mean():.4f} of the outputs differ from the exactly accumulated product"
    return None


def run_case(gpu, c):
    rng = np.random.default_rng(c["seed"] + (1 << 41))
    L, D, dtype, B = gpu.lib, gpu.DeviceBuffer, c.get("dtype"), c.get("B")

    def ok(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {gpu.last_error()}")
    if c["kind"] == "fp8_decode":
        import test_kv_fp8_gpu as T8
        import fuzz_parity as FP
        from util import ulp_tol, attn_atol
        from oracle import fp8_oracle as F8, attn_oracle as A
        d, h, hk, page, lens = 128, c["h"], c["hk"], c["page"], np.asarray(c["lens"], np.int32)
        nb = int(sum((int(x) + page - 1) // page for x in lens)) + 3
        kc8, vc8, ks, vs, bt = T8.make_fp8_cache(rng, nb, page, hk, d, lens)
        q = rand_half(rng, (B, h, d), dtype)
        scale = np.float32(d ** -0.5)
        ok(L.atoma_set_option(b"decode_fp8_mqk", c["mqk"]), "set_option")
        try:
            out = T8.gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, scale, dtype)
        finally:
            L.atoma_set_option(b"decode_fp8_mqk", 1)
        ref = T8.oracle_decode(q, kc8, vc8, ks, vs, bt, lens, scale, dtype)
        g, r = to_f32(out, dtype), to_f32(ref, dtype)
        if not np.isfinite(g).all():
            return "non-finite output"
        kf = vf = None
        for b in range(B):
            n = int(lens[b])
            if n == 0:
                if out[b].any():
                    return f"seq {b}: an empty sequence must give exact zeros"
                continue
            err = np.abs(g[b] - r[b])
            if (err > ulp_tol(r[b], dtype, attn_atol(dtype, n))).any():
                if kf is None:
                    kf, vf = F8.dequantize(kc8, ks), F8.dequantize(vc8, vs)
                kb, vb = A.gather_paged(kf, bt[b], n, page), A.gather_paged(vf, bt[b], n, page)
                ab, sq = FP.p_bounds(to_f32(q[b][None], dtype), kb, vb, scale, False, None)
                own = ulp_tol(r[b], dtype, 1e-3) + np.minimum(FP.E_MAX[dtype] * ab[0], 6 * FP.E_SIG[dtype] * sq[0])
                if (err > own).any():
                    return f"seq {b} (L={n}): max err {err.max():.3e} beyond the bound of its own probabilities"
        return None
    if c["kind"] == "swap":
        import ctypes as C
        block, nt, nb, kind = c["block"], c["nt"], c["nb"], c["dir"]
        src_pages, dst_pages = rng.permutation(nb)[:c["pairs"]], rng.permutation(nb)[:c["pairs"]]
        m = np.stack([src_pages, dst_pages], 1).astype(np.int64)
        off = 0 if c["pinned"] else c["offset"]
        keep, host_ptrs = [], []

        def host_tensor():
            if c["pinned"]:
                p = L.atoma_host_alloc(nb * block)
                if not p:
                    raise RuntimeError("atoma_host_alloc failed")
                a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb * block,)).reshape(nb, block)
                host_ptrs.append(p)
            else:
                raw = np.empty(nb * block + 64, np.uint8)
                keep.append(raw)
                a = raw[off:off + nb * block].reshape(nb, block)
            a[...] = rng.integers(0, 256, (nb, 1), dtype=np.uint8) ^ np.arange(block, dtype=np.uint8)[None, :]
            return a
        dev_np = [(rng.integers(0, 256, (nb, 1), dtype=np.uint8) + np.arange(block, dtype=np.uint8)[None, :] * 3).astype(np.uint8) for _ in range(nt)]
        devs = [D.from_numpy(a) for a in dev_np]
        try:
            if kind == 0:
                dst_np = [np.full((nb, block), 7 + i, np.uint8) for i in range(nt)]
                dsts = [D.from_numpy(a) for a in dst_np]
                sp, dp = (C.c_void_p * nt)(*[b.ptr for b in devs]), (C.c_void_p * nt)(*[b.ptr for b in dsts])
                ok(L.atoma_swap_blocks_multi(sp, dp, nt, m.ctypes.data, len(m), block, 0, None), "swap_blocks_multi gpu -> gpu")
                gpu.synchronize()
                got = [b.numpy(np.uint8, (nb, block)) for b in dsts]
                want = [a.copy() for a in dst_np]
                for t in range(nt):
                    want[t][dst_pages] = dev_np[t][src_pages]
            else:
                hosts = [host_tensor() for _ in range(nt)]
                hp = (C.c_void_p * nt)(*[h.ctypes.data for h in hosts])
                gp = (C.c_void_p * nt)(*[b.ptr for b in devs])
                if kind == 1:                                                   # cpu -> gpu
                    ok(L.atoma_swap_blocks_multi(hp, gp, nt, m.ctypes.data, len(m), block, 1, None), "swap_blocks_multi cpu -> gpu")
                    gpu.synchronize()
                    got = [b.numpy(np.uint8, (nb, block)) for b in devs]
                    want = [a.copy() for a in dev_np]
                    for t in range(nt):
                        want[t][dst_pages] = hosts[t][src_pages]
                else:
                    want = [h.copy() for h in hosts]
                    ok(L.atoma_swap_blocks_multi(gp, hp, nt, m.ctypes.data, len(m), block, 2, None), "swap_blocks_multi gpu -> cpu")
                    gpu.synchronize()
                    got = [h.copy() for h in hosts]
                    for t in range(nt):
                        want[t][dst_pages] = dev_np[t][src_pages]
            for t in range(nt):
                if not np.array_equal(got[t], want[t]):
                    bad = np.argwhere((got[t] != want[t]).any(axis=1))[:, 0]
                    return f"tensor {t}: pages {bad[:8].tolist()} differ ({len(bad)} pages)"
            return None
        finally:
            for p in host_ptrs:
                L.atoma_host_free(p)
    if c["kind"] == "linear":
        K, N, ep = c["K"], c["N"], c["ep"]
        xs, ws = K + c["xpad"], K + c["wpad"]
        rows_w = 2 * N if ep in (2, 4) else N
        ys = N + c["ypad"]
        x = rand_half(rng, (B, xs), dtype)
        w = rand_half(rng, (rows_w, ws), dtype, K ** -0.5)
        dx, dw = D.from_numpy(x), D.from_numpy(w)
        xk, wk = np.ascontiguousarray(x[:, :K]), np.ascontiguousarray(w[:, :K])

        def out_buf(width, stride):
            b = D(B * stride * 2)
            b.fill_bytes(0xAB)
            return b

        def read(b, width, stride):
            a = b.numpy(np.uint16, (B, stride))
            if stride > width and not (a[:, width:] == 0xABAB).all():
                raise RuntimeError("padding between output rows was written")
            return a[:, :width].copy()
        if ep == 0:
            y, y2 = out_buf(N, ys), out_buf(N, ys)
            ok(L.atoma_linear_decode(dx.ptr, dw.ptr, y.ptr, B, K, N, xs, ws, ys, dtype, None), "linear_decode")
            ok(L.atoma_linear_decode(dx.ptr, dw.ptr, y2.ptr, B, K, N, xs, ws, ys, dtype, None), "linear_decode")
            gpu.synchronize()
            a, b = read(y, N, ys), read(y2, N, ys)
            if not np.array_equal(a, b):
                return "two identical calls differ"
            return _ulp_check(a, LO.linear(xk, wk, dtype), dtype, "linear_decode")
        if ep == 1:
            res = rand_half(rng, (B, N), dtype)
            dr = D.from_numpy(res)
            y, y2 = out_buf(N, ys), out_buf(N, ys)
            ok(L.atoma_linear_decode(dx.ptr, dw.ptr, y.ptr, B, K, N, xs, ws, ys, dtype, None), "linear_decode")
            gpu.synchronize()
            plain = read(y, N, ys)
            dp, ds = D.from_numpy(plain), D(B * N * 2)
            ok(L.atoma_add(dp.ptr, dr.ptr, ds.ptr, B * N, dtype, None), "add")
            ok(L.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, y2.ptr, B, K, N, xs, ws, N, ys, dtype, None), "linear_decode_residual")
            gpu.synchronize()
            if not np.array_equal(ds.numpy(np.uint16, (B, N)), read(y2, N, ys)):
                return "_residual differs from projection + add"
            return _ulp_check(plain, LO.linear(xk, wk, dtype), dtype, "linear_decode")
        if ep == 2:
            gu, act, act2 = D(B * 2 * N * 2), D(B * N * 2), out_buf(N, ys)
            ok(L.atoma_linear_decode(dx.ptr, dw.ptr, gu.ptr, B, K, 2 * N, xs, ws, 2 * N, dtype, None), "linear_decode")
            ok(L.atoma_silu_mul(gu.ptr, gu.ptr + N * 2, act.ptr, B, N, 2 * N, 2 * N, N, dtype, None), "silu_mul")
            ok(L.atoma_linear_decode_silu_mul(dx.ptr, dw.ptr, act2.ptr, B, K, N, xs, ws, ys, dtype, None), "linear_decode_silu_mul")
            gpu.synchronize()
            if not np.array_equal(act.numpy(np.uint16, (B, N)), read(act2, N, ys)):
                return "_silu_mul differs from projection + silu_mul"
            return _ulp_check(gu.numpy(np.uint16, (B, 2 * N)), LO.linear(xk, wk, dtype), dtype, "linear_decode (stacked gate / up)")
        g = rand_half(rng, (K,), dtype)
        dg, xn, scratch = D.from_numpy(g), D(B * K * 2), D(B * K * 2)
        eps = 1e-5
        ok(L.atoma_rms_norm(dx.ptr, dg.ptr, xn.ptr, B, K, xs, K, eps, dtype, None), "rms_norm")
        if ep == 3:
            y, y2 = out_buf(N, ys), out_buf(N, ys)
            ok(L.atoma_linear_decode(xn.ptr, dw.ptr, y.ptr, B, K, N, K, ws, ys, dtype, None), "linear_decode")
            ok(L.atoma_linear_decode_rmsnorm(dx.ptr, dg.ptr, eps, dw.ptr, y2.ptr, scratch.ptr, B, K, N, xs, ws, ys, dtype, None), "linear_decode_rmsnorm")
            gpu.synchronize()
            return None if np.array_equal(read(y, N, ys), read(y2, N, ys)) else "_rmsnorm differs from rms_norm + projection"
        act, act2 = out_buf(N, ys), out_buf(N, ys)
        ok(L.atoma_linear_decode_silu_mul(xn.ptr, dw.ptr, act.ptr, B, K, N, K, ws, ys, dtype, None), "linear_decode_silu_mul")
        ok(L.atoma_linear_decode_rmsnorm_silu_mul(dx.ptr, dg.ptr, eps, dw.ptr, act2.ptr, scratch.ptr, B, K, N, xs, ws, ys, dtype, None), "linear_decode_rmsnorm_silu_mul")
        gpu.synchronize()
        return None if np.array_equal(read(act, N, ys), read(act2, N, ys)) else "_rmsnorm_silu_mul differs from rms_norm + _silu_mul"
    if c["kind"] == "qkv_rope":
        K, h, hk, d, page = c["K"], c["h"], c["hk"], c["d"], c["page"]
        width, nb = (h + 2 * hk) * d, max(4, B // page + 3)
        x = rand_half(rng, (B, K), dtype)
        w = rand_half(rng, (width, K), dtype, K ** -0.5)
        cos, sin = rand_half(rng, (2048, d // 2), dtype), rand_half(rng, (2048, d // 2), dtype)
        pos = rng.integers(0, 2048, B).astype(np.int64)
        slots = rng.permutation(nb * page)[:B].astype(np.int64)
        slots[rng.integers(0, B)] = -1
        dx, dw, dc, ds, dp, dsl = (D.from_numpy(a) for a in (x, w, cos, sin, pos, slots))
        outs = []
        for fused in (False, True):
            qkv = D.zeros((B, width), np.uint16)
            kc, vc = D.zeros((nb * page * hk * d,), np.uint16), D.zeros((nb * page * hk * d,), np.uint16)
            if fused:
                ok(L.atoma_linear_decode_qkv_rope_cache(dx.ptr, dw.ptr, qkv.ptr, kc.ptr, vc.ptr, dsl.ptr, dc.ptr, ds.ptr, dp.ptr, B, K, h, hk, d, K, K, width,
                                                        page * hk * d, page, dtype, 1, None), "linear_decode_qkv_rope_cache")
            else:
                ok(L.atoma_linear_decode(dx.ptr, dw.ptr, qkv.ptr, B, K, width, K, K, width, dtype, None), "linear_decode")
                ok(L.atoma_rope_qk_cache(qkv.ptr, qkv.ptr + h * d * 2, qkv.ptr + (h + hk) * d * 2, kc.ptr, vc.ptr, dsl.ptr, dc.ptr, ds.ptr, dp.ptr, B, h, hk, d,
                                         width, width, width, page * hk * d, page, dtype, 1, None), "rope_qk_cache")
            gpu.synchronize()
            outs.append((qkv.numpy(np.uint16, (B, width)), kc.numpy(np.uint16, (nb * page * hk * d,)), vc.numpy(np.uint16, (nb * page * hk * d,))))
        for name, a, b in zip(("q/k/v", "key cache", "value cache"), *outs):
            if not np.array_equal(a, b):
                return f"the fused entry's {name} differs from projection + rope_qk_cache ({(a != b).sum()} elements)"
        return None
    # norm_rope
    H, heads, d, pad, eps = c["hidden"], c["heads"], c["d"], c["pad"], c["eps"]
    xs = H + pad
    x, r = rand_half(rng, (B, xs), dtype), rand_half(rng, (B, xs), dtype)
    wt = rand_half(rng, (H,), dtype)
    dx, dr, dwt = D.from_numpy(x), D.from_numpy(r), D.from_numpy(wt)
    y, s, y2, s2 = D(B * H * 2), D(B * H * 2), D(B * H * 2), D(B * H * 2)
    ok(L.atoma_rms_norm(dx.ptr, dwt.ptr, y.ptr, B, H, xs, H, eps, dtype, None), "rms_norm")
    gpu.synchronize()
    ref = NR.rms_norm(np.ascontiguousarray(x[:, :H]), wt, eps, dtype)
    diff = np.abs(y.numpy(np.uint16, (B, H)).astype(np.int32) - ref.astype(np.int32))
    if diff.max() > 1:
        return f"rms_norm: {int(diff.max())} ulp from the oracle"
    xa, ra = D.from_numpy(np.ascontiguousarray(x[:, :H])), D.from_numpy(np.ascontiguousarray(r[:, :H]))
    ok(L.atoma_add(xa.ptr, ra.ptr, s.ptr, B * H, dtype, None), "add")
    ok(L.atoma_rms_norm(s.ptr, dwt.ptr, y.ptr, B, H, H, H, eps, dtype, None), "rms_norm")
    ok(L.atoma_add_rms_norm(dx.ptr, dr.ptr, dwt.ptr, s2.ptr, y2.ptr, B, H, xs, xs, H, H, eps, dtype, None), "add_rms_norm")
    gpu.synchronize()
    if not (np.array_equal(s.numpy(np.uint16, (B, H)), s2.numpy(np.uint16, (B, H))) and np.array_equal(y.numpy(np.uint16, (B, H)), y2.numpy(np.uint16, (B, H)))):
        return "add_rms_norm differs from add + rms_norm"
    cos, sin = NR.rope_table(512, d, 10000.0, dtype)
    pos = rng.integers(0, 512, B).astype(np.int64)
    q = rand_half(rng, (B, heads, d), dtype)
    dq, dc, dsn, dp, dy = D.from_numpy(q), D.from_numpy(cos), D.from_numpy(sin), D.from_numpy(pos), D(q.nbytes)
    ok(L.atoma_rope(dq.ptr, dy.ptr, dc.ptr, dsn.ptr, dp.ptr, B, heads, d, heads * d, d, heads * d, d, dtype, 1, None), "rope")
    gpu.synchronize()
    if not np.array_equal(dy.numpy(np.uint16, q.shape), NR.rope(q, cos, sin, pos, dtype)):
        return "rope (per-op rounding) differs from the oracle"
    return None


def try_case(gpu, c):
    try:
        return run_case(gpu, c)
    except (RuntimeError, AssertionError) as e:
        return f"raised {type(e).__name__}: {str(e)[:300]}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--kinds", default=",".join(KINDS))
    a = ap.parse_args()
    import atoma_hip as gpu
    gpu.set_device(0)
    kinds = tuple(a.kinds.split(","))
    t0, n, fails, per_kind, seed = time.time(), 0, [], {}, a.seed
    while time.time() - t0 < a.seconds:
        c = draw(seed + (SWAP_BASE if n % 8 == 7 else FP8_BASE if n % 8 == 3 else 0), kinds)
        msg = try_case(gpu, c)
        per_kind[c["kind"]] = per_kind.get(c["kind"], 0) + 1
        if msg:
            fails.append(dict(case=c, finding=msg))
            print(json.dumps(fails[-1]), file=sys.stderr, flush=True)
        n, seed = n + 1, seed + 1
    print(json.dumps(dict(cases=n, seeds=[a.seed, seed - 1], per_kind=per_kind, seconds=round(time.time() - t0, 1), failures=len(fails), findings=fails[:40])), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
