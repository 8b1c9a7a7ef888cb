"""numpy restatement of RMSNorm and RoPE as the reference calls them (TEST INFRASTRUCTURE).

PARITY UNPINNED: the arithmetic of both ops lives in candle-nn / candle-kernels
0.9.2-alpha.1 (Cargo.lock:523-535), which is not under /root/reference, and no
reference test holds a value for either op.  What is restated here is the published
algorithm of that pinned version, anchored on the reference's call sites:

* RMSNorm  models/src/llama.rs:402,408,474 -> candle_nn::ops::rms_norm.  CUDA kernel
  (candle-kernels reduce.cu ``rmsnorm``): f32 sum of squares, ``scale = rsqrt(mean+eps)``,
  ``y = T(scale * f32(x) * f32(w))`` -- one rounding.  ``mode="cpu"`` restates Candle's CPU
  path instead (f64 sum, ``m = T(sqrt(mean+eps))``, ``y = x / m * w`` with a rounding per op).
* RoPE     models/src/llama.rs:218-251 -> candle_nn::rotary_emb::rope (non-interleaved,
  "rotate half"): ``y1 = x1*c - x2*s ; y2 = x1*s + x2*c`` evaluated *in the tensor dtype*
  (each product and the sum rounded to bf16/f16) on both Candle backends;
  ``mode="fused"`` is the f32 round-once variant, kept only to report the ulp distance.
* cos/sin table  models/src/llama.rs:146-200 (``Cache::new``): ``inv_freq_j = theta^(-2j/d)``
  in f32, optional Llama-3 wavelength scaling, ``angle = pos * inv_freq`` in f32, table
  rounded to the model dtype.
"""
import numpy as np

from .halfs import to_f32, from_f32, round_through


def rms_norm(x, w, eps, dtype, mode="cuda"):
    """x ``[rows, hidden]`` storage-form, w ``[hidden]`` storage-form -> storage-form."""
    xf, wf = to_f32(x, dtype), to_f32(w, dtype)
    n = xf.shape[-1]
    if mode == "cuda":
        ss = (xf.astype(np.float64) ** 2).sum(-1, keepdims=True)       # exact-ish f32 sum
        scale = (1.0 / np.sqrt(ss / n + np.float64(np.float32(eps)))).astype(np.float32)
        return from_f32((scale * xf) * wf, dtype)
    if mode == "cpu":
        ss = (xf.astype(np.float64) ** 2).sum(-1, keepdims=True)
        m = round_through(np.sqrt(ss / n + float(eps)).astype(np.float32), dtype)
        q = round_through(xf / m, dtype)
        return from_f32(q * wf, dtype)
    raise ValueError(mode)


def inv_freq(head_dim, rope_theta, rope_scaling=None):
    """models/src/llama.rs:146-187.  ``rope_scaling`` = dict(factor, low_freq_factor,
    high_freq_factor, original_max_position_embeddings) for the Llama-3 rule, or None."""
    i = np.arange(0, head_dim, 2, dtype=np.float32)
    f = (np.float32(1) / np.power(np.float32(rope_theta), i / np.float32(head_dim))).astype(np.float32)
    if rope_scaling is None:
        return f
    orig = np.float32(rope_scaling["original_max_position_embeddings"])
    lo_f, hi_f = np.float32(rope_scaling["low_freq_factor"]), np.float32(rope_scaling["high_freq_factor"])
    factor = np.float32(rope_scaling["factor"])
    low_wl, high_wl = orig / lo_f, orig / hi_f
    wavelen = np.float32(2) * np.float32(np.pi) / f
    smooth = (orig / wavelen - lo_f) / (hi_f - lo_f)
    mid = (np.float32(1) - smooth) * f / factor + smooth * f
    out = np.where(wavelen < high_wl, f, np.where(wavelen > low_wl, f / factor, mid))
    return out.astype(np.float32)


def rope_table(max_pos, head_dim, rope_theta, dtype, rope_scaling=None):
    """cos,sin ``[max_pos, d/2]`` storage-form (models/src/llama.rs:189-199)."""
    ang = np.arange(max_pos, dtype=np.float32)[:, None] * inv_freq(head_dim, rope_theta, rope_scaling)[None, :]
    ang = ang.astype(np.float32)
    return from_f32(np.cos(ang).astype(np.float32), dtype), from_f32(np.sin(ang).astype(np.float32), dtype)


def rope(x, cos_table, sin_table, positions, dtype, mode="per_op"):
    """x ``[T, heads, d]`` storage-form; tables ``[max_pos, d/2]``; positions ``[T]`` int64.

    Same numbers as the reference's ``rope(x[1,h,T,d], cos[T,d/2], sin[T,d/2])`` after its
    transposes (models/src/llama.rs:236-250,273-303): the op is independent per (t, head)."""
    xf = to_f32(x, dtype)
    T, H, d = xf.shape
    c = to_f32(cos_table, dtype)[np.asarray(positions)][:, None, :]
    s = to_f32(sin_table, dtype)[np.asarray(positions)][:, None, :]
    x1, x2 = xf[..., : d // 2], xf[..., d // 2:]
    if mode == "per_op":
        r = lambda a: round_through(a.astype(np.float32), dtype)
        y1 = r(r(x1 * c) - r(x2 * s))
        y2 = r(r(x1 * s) + r(x2 * c))
    elif mode == "fused":
        y1 = x1 * c - x2 * s
        y2 = x1 * s + x2 * c
    else:
        raise ValueError(mode)
    return from_f32(np.concatenate([y1, y2], -1), dtype)
