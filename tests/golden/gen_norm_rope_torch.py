#!/usr/bin/env python3
"""Cross-check fixtures for the RMSNorm / RoPE oracles, generated with torch in the BUILD container.

Why: the arithmetic of both ops lives in candle-nn 0.9.2-alpha.1, which is not under /root/reference, and no reference
test holds a value (parity unpinned).  torch is a third, independent implementation of the same published
formulas; it is NOT the reference, so the parity stays "unpinned" -- but a transcription error in oracle/norm_rope_oracle.py
(or in the kernels) would have to be repeated in torch's own kernels to go unnoticed.

What is stored (tests/golden/norm_rope_torch.npz): inputs as storage bits and torch's outputs --
  * rms_norm: y = (x.float() * rsqrt(mean(x^2) + eps) * w.float()) rounded once       (torch.nn.functional.rms_norm on f32)
  * rope, per-op: rotate-half evaluated with bf16 / f16 TENSOR ops (every product and the sum rounded by torch)
  * rope, f32: the same formula in f32, rounded once
  * cos / sin table: torch.cos / torch.sin of pos * theta^(-2j/d) in f32, rounded to the dtype
Nothing of torch travels to the GPU box: only this .npz does.   Run:  python tests/golden/gen_norm_rope_torch.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TD = {0: torch.float16, 1: torch.bfloat16}


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    g = torch.Generator().manual_seed(20260928)
    out = {}
    for code, td in TD.items():
        tag = "f16" if code == 0 else "bf16"
        # ---- rms_norm: rows x hidden, including a large-magnitude row and a tiny one ----
        x = (torch.randn(9, 4096, generator=g) * torch.tensor([1, 1, 1, 30, 1e-2, 1, 5, 1, 1]).view(9, 1)).to(td)
        w = (1.0 + 0.1 * torch.randn(4096, generator=g)).to(td)
        eps = 1e-5
        y = torch.nn.functional.rms_norm(x.float(), (4096,), w.float(), eps).to(td)
        out[f"rms_{tag}_x"], out[f"rms_{tag}_w"], out[f"rms_{tag}_y"] = bits(x), bits(w), bits(y)
        out[f"rms_{tag}_eps"] = np.float32(eps)
        # ---- cos / sin table (llama.rs:154-200 without Llama-3 scaling), head_dim 128, theta 5e5 ----
        d, theta, max_pos = 128, 500000.0, 640
        inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float32), torch.arange(0, d, 2, dtype=torch.float32) / d)
        ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
        cos, sin = torch.cos(ang).to(td), torch.sin(ang).to(td)
        out[f"tab_{tag}_cos"], out[f"tab_{tag}_sin"] = bits(cos), bits(sin)
        # ---- rope on [T, heads, d] ----
        T, H = 37, 5
        xr = torch.randn(T, H, d, generator=g).to(td)
        pos = torch.randint(0, max_pos, (T,), generator=g)
        c, s = cos[pos][:, None, :], sin[pos][:, None, :]
        x1, x2 = xr[..., : d // 2], xr[..., d // 2:]
        per_op = torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], -1)                       # tensor-dtype arithmetic, a rounding per op
        cf, sf, x1f, x2f = c.float(), s.float(), x1.float(), x2.float()
        fused = torch.cat([x1f * cf - x2f * sf, x1f * sf + x2f * cf], -1).to(td)        # f32, one rounding
        out[f"rope_{tag}_x"], out[f"rope_{tag}_pos"] = bits(xr), pos.numpy().astype(np.int64)
        out[f"rope_{tag}_per_op"], out[f"rope_{tag}_fused"] = bits(per_op), bits(fused)
    np.savez_compressed(os.path.join(HERE, "norm_rope_torch.npz"), **out)
    print("wrote norm_rope_torch.npz with", len(out), "arrays; torch", torch.__version__)


if __name__ == "__main__":
    main()
