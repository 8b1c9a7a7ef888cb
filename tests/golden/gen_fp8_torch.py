#!/usr/bin/env python3
"""Pins oracle/fp8_oracle.py's e4m3fn conversion against torch.float8_e4m3fn (build container only; the .npz travels).

Stored: every one of the 256 codes decoded by torch; a sweep of float32 inputs (all binades the format covers, exact
ties between neighbouring codes, values beyond +-448) with torch's encoding -- torch's cast does NOT saturate (it
produces NaN beyond the range), so inputs are clamped to +-448 first, which is exactly the cache-write rule.
Run:  python tests/golden/gen_fp8_torch.py"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    codes = torch.arange(256, dtype=torch.uint8)
    dec = codes.view(torch.float8_e4m3fn).float().numpy()
    rng = np.random.default_rng(8)
    finite = dec[np.isfinite(dec)]
    pos = np.sort(np.unique(np.abs(finite)))
    ties = (pos[1:] + pos[:-1]) / 2                                    # exactly representable in f32
    x = np.concatenate([finite, ties, -ties, np.nextafter(ties.astype(np.float32), np.float32(np.inf)),
                        np.nextafter(ties.astype(np.float32), np.float32(0)), rng.standard_normal(20000).astype(np.float32) * 3,
                        (rng.standard_normal(5000) * 200).astype(np.float32), np.float32([447.9, 448, 449, 464, 480, 1e4, -1e4, 1e-3, 2 ** -9, 2 ** -10, 3 * 2 ** -11, 0.0, -0.0]),
                        (10.0 ** rng.uniform(-4, 3, 20000)).astype(np.float32)]).astype(np.float32)
    enc = torch.from_numpy(np.clip(x, -448, 448)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    np.savez_compressed(os.path.join(HERE, "fp8_torch.npz"), decode=dec, x=x, encode_clamped=enc)
    print("wrote fp8_torch.npz:", x.size, "inputs; torch", torch.__version__)


if __name__ == "__main__":
    main()
