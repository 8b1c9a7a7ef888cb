"""Which decode kernel the dispatcher takes, shape by shape (atoma_last_decode_kernel()): the rows of DESIGN.md 4.1's table and of
tests/test_decode_dispatch_gpu.py.      python tools/probes/decode_dispatch_table.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
import numpy as np  # noqa: E402
import atoma_hip as ah  # noqa: E402

# (B, L, h, h_k, d, ragged): the shapes BASELINE.json's configurations and the tests produce
SHAPES = [
    (256, 4096, 32, 8, 128, False),   # configs[1] headline
    (256, 4096, 32, 8, 128, True),    # ragged batch
    (64, 4096, 8, 1, 128, False),     # 70B TP = 8 rank shard
    (1, 4096, 32, 8, 128, False),     # one sequence
    (16, 8192, 32, 8, 128, False),    # few long sequences
    (256, 1024, 32, 8, 128, False),
    (256, 4096, 32, 32, 128, False),  # MHA
    (256, 4096, 64, 8, 128, False),   # 8 q heads per kv head (70B on one GPU)
    (256, 4096, 32, 8, 64, False),    # Llama-3.2-1B head size
    (8, 2048, 32, 8, 64, False),
    (512, 512, 32, 8, 128, True),     # configs[4] batch, short contexts
    (3, 257, 12, 2, 128, False),
]


def kernel_for(B, L, h, hk, d, ragged, page=16, seed=0):
    rng = np.random.default_rng(seed)
    lens = (rng.integers(L // 2, L + 1, B) if ragged else np.full(B, L)).astype(np.int32)
    pps = (L + page - 1) // page
    nb = B * pps
    kc = ah.DeviceBuffer(nb * page * hk * d * 2)
    vc = ah.DeviceBuffer(nb * page * hk * d * 2)
    bt = ah.DeviceBuffer.from_numpy(np.arange(nb, dtype=np.int32).reshape(B, pps))
    q = ah.DeviceBuffer(B * h * d * 2)
    o = ah.DeviceBuffer(B * h * d * 2)
    dl = ah.DeviceBuffer.from_numpy(lens)
    ah.run_mha(q, kc, vc, o, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5, is_bf16=1,
               q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d),
               cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=bt, block_table_batch_stride=pps, page_block_size=page,
               force_split_kernel=True, unpadded_lse=False)
    ah.synchronize()
    name = (ah.lib.atoma_last_decode_kernel() or b"").decode()
    for b_ in (kc, vc, bt, q, o, dl):
        b_.free()
    return name


if __name__ == "__main__":
    ah.set_device(0)
    for s in SHAPES:
        print(s, "->", kernel_for(*s), flush=True)
