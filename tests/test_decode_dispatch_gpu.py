"""The decode dispatcher's choices, shape by shape: which kernel, how many q heads per wavefront, which launch
mode -- as atoma_last_decode_kernel() names them after a real call.  The table of DESIGN.md 4.1 is this test's EXPECT; a change of a
default (the decode_mqk mask, decode_stream, the split heuristic) must change both."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes"))
pytestmark = pytest.mark.gpu

# (B, L, h, h_k, d, ragged) -> (kernel, q heads per wavefront G, mode)
EXPECT = [
    ((256, 4096, 32, 8, 128, False), ("paged_decode_mqk_kernel", 4, "balanced")),          # configs[1]: the headline
    ((256, 4096, 32, 8, 128, True), ("paged_decode_mqk_kernel", 4, "balanced")),           # ragged batch: same line
    ((64, 4096, 8, 1, 128, False), ("paged_decode_mqk_kernel", 8, "KV splits + combine")),  # one 70B TP = 8 rank
    ((1, 4096, 32, 8, 128, False), ("paged_decode_mqk_kernel", 4, "KV splits + combine")),  # one sequence
    ((16, 8192, 32, 8, 128, False), ("paged_decode_mqk_kernel", 4, "KV splits + combine")),
    ((256, 1024, 32, 8, 128, False), ("paged_decode_mqk_kernel", 4, "balanced")),
    ((256, 4096, 32, 32, 128, False), ("paged_decode_kernel", 1, "balanced")),             # MHA keeps the dot2 kernel
    ((256, 4096, 64, 8, 128, False), ("paged_decode_mqk_kernel", 8, "balanced")),          # 8 q heads per kv head
    ((256, 4096, 32, 8, 64, False), ("paged_decode_mqk_kernel(kv-head pairs)", 8, "balanced")),   # head_dim 64: two kv heads per wavefront (Llama-3.2-1B)
    ((8, 2048, 32, 8, 64, False), ("paged_decode_mqk_kernel(kv-head pairs)", 8, "KV splits + combine")),
    ((32, 1024, 9, 3, 64, False), ("paged_decode_kernel", 4, "KV splits + combine")),       # ... an odd number of kv heads: the dot2 kernel
    ((512, 512, 32, 8, 128, True), ("paged_decode_mqk_kernel", 4, "balanced")),            # configs[4]'s batch
    ((3, 257, 12, 2, 128, False), ("paged_decode_mqk_kernel", 8, "KV splits + combine")),   # 6 q heads per kv head -> one pass of 8
]


@pytest.mark.parametrize("shape,want", EXPECT, ids=[f"B{s[0]}xL{s[1]}_h{s[2]}_{s[3]}_d{s[4]}{'_ragged' if s[5] else ''}" for s, _ in EXPECT])
def test_decode_dispatch_table(gpu, shape, want):
    import decode_dispatch_table as T
    name = T.kernel_for(*shape)
    kernel, G, mode = want
    assert name.startswith(kernel + "<bf16,"), name
    assert f"D={shape[4]},G={G}," in name, name
    assert mode in name, name
    assert "nt" in name.split(",")        # non-temporal K/V loads are the default
