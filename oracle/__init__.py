"""CPU oracle for the atoma-infer paged-attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import or execute it, and there only as the checker / reported CPU
baseline -- never as the thing shipped or measured as the GPU path.

What it restates (reference = /root/reference, read-only, Rust + CUDA):

* attention result definition ........ csrc/tests/flash_attn_tests.rs:19-29 (``fa_acausal``)
* sequence-length / offset rules ..... csrc/kernels/block_info.h:11-39
* causal / tail masking, ALiBi ....... csrc/kernels/mask.h:110-209
* softmax numerics (exp2 domain) ..... csrc/kernels/softmax.h:65-185
* paged addressing ................... csrc/kernels/utils.h:296-314
* split-KV merge ..................... csrc/kernels/flash_fwd_kernel.h:1131-1313
* cache kernels ...................... csrc/kernels/cache_manager.cu:15-37,139-170
* swap offsets ....................... csrc/src/cache_manager.rs:18-128
* RMSNorm / RoPE call sites .......... models/src/llama.rs:146-251,402-474
  (arithmetic lives in candle-nn 0.9.2-alpha.1, not vendored: PARITY UNPINNED
  for those two ops -- no reference test holds a value for them)

Pinning: the attention oracle reproduces the reference's two golden tables
(csrc/tests/flash_attn_tests.rs:53-89 and models/src/flash_attention.rs:688-704)
exactly and satisfies its two equivalence properties; the cache oracles satisfy
the reference's bit-exact properties (csrc/tests/cache_manager_tests.rs).
``oracle/_ref`` (a build of the reference itself) does not exist: the reference
is Rust + CUDA + an un-vendored CUTLASS submodule and cannot be built here.
"""
