// FlashAttention-2 prefill, head_dim 128: the hand-scheduled instruction stream of tools/pfasm/kernel.py (generated into
// build/gen/pfa_*.inc by tools/gen_prefill_asm.py) behind a thin HIP kernel.
//
// Replaces /root/reference/csrc/kernels/flash_fwd_kernel.h:56-500 (compute_attn_1rowblock) and the seqlen_q > 1 use of the paged
// split-KV kernel (:504-1092) for the shapes the Llama configurations of BASELINE.json produce; everything else stays on
// prefill_mfma.hip.  Division of labour: this file maps a workgroup to its 256-row query block (prefill_map.h: XCD- and
// shader-engine-aware order, longest block first) and writes one parameter block per wavefront into LDS; the asm statement
// owns every vector / accumulator register and s36.. (clobber list PFA_CLOBBERS) from the first Q load to the last store of O.
// Structure, register map, schedule and numerics: tools/pfasm/kernel.py and DESIGN.md 4.2b; the same instruction list is
// executed by the functional simulator in tests/test_prefill_asm_sim.py.
//
// Two arithmetic variants per (storage type, K/V addressing):
//   fast  -- Q is multiplied by scale.log2(e) and rounded to the storage type once per block; scores leave the matrix pipe as
//            s~ - m, ready for v_exp_f32 (3 VALU instructions per score).  The rounding of Q moves a probability by ~2^-9
//            relative: invisible once a row's softmax mass is spread over hundreds of keys (the 1e-3 tolerance of BASELINE.json),
//            not for the first rows of a causal sequence;
//   exact -- Q untouched, one v_mul_f32 in front of every v_exp_f32: the reference's arithmetic (softmax.h:65-185) to the last
//            rounding.  Chosen per query block when its first row sees fewer than `prefill_exact_keys` (512) keys.
#include "attn_params.h"
#include "prefill_map.h"
#include "prefill_asm_gen.h"
#include <atomic>
#include <stdlib.h>

namespace atoma {

constexpr int PFA_BM = 256;   // query rows per workgroup
constexpr int PFA_BIG = 0x3FFFFFFF;

// one 32-row block of the workgroup's 256 rows
struct PfaBlock {
    int r0, rows, n;
};

template <bool BF16, bool PAGED>
__global__ void __launch_bounds__(256) prefill_asm_kernel(const AttnParams p, const int exact_keys, const int simple) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // K ring | V ring | 4 parameter blocks (the only LDS of the kernel: base 0)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PfWork wk;
    if (!pf_map_workgroup(p, PFA_BM, wk)) return;
    const int b = wk.b, hq = wk.hq, mblk = wk.mblk;
    const int hk = hq / (p.h / p.h_k);
    const SeqInfo si(p, b);
    const int m0 = mblk * PFA_BM;
    if (m0 >= si.len_q) return;
    const int len_q = si.len_q, len_k = si.len_k, shift = len_k - len_q;
    const bool causal = p.is_causal != 0;

    auto block = [&](int j) {
        PfaBlock x;
        x.r0 = m0 + 32 * j;
        x.rows = min(max(len_q - x.r0, 0), 32);
        x.n = 0;
        if (x.rows > 0) {
            const int last = x.r0 + x.rows - 1;
            const int vis = causal ? min(len_k, last + shift + 1) : len_k;
            x.n = vis > 0 ? (vis + 63) / 64 : 0;
        }
        return x;
    };
    // slots of this wavefront: blocks w and 7 - w, the one that needs fewer K/V tiles first
    const PfaBlock ba = block(wave), bb = block(7 - wave);
    const bool a_first = ba.n < bb.n || (ba.n == bb.n && ba.r0 <= bb.r0);
    const PfaBlock s0 = a_first ? ba : bb, s1 = a_first ? bb : ba;
    int n_tiles;
    {
        const int last = min(m0 + PFA_BM, len_q) - 1;
        const int vis = causal ? min(len_k, last + shift + 1) : len_k;
        n_tiles = vis > 0 ? (vis + 63) / 64 : 0;
    }
    int n0 = s0.n;
    const int n1 = s1.n;
    if (simple && n0 > 0) n0 = n1;
    // fewest keys a row of this workgroup sees -> arithmetic variant
    const int min_keys = causal ? min(len_k, m0 + shift + 1) : len_k;
    const bool exact = __builtin_amdgcn_readfirstlane((int)(min_keys < exact_keys)) != 0;   // provably uniform: the branch below must not be predicated

    uint32_t *par = reinterpret_cast<uint32_t *>(smem + PFA_LDS_PARAMS) + wave * PFA_PARAM_DWORDS;
    if ((tid & 63) == 0) {
        auto put64 = [&](int idx, uint64_t v) { par[idx] = (uint32_t)v; par[idx + 1] = (uint32_t)(v >> 32); };
        const int64_t q_off = si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)hq * p.q_head_stride;
        const int64_t o_off = si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)hq * p.o_head_stride;
        const PfaBlock sl[2] = {s0, s1};
        int tm[2];
        for (int s = 0; s < 2; ++s) {
            put64(PFA_q0_lo + 2 * s, (uint64_t)(uintptr_t)(p.q + q_off + (int64_t)sl[s].r0 * p.q_row_stride));
            put64(PFA_o0_lo + 2 * s, (uint64_t)(uintptr_t)(p.o + o_off + (int64_t)sl[s].r0 * p.o_row_stride));
            uint64_t lse = 0;
#ifndef PFA_TIMING
            if (p.lse) {
                const float *l = (p.unpadded_lse && p.cu_seqlens_q) ? p.lse + (int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q
                                                                     : p.lse + ((int64_t)b * p.h + hq) * p.seqlen_q;
                lse = (uint64_t)(uintptr_t)(l + sl[s].r0);
            }
#endif
            put64(PFA_lse0_lo + 2 * s, lse);
            par[PFA_rows0 + s] = (uint32_t)sl[s].rows;
            const int lim = causal ? sl[s].r0 + shift : len_k - 1;
            par[PFA_lim0 + s] = (uint32_t)lim;
            const int minlim = min(lim, len_k - 1);
            tm[s] = minlim >= 0 ? max(0, (minlim + 1) >> 6) : 0;
        }
        par[PFA_q_stride] = (uint32_t)(p.q_row_stride * 2);
        par[PFA_o_stride] = (uint32_t)(p.o_row_stride * 2);
        if (PAGED) {
            put64(PFA_k_lo, (uint64_t)(uintptr_t)(p.k + (int64_t)hk * p.k_head_stride));
            put64(PFA_v_lo, (uint64_t)(uintptr_t)(p.v + (int64_t)hk * p.v_head_stride));
            put64(PFA_bt_lo, (uint64_t)(uintptr_t)(p.block_table + (int64_t)b * p.block_table_batch_stride));
            par[PFA_page_shift] = (uint32_t)__builtin_ctz(p.page_size);
            par[PFA_k_page_bytes] = (uint32_t)(p.k_batch_stride * 2);
            par[PFA_v_page_bytes] = (uint32_t)(p.v_batch_stride * 2);
        } else {
            put64(PFA_k_lo, (uint64_t)(uintptr_t)(p.k + (int64_t)hk * p.k_head_stride + si.k_offset(p.k_batch_stride, p.k_row_stride, b)));
            put64(PFA_v_lo, (uint64_t)(uintptr_t)(p.v + (int64_t)hk * p.v_head_stride + si.k_offset(p.v_batch_stride, p.v_row_stride, b)));
        }
        par[PFA_k_stride] = (uint32_t)(p.k_row_stride * 2);
        par[PFA_v_stride] = (uint32_t)(p.v_row_stride * 2);
        par[PFA_k_bytes] = len_k > 0 ? (uint32_t)((int64_t)(len_k - 1) * p.k_row_stride * 2 + 256) : 0u;
        par[PFA_v_bytes] = len_k > 0 ? (uint32_t)((int64_t)(len_k - 1) * p.v_row_stride * 2 + 256) : 0u;
        par[PFA_len_k] = (uint32_t)len_k;
        par[PFA_n_tiles] = (uint32_t)n_tiles;
        par[PFA_n0] = (uint32_t)n0;
        par[PFA_n1] = (uint32_t)n1;
        const int tm0 = n0 > 0 ? tm[0] : PFA_BIG, tm1 = n1 > 0 ? tm[1] : PFA_BIG, tmm = min(tm0, tm1);
        par[PFA_tm0] = (uint32_t)tm0;
        par[PFA_tm1] = (uint32_t)tm1;
        par[PFA_tmm] = (uint32_t)tmm;
        par[PFA_n_steady] = (uint32_t)max(0, min(n0, tmm) - 1);
        par[PFA_lim_step] = causal ? 1u : 0u;
        par[PFA_scale_log2] = __float_as_uint(p.scale_log2);
        par[PFA_wave] = (uint32_t)wave;
        par[PFA_thr] = __float_as_uint(exact ? 8.0f / p.scale_log2 : 8.0f);
#ifdef PFA_TIMING   // phase timers (make timing; tools/probes/pfa_phases.py): 8 dwords per wavefront into the LSE array
        put64(PFA_dbg_lo, p.lse ? (uint64_t)(uintptr_t)(p.lse + ((int64_t)blockIdx.x * 4 + wave) * 8) : 0);
#else
        put64(PFA_dbg_lo, 0);
#endif
    }
    // the wavefront reads its own block back (LDS operations of one wavefront stay in order): no barrier
    const uint32_t paddr = __builtin_amdgcn_readfirstlane((uint32_t)(PFA_LDS_PARAMS + wave * PFA_PARAM_DWORDS * 4));
    // one statement per variant; `if constexpr` keeps only this instantiation's two (fast / exact) in the code object
    if constexpr (BF16 && PAGED) {
        if (exact) asm volatile(
#include "pfa_bf16_paged_exact.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
        else asm volatile(
#include "pfa_bf16_paged_fast.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
    } else if constexpr (BF16) {
#ifdef PFA_TIMING
        if (exact) asm volatile(
#include "pfa_bf16_contig_exact_timing.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
        else asm volatile(
#include "pfa_bf16_contig_fast_timing.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
#else
        if (exact) asm volatile(
#include "pfa_bf16_contig_exact.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
        else asm volatile(
#include "pfa_bf16_contig_fast.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
#endif
    } else if constexpr (PAGED) {
        if (exact) asm volatile(
#include "pfa_f16_paged_exact.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
        else asm volatile(
#include "pfa_f16_paged_fast.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
    } else {
        if (exact) asm volatile(
#include "pfa_f16_contig_exact.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
        else asm volatile(
#include "pfa_f16_contig_fast.inc"
                : : "s"(paddr) : PFA_CLOBBERS);
    }
}

// atoma_set_option("prefill_exact_keys", n): query blocks whose first row sees fewer than n keys take the exact arithmetic
// (default 512; 0 = never, 0x7fffffff = always).  "prefill_simple" = 1: both 32-row slots of a wavefront run until the later
// one is done (masked), instead of dropping the earlier slot's MFMAs on the causal diagonal (A/B knob).
std::atomic<int> prefill_exact_keys{512};
std::atomic<int> prefill_simple{0};

bool prefill_asm_supported(const AttnParams &p) {
    if (p.d != 128 || p.alibi_slopes != nullptr || p.seqlen_q <= 1) return false;
    if (p.block_table) {
        if (p.page_size < 16 || (p.page_size & (p.page_size - 1)) != 0) return false;           // pages of 2^k >= 16 tokens
        if (p.k_batch_stride * 2 >= (int64_t)1 << 32 || p.v_batch_stride * 2 >= (int64_t)1 << 32) return false;
    }
    // 32-bit byte offsets inside a sequence (buffer descriptors): rows x stride below 2 GiB
    const int64_t worst = (int64_t)p.seqlen_k * 2 * (p.k_row_stride > p.v_row_stride ? p.k_row_stride : p.v_row_stride);
    if (!p.block_table && worst >= ((int64_t)1 << 31)) return false;
    if ((int64_t)PFA_BM * 2 * (p.q_row_stride > p.o_row_stride ? p.q_row_stride : p.o_row_stride) >= ((int64_t)1 << 31)) return false;
    return true;
}

template <bool BF16, bool PAGED>
static void launch_pfa(const AttnParams &p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_asm_kernel<BF16, PAGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  PFA_LDS_TOTAL);
        attr_set = true;
    }
    const int64_t m_blocks = cdiv(p.seqlen_q, PFA_BM), n_units = (int64_t)p.b * p.h;
    const int64_t nu_max = n_units / 8 + (n_units % 8 ? 1 : 0);
    dim3 grid((unsigned)(8 * nu_max * m_blocks));   // padded: see prefill_map.h
    static const int env_exact = [] { const char *e = getenv("ATOMA_PREFILL_EXACT_KEYS"); return e ? atoi(e) : -1; }();
    const int exact_keys = env_exact >= 0 ? env_exact : prefill_exact_keys.load();
    hipLaunchKernelGGL((prefill_asm_kernel<BF16, PAGED>), grid, dim3(256), PFA_LDS_TOTAL, stream, p, exact_keys, prefill_simple.load());
    ATOMA_CHECK_LAUNCH("prefill_asm_kernel");
}

void launch_prefill_asm(const AttnParams &p, bool is_bf16, hipStream_t stream) {
    if (p.b <= 0 || p.h <= 0 || p.seqlen_q <= 0) return;
    const bool paged = p.block_table != nullptr;
    if (is_bf16) { if (paged) launch_pfa<true, true>(p, stream); else launch_pfa<true, false>(p, stream); }
    else { if (paged) launch_pfa<false, true>(p, stream); else launch_pfa<false, false>(p, stream); }
}

}  // namespace atoma
