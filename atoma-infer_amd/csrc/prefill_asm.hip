// FlashAttention-2 prefill, head_dim 128: the hand-scheduled instruction stream of tools/pfasm/kernel.py (generated into
// build/gen/pfa_*.inc by tools/gen_prefill_asm.py) behind thin HIP kernels.
//
// Replaces /root/reference/csrc/kernels/flash_fwd_kernel.h:56-500 (compute_attn_1rowblock) and the seqlen_q > 1 use of the paged
// split-KV kernel (:504-1092) for the shapes the Llama configurations of BASELINE.json produce; everything else stays on
// prefill_mfma.hip.  Structure, register map, schedule and numerics: tools/pfasm/kernel.py and DESIGN.md 4.2b; the same instruction
// list is executed by the functional simulator in tests/test_prefill_asm_sim.py.
//
//   contiguous K/V:  pfa_plan_kernel writes one 256-byte entry per (256-row query block, wavefront) -- descriptors, tile counts,
//                    mask limits, arithmetic flag -- in the XCD-aware, longest-block-first order of prefill_map.h; then ONE
//                    workgroup per CU (prefill_asm_persistent) walks its share of the table: the next block's entry, Q rows and
//                    first K/V tiles are requested around the current block's output stores, so launch latency, parameter
//                    arithmetic and the first round trips to memory are paid once per CU, not once per block;
//   paged K/V:       one workgroup per query block (prefill_asm_paged), the same entries written to LDS by its own preamble.
//
// Two arithmetic variants live in every instruction stream and are chosen per query block (entry flag):
//   fast  -- Q is multiplied by scale.log2(e) and rounded to the storage type once per block; scores leave the matrix pipe as
//            s~ - m, ready for v_exp_f32 (3 VALU instructions per score).  The rounding of Q moves a probability by ~2^-9
//            relative: invisible once a row's softmax mass is spread over hundreds of keys (the 1e-3 tolerance of BASELINE.json),
//            not for the first rows of a causal sequence;
//   exact -- Q untouched, one v_mul_f32 in front of every v_exp_f32: the reference's arithmetic (softmax.h:65-185) to the last
//            rounding.  Chosen when the block's first row sees fewer than `prefill_exact_keys` (512) keys.
#include "attn_params.h"
#include "prefill_map.h"
#include "prefill_asm_gen.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>

namespace atoma {

void *workspace(hipStream_t stream, size_t bytes);   // runtime.hip: grow-only per-stream scratch

constexpr int PFA_BM = 256;   // query rows per workgroup
constexpr int PFA_BIG = 0x3FFFFFFF;

struct PfaBlock {             // one 32-row block of the workgroup's 256 rows
    int r0, rows, n;
};

// The entry of (work item wk, wavefront `wave`): tools/pfasm/harness.py wave_entry() is the same arithmetic in Python.
__device__ __forceinline__ void pfa_fill_entry(const AttnParams &p, const PfWork &wk, int wave, int exact_keys, int simple, uint32_t *e) {
    const int b = wk.b < 0 ? 0 : wk.b, hq = wk.hq, hk = hq / (p.h / p.h_k);
    const SeqInfo si(p, b);
    const int m0 = wk.mblk * PFA_BM;
    const int len_q = si.len_q, len_k = si.len_k, shift = len_k - len_q;
    const bool causal = p.is_causal != 0, paged = p.block_table != nullptr;
    for (int i = 0; i < PFA_PARAM_DWORDS; ++i) e[i] = 0;
    // the launch-wide strides sit in every entry (the kernel derives its lane constants from the first one it reads)
    e[PFA_q_stride] = (uint32_t)(p.q_row_stride * 2);
    e[PFA_o_stride] = (uint32_t)(p.o_row_stride * 2);
    e[PFA_k_stride] = (uint32_t)(p.k_row_stride * 2);
    e[PFA_v_stride] = (uint32_t)(p.v_row_stride * 2);
    e[PFA_k_tile] = (uint32_t)(p.k_row_stride * 128);
    e[PFA_v_tile] = (uint32_t)(p.v_row_stride * 128);
    if (wk.b < 0 || m0 >= len_q) return;                       // flags = 0: an empty entry, skipped by the kernel
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
    const PfaBlock sl[2] = {a_first ? ba : bb, a_first ? bb : ba};
    int n_tiles;
    {
        const int last = min(m0 + PFA_BM, len_q) - 1;
        const int vis = causal ? min(len_k, last + shift + 1) : len_k;
        n_tiles = vis > 0 ? (vis + 63) / 64 : 0;
    }
    int n0 = sl[0].n;
    const int n1 = sl[1].n;
    if (simple && n0 > 0) n0 = n1;
    const int min_keys = causal ? min(len_k, m0 + shift + 1) : len_k;   // fewest keys a row of this block sees
    const bool exact = min_keys < exact_keys;
    auto put64 = [&](int idx, uint64_t v) { e[idx] = (uint32_t)v; e[idx + 1] = (uint32_t)(v >> 32); };
    const int64_t q_off = si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)hq * p.q_head_stride;
    const int64_t o_off = si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)hq * p.o_head_stride;
    const uint32_t o_stride = (uint32_t)(p.o_row_stride * 2);
    int tm[2];
    for (int s = 0; s < 2; ++s) {
        put64(PFA_q0_lo + 2 * s, (uint64_t)(uintptr_t)(p.q + q_off + (int64_t)sl[s].r0 * p.q_row_stride));
        put64(PFA_o0_lo + 4 * s, (uint64_t)(uintptr_t)(p.o + o_off + (int64_t)sl[s].r0 * p.o_row_stride));
        e[PFA_o0_bytes + 4 * s] = sl[s].rows ? (uint32_t)(sl[s].rows - 1) * o_stride + 256u : 0u;
        e[PFA_o0_flags + 4 * s] = 0x00020000u;
        uint64_t lse = 0;
#ifndef PFA_TIMING
        if (p.lse) {
            const float *l = (p.unpadded_lse && p.cu_seqlens_q) ? p.lse + (int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q
                                                                 : p.lse + ((int64_t)b * p.h + hq) * p.seqlen_q;
            lse = (uint64_t)(uintptr_t)(l + sl[s].r0);
        }
#endif
        put64(PFA_lse0_lo + 2 * s, lse);
        e[PFA_rows0 + s] = (uint32_t)sl[s].rows;
        const int lim = causal ? sl[s].r0 + shift : len_k - 1;
        e[PFA_lim0 + s] = (uint32_t)lim;
        const int minlim = min(lim, len_k - 1);
        tm[s] = minlim >= 0 ? max(0, (minlim + 1) >> 6) : 0;
    }
    e[PFA_q_stride] = (uint32_t)(p.q_row_stride * 2);
    e[PFA_o_stride] = o_stride;
    if (paged) {
        put64(PFA_k_lo, (uint64_t)(uintptr_t)(p.k + (int64_t)hk * p.k_head_stride));
        put64(PFA_v_lo, (uint64_t)(uintptr_t)(p.v + (int64_t)hk * p.v_head_stride));
        put64(PFA_bt_lo, (uint64_t)(uintptr_t)(p.block_table + (int64_t)b * p.block_table_batch_stride));
        e[PFA_page_shift] = (uint32_t)__builtin_ctz(p.page_size);
        e[PFA_k_page] = (uint32_t)(p.k_batch_stride * 2);
        e[PFA_v_page] = (uint32_t)(p.v_batch_stride * 2);
    } else {
        put64(PFA_k_lo, (uint64_t)(uintptr_t)(p.k + (int64_t)hk * p.k_head_stride + si.k_offset(p.k_batch_stride, p.k_row_stride, b)));
        put64(PFA_v_lo, (uint64_t)(uintptr_t)(p.v + (int64_t)hk * p.v_head_stride + si.k_offset(p.v_batch_stride, p.v_row_stride, b)));
    }
    e[PFA_k_stride] = (uint32_t)(p.k_row_stride * 2);
    e[PFA_v_stride] = (uint32_t)(p.v_row_stride * 2);
    e[PFA_k_tile] = (uint32_t)(p.k_row_stride * 128);
    e[PFA_v_tile] = (uint32_t)(p.v_row_stride * 128);
    e[PFA_k_bytes] = len_k > 0 ? (uint32_t)((int64_t)(len_k - 1) * p.k_row_stride * 2 + 256) : 0u;
    e[PFA_v_bytes] = len_k > 0 ? (uint32_t)((int64_t)(len_k - 1) * p.v_row_stride * 2 + 256) : 0u;
    e[PFA_k_flags] = 0x00020000u;
    e[PFA_v_flags] = 0x00020000u;
    e[PFA_len_k] = (uint32_t)len_k;
    e[PFA_n_tiles] = (uint32_t)n_tiles;
    e[PFA_n0] = (uint32_t)n0;
    e[PFA_n1] = (uint32_t)n1;
    const int tm0 = n0 > 0 ? tm[0] : PFA_BIG, tm1 = n1 > 0 ? tm[1] : PFA_BIG, tmm = min(tm0, tm1);
    e[PFA_tm0] = (uint32_t)tm0;
    e[PFA_tm1] = (uint32_t)tm1;
    e[PFA_tmm] = (uint32_t)tmm;
    e[PFA_n_steady] = (uint32_t)max(0, min(n0, tmm) - 1);
    e[PFA_lim_step] = causal ? 1u : 0u;
    e[PFA_scale_log2] = __float_as_uint(p.scale_log2);
    e[PFA_thr] = __float_as_uint(exact ? 8.0f / p.scale_log2 : 8.0f);
    e[PFA_mscale] = __float_as_uint(exact ? p.scale_log2 : 1.0f);
    e[PFA_flags] = (exact ? 1u : 0u) | 2u;
    e[PFA_wave] = (uint32_t)wave;
}

// one thread per (table entry, wavefront)
__global__ void __launch_bounds__(256) pfa_plan_kernel(const AttnParams p, uint32_t *tab, int n_entries, int exact_keys, int simple) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n_entries * 4) return;
    const int L = t >> 2, wave = t & 3;
    uint32_t e[PFA_PARAM_DWORDS];
    PfWork wk;
    if (!pf_map_index(p, PFA_BM, L, wk)) { wk.b = -1; wk.hq = 0; wk.mblk = 0; }   // the padded tail of an XCD's list
    pfa_fill_entry(p, wk, wave, exact_keys, simple, e);
    uint4 *dst = reinterpret_cast<uint4 *>(tab + (size_t)t * PFA_PARAM_DWORDS);
#pragma unroll
    for (int i = 0; i < PFA_PARAM_DWORDS / 4; ++i) dst[i] = make_uint4(e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]);
}

// One workgroup per CU.  Workgroup g sits on XCD g % 8 and takes the entries L = 8 i + g % 8 of that XCD's list; with nw workgroups
// per XCD and w = g / 8 its positions are i = nw r + (r even ? w : nw - 1 - w), r = 0, 1, ..: the list is longest-block-first, so a
// workgroup that got an early (long) block of one round gets a late (short) one of the next.
template <bool BF16>
__global__ void __launch_bounds__(256) prefill_asm_persistent(const uint32_t *tab, const int n_entries, uint32_t *dbg) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // K ring | V ring | staging (the only LDS of the kernel: base 0)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t first = blockIdx.x;
    const uint32_t nw = gridDim.x >> 3, w = blockIdx.x >> 3;
    const uint32_t g_even = 8u * (2u * nw - 1u - 2u * w), g_odd = 8u * (1u + 2u * w);
    const uint64_t tabp = (uint64_t)(uintptr_t)tab;
    const uint64_t dbgp = dbg ? (uint64_t)(uintptr_t)(dbg + ((size_t)blockIdx.x * 4 + wave) * 8) : 0;
    if constexpr (BF16) {
#if defined(PFA_TIMING) && PFA_TIMING == 2
        asm volatile(
#include "pfa_bf16_contig_timing_block.inc"
            : : "s"(tabp), "s"(first), "s"((uint32_t)n_entries), "s"(g_even), "s"(wave), "s"(dbgp), "s"(g_odd) : PFA_CLOBBERS);
#elif defined(PFA_TIMING)
        asm volatile(
#include "pfa_bf16_contig_timing.inc"
            : : "s"(tabp), "s"(first), "s"((uint32_t)n_entries), "s"(g_even), "s"(wave), "s"(dbgp), "s"(g_odd) : PFA_CLOBBERS);
#else
        asm volatile(
#include "pfa_bf16_contig.inc"
            : : "s"(tabp), "s"(first), "s"((uint32_t)n_entries), "s"(g_even), "s"(wave), "s"(dbgp), "s"(g_odd) : PFA_CLOBBERS);
#endif
    } else {
        asm volatile(
#include "pfa_f16_contig.inc"
            : : "s"(tabp), "s"(first), "s"((uint32_t)n_entries), "s"(g_even), "s"(wave), "s"(dbgp), "s"(g_odd) : PFA_CLOBBERS);
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256) prefill_asm_paged(const AttnParams p, const int exact_keys, const int simple) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PfWork wk;
    if (!pf_map_workgroup(p, PFA_BM, wk)) return;
    {
        const SeqInfo si(p, wk.b);
        if (wk.mblk * PFA_BM >= si.len_q) return;
    }
    uint32_t *par = reinterpret_cast<uint32_t *>(smem + PFA_LDS_PARAMS) + wave * PFA_PARAM_DWORDS;
    if ((tid & 63) == 0) pfa_fill_entry(p, wk, wave, exact_keys, simple, par);
    // the wavefront reads its own entry back (LDS operations of one wavefront stay in order): no barrier
    const uint32_t paddr = __builtin_amdgcn_readfirstlane((uint32_t)(PFA_LDS_PARAMS + wave * PFA_PARAM_DWORDS * 4));
    if constexpr (BF16) {
        asm volatile(
#include "pfa_bf16_paged.inc"
            : : "s"(paddr) : PFA_CLOBBERS);
    } else {
        asm volatile(
#include "pfa_f16_paged.inc"
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
    if (!(p.scale_log2 > 0.f) || !(p.scale_log2 < INFINITY)) return false;   // the deferred-raise threshold is 8 / scale_log2: positive finite scales only
    if (p.block_table) {
        if (p.page_size < 16 || (p.page_size & (p.page_size - 1)) != 0) return false;           // pages of 2^k >= 16 tokens
        if (p.k_batch_stride * 2 >= (int64_t)1 << 32 || p.v_batch_stride * 2 >= (int64_t)1 << 32) return false;
    }
    // 32-bit byte offsets inside a sequence (buffer descriptors, the tile offset in soffset): rows x stride below 2 GiB
    const int64_t worst = ((int64_t)p.seqlen_k + 512) * 2 * (p.k_row_stride > p.v_row_stride ? p.k_row_stride : p.v_row_stride);
    if (!p.block_table && worst >= ((int64_t)1 << 31)) return false;
    if ((int64_t)PFA_BM * 2 * (p.q_row_stride > p.o_row_stride ? p.q_row_stride : p.o_row_stride) >= ((int64_t)1 << 31)) return false;
    return true;
}

static int pfa_exact_keys() {
    static const int env_exact = [] { const char *e = getenv("ATOMA_PREFILL_EXACT_KEYS"); return e ? atoi(e) : -1; }();
    return env_exact >= 0 ? env_exact : prefill_exact_keys.load();
}

// The 160 KiB dynamic-LDS opt-in of the four kernels, once per DEVICE (the reference drives every GPU of the node from one process,
// one thread each: model_executor.rs:428-440), return code checked, never first set inside a capture when atoma_warmup ran
// (prefill_asm_prepare below).
static bool pfa_attrs_for_device() {
    constexpr int MAX_DEV = 64;
    static std::mutex mu;
    static bool done[MAX_DEV] = {};
    int dev = 0;
    if (!check_hip(hipGetDevice(&dev), "hipGetDevice")) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < MAX_DEV && done[dev]) return true;
    const void *kernels[4] = {reinterpret_cast<const void *>(&prefill_asm_paged<true>), reinterpret_cast<const void *>(&prefill_asm_paged<false>),
                              reinterpret_cast<const void *>(&prefill_asm_persistent<true>), reinterpret_cast<const void *>(&prefill_asm_persistent<false>)};
    for (const void *k : kernels)
        if (!check_hip(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PFA_LDS_TOTAL), "prefill_asm: 160 KiB dynamic LDS opt-in")) return false;
    if (dev >= 0 && dev < MAX_DEV) done[dev] = true;
    return true;
}
bool prefill_asm_prepare() { return pfa_attrs_for_device(); }

// entries of the persistent kernel's plan table for a call with `seqs` sequences, `heads` q heads and a longest sequence of `max_seqlen_q` rows:
// the table is padded to (units rounded up to 8) x (256-row blocks of the LONGEST sequence) -- prefill_map.h.  ONE formula for launch_pfa and
// atoma_warmup_prefill (the warm-up's own estimate, tokens / 256 + seqs blocks, was far too small for ragged batches).
static int64_t pfa_table_entries(int64_t seqs, int64_t heads, int64_t max_seqlen_q) {
    const int64_t n_units = seqs * heads;
    return 8 * cdiv(n_units, 8) * cdiv(max_seqlen_q, PFA_BM);
}
size_t prefill_asm_workspace_bound(int64_t max_seqlen_q, int64_t seqs, int64_t heads) {
    return (size_t)pfa_table_entries(seqs, heads, max_seqlen_q) * 4 * PFA_PARAM_DWORDS * 4;
}

// The fast arithmetic rounds q . scale . log2(e) to the storage type once per block (header of this file): safe while that factor neither
// overflows f16 nor pushes small q into its subnormals -- otherwise every block takes the exact variant (the reference's arithmetic,
// softmax.h:65-91).  (A scale that is zero, negative or not finite never gets here: prefill_asm_supported.)
template <bool BF16> static int pfa_exact_keys_for(const AttnParams &p) {
    const bool fast_ok = BF16 ? (p.scale_log2 >= 0x1p-20f && p.scale_log2 <= 0x1p20f) : (p.scale_log2 >= 0x1p-6f && p.scale_log2 <= 1.0f);
    return fast_ok ? pfa_exact_keys() : 0x7fffffff;
}

template <bool BF16>
static int launch_pfa(const AttnParams &p, hipStream_t stream) {
    const int64_t n_entries = pfa_table_entries(p.b, p.h, p.seqlen_q);   // padded: see prefill_map.h
    if (!pfa_attrs_for_device()) return -1;
    const int exact_keys = pfa_exact_keys_for<BF16>(p);
    if (p.block_table) {
        hipLaunchKernelGGL((prefill_asm_paged<BF16>), dim3((unsigned)n_entries), dim3(256), PFA_LDS_TOTAL, stream, p, exact_keys, prefill_simple.load());
        return ATOMA_CHECK_LAUNCH("prefill_asm_paged") ? 0 : -1;
    }
    uint32_t *tab = static_cast<uint32_t *>(workspace(stream, (size_t)n_entries * 4 * PFA_PARAM_DWORDS * 4));
    if (!tab) return -1;   // workspace() has set the error (e.g. first use inside a graph capture: atoma_warmup / atoma_warmup_prefill first)
    hipLaunchKernelGGL(pfa_plan_kernel, dim3((unsigned)cdiv(n_entries * 4, 256)), dim3(256), 0, stream, p, tab, (int)n_entries, exact_keys, prefill_simple.load());
    if (!ATOMA_CHECK_LAUNCH("pfa_plan_kernel")) return -1;
    int g = device_num_cus() & ~7;                     // one workgroup per CU (160 KiB of LDS each), a multiple of the 8 XCDs
    if (g < 8) g = 8;
    if ((int64_t)g > n_entries) g = (int)n_entries;    // n_entries is a multiple of 8
    uint32_t *dbg = nullptr;
#ifdef PFA_TIMING
    dbg = reinterpret_cast<uint32_t *>(p.lse);
#endif
    hipLaunchKernelGGL((prefill_asm_persistent<BF16>), dim3((unsigned)g), dim3(256), PFA_LDS_TOTAL, stream, tab, (int)n_entries, dbg);
    return ATOMA_CHECK_LAUNCH("prefill_asm_persistent") ? 0 : -1;
}

// 0 = launched; -1 = not launched (the error is in atoma_last_error)
int launch_prefill_asm(const AttnParams &p, bool is_bf16, hipStream_t stream) {
    if (p.b <= 0 || p.h <= 0 || p.seqlen_q <= 0) return 0;
    return is_bf16 ? launch_pfa<true>(p, stream) : launch_pfa<false>(p, stream);
}

}  // namespace atoma
