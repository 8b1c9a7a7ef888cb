// Workgroup -> (sequence, q head, query block) for the prefill kernels (prefill_mfma.hip, prefill_asm.hip).
#pragma once
#include "attn_params.h"

namespace atoma {

constexpr int PF_SGU = 8;              // (sequence, q head) units scheduled together on an XCD (2 kv groups at g = 4)

// Workgroup -> (sequence, q head, query block).  Measured dispatch policy of the chip
// (tools/probes/dispatch_probe.hip): workgroup L runs on XCD L % 8 and shader engine (L / 8) % 4 --
// both fixed by the index -- and on the first free CU of that engine, in index order.  So:
//  * units (sequence, q head) are dealt to XCDs in contiguous slices: all query blocks of all q heads
//    of a kv head read the same K/V (1 MiB at S = 2048) and stay behind one 4 MiB L2;
//  * inside an XCD, PF_SGU units at a time, blocks are ordered longest-first (causal: last block
//    first) across those units, and dealt to the 4 shader engines in snake order, so every engine
//    gets the same amount of work and its CUs take it in LPT order.
// The grid is padded to 8 * (largest slice) * m_blocks; surplus workgroups get `false`.
struct PfWork {
    int b, hq, mblk;
};
__device__ __forceinline__ bool pf_map_index(const AttnParams &p, int block_rows, int L, PfWork &w) {
    const int xcd = L & 7, i = L >> 3;
    const int m_blocks = (p.seqlen_q + block_rows - 1) / block_rows;
    const int n_units = p.b * p.h, uq = n_units >> 3, ur = n_units & 7;
    const int nu_x = uq + (xcd < ur ? 1 : 0);                                  // units of this XCD
    const int u0_x = xcd < ur ? xcd * (uq + 1) : ur * (uq + 1) + (xcd - ur) * uq;
    const int count_x = nu_x * m_blocks;
    if (i >= count_x) return false;
    int rank = i;
    {
        const int q4 = i >> 2, s4 = i & 3;
        if (q4 * 4 + 4 <= count_x) rank = q4 * 4 + ((q4 & 1) ? 3 - s4 : s4);     // snake over the 4 shader engines
    }
    const int sg_items = PF_SGU * m_blocks, sg = rank / sg_items, rr = rank - sg * sg_items;
    const int units_in_sg = min(PF_SGU, nu_x - sg * PF_SGU);
    const int mpos = rr / units_in_sg, uu = rr - mpos * units_in_sg;
    const int unit = u0_x + sg * PF_SGU + uu;
    w.b = unit / p.h;
    w.hq = unit - w.b * p.h;                                                     // q heads of a kv group are adjacent
    w.mblk = m_blocks - 1 - mpos;                                                // longest (most keys) first
    return true;
}

__device__ __forceinline__ bool pf_map_workgroup(const AttnParams &p, int block_rows, PfWork &w) {
    return pf_map_index(p, block_rows, (int)blockIdx.x, w);
}

}  // namespace atoma
