// Host-side batch preparation of one engine step: the tensors ModelWorker::prepare_input_tensors
// builds from the scheduler's sequence metadata (/root/reference/backends/vllm/src/worker.rs:224-460) -- token ids,
// positions, slot mapping, sequence / context lengths, cumulative query and sequence starts, padded block tables --
// packed into ONE staging buffer and sent with ONE host-to-device copy.  The reference creates each of them as its own
// Candle tensor (one small H2D copy each) and builds the padded block table from one [1, max_len] tensor PER SEQUENCE
// concatenated on the device (utils::make_tensor_with_pad, worker.rs:670-683): B + 10 copies and a concat per step.
// Pure integer work, bit-exact against the oracle (oracle/batch_prep_oracle.py); no reference test pins it.
// The step numbers in the comments are the reference's own ("// 1. Context length" ...).
#include "common.h"
#include <algorithm>
#include <string.h>
#include <vector>

namespace atoma {

static const int64_t PAD_SLOT_ID = -1;   // worker.rs:13
static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

}  // namespace atoma

namespace atoma { void note_decode_lengths(int64_t min_len, int64_t max_len, int64_t count); }   // paged_decode.hip

extern "C" int atoma_prepare_inputs(const atoma_seq_desc *seqs, int64_t num_sequences, int64_t block_size, int64_t sliding_window,
                                    int enable_chunked_prefill, void *host_staging, int64_t host_capacity, void *device_buffer,
                                    int64_t device_capacity, atoma_batch_layout *layout, void *stream) {
    using namespace atoma;
    clear_error();
    if (!layout) { set_error("prepare_inputs: layout must not be null"); return -1; }
    if (num_sequences <= 0 || !seqs) { set_error("prepare_inputs: the batch is empty"); return -1; }   // worker.rs:411 unwraps a max over the batch
    if (block_size <= 0) { set_error("prepare_inputs: block_size must be positive"); return -1; }
    const bool has_sw = sliding_window > 0;

    // ---- pass 1: sizes (steps 1, 2, 4, 6, 7) ----
    struct Seq { int64_t context, seq_len, query, sliding_seq_len, bt_start, bt_len, slots; };
    std::vector<Seq> info((size_t)num_sequences);
    atoma_batch_layout L;
    memset(&L, 0, sizeof L);
    L.num_sequences = num_sequences;
    int64_t min_decode_seq_len = INT64_MAX;
    for (int64_t i = 0; i < num_sequences; ++i) {
        const atoma_seq_desc &s = seqs[i];
        Seq &q = info[(size_t)i];
        if (s.length < 0 || s.num_computed_tokens < 0 || s.token_chunk_size < 0 || s.block_table_len < 0) {
            set_error("prepare_inputs: negative length in a sequence descriptor");
            return -1;
        }
        if (!s.is_prompt && s.length == 0) {   // worker.rs:267-272
            set_error("Empty prompts should not be received in `ModelWorker`");
            return -1;
        }
        if (!s.is_prompt && s.token_chunk_size < 1) { set_error("prepare_inputs: a decode sequence is scheduled with one token"); return -1; }
        q.context = s.is_prompt ? s.num_computed_tokens : s.length - 1;                    // 1.
        q.seq_len = std::min(s.length, q.context + s.token_chunk_size);                    // 2.
        if (s.is_prompt && q.seq_len < q.context) { set_error("prepare_inputs: computed tokens exceed the sequence length"); return -1; }
        q.query = s.is_prompt ? q.seq_len - q.context : 1;                                 // 4.
        q.sliding_seq_len = (has_sw && !s.is_prompt) ? std::min(sliding_window, q.seq_len) : q.seq_len;   // 5.
        q.bt_start = 0;
        q.bt_len = 0;
        if (enable_chunked_prefill || !s.is_prompt) {                                      // 6.
            if (!s.block_table) {
                set_error("Block table should be allocated for sequence on decoding phase");
                return -1;
            }
            q.bt_len = s.block_table_len;
            if (has_sw) {                                                                  // 7.
                const int64_t sw_blocks = (sliding_window + block_size - 1) / block_size;
                q.bt_start = std::max<int64_t>(0, q.bt_len - sw_blocks);
                q.bt_len -= q.bt_start;
            }
        }
        const int64_t ntok = s.is_prompt ? q.query : 1;                                    // 3.
        if (ntok > 0 && !s.token_ids) { set_error("prepare_inputs: token_ids must not be null"); return -1; }
        // 10.: without any block table (memory profiling) the reference pads `sequence_length` slots, not `query` (worker.rs:369)
        q.slots = s.no_block_tables ? q.seq_len : q.seq_len - q.context;
        if (!s.no_block_tables && q.seq_len > q.context) {
            if (!s.block_table) { set_error("Block table should exist for a sequence on decoding phase"); return -1; }
            if ((q.seq_len - 1) / block_size >= s.block_table_len) { set_error("prepare_inputs: block table too short for the sequence"); return -1; }
        }
        L.num_tokens += ntok;
        L.num_slots += q.slots;
        L.max_query_len = std::max(L.max_query_len, q.query);
        L.max_block_table_len = std::max(L.max_block_table_len, q.bt_len);
        if (s.is_prompt) {                                                                 // 9.
            L.num_prefills += 1;
            L.num_prefill_tokens += ntok;
            L.max_prefill_seq_len = std::max(L.max_prefill_seq_len, q.seq_len);
        } else {
            L.num_decode_tokens += 1;
            L.max_decode_seq_len = std::max(L.max_decode_seq_len, q.sliding_seq_len);
            min_decode_seq_len = std::min(min_decode_seq_len, q.sliding_seq_len);
        }
    }
    // ---- layout: every tensor 256-byte aligned inside the one buffer ----
    int64_t off = 0;
    auto place = [&](int64_t bytes) { const int64_t o = off; off = align_up(off + bytes, 256); return o; };
    L.off_input_tokens = place(L.num_tokens * 4);
    L.off_input_positions = place(L.num_tokens * 8);
    L.off_slot_mapping = place(L.num_slots * 8);
    L.off_seq_lens = place(num_sequences * 4);
    L.off_context_lens = place(num_sequences * 4);
    L.off_query_start_loc = place((num_sequences + 1) * 4);
    L.off_seq_start_loc = place((num_sequences + 1) * 4);
    L.off_block_tables = place(num_sequences * L.max_block_table_len * 4);
    L.total_bytes = off;
    *layout = L;
    if (!host_staging) return 0;                       // sizing query
    if (host_capacity < L.total_bytes) { set_error("prepare_inputs: host staging buffer too small"); return -1; }
    if (device_buffer && device_capacity < L.total_bytes) { set_error("prepare_inputs: device buffer too small"); return -1; }

    // ---- pass 2: fill (steps 3, 8, 10, 11) ----
    char *base = static_cast<char *>(host_staging);
    uint32_t *tokens = reinterpret_cast<uint32_t *>(base + L.off_input_tokens);
    int64_t *positions = reinterpret_cast<int64_t *>(base + L.off_input_positions);
    int64_t *slots = reinterpret_cast<int64_t *>(base + L.off_slot_mapping);
    uint32_t *seq_lens = reinterpret_cast<uint32_t *>(base + L.off_seq_lens);
    uint32_t *ctx_lens = reinterpret_cast<uint32_t *>(base + L.off_context_lens);
    uint32_t *q_start = reinterpret_cast<uint32_t *>(base + L.off_query_start_loc);
    uint32_t *s_start = reinterpret_cast<uint32_t *>(base + L.off_seq_start_loc);
    uint32_t *bt = reinterpret_cast<uint32_t *>(base + L.off_block_tables);
    int64_t t = 0, sl = 0;
    q_start[0] = 0;
    s_start[0] = 0;
    for (int64_t i = 0; i < num_sequences; ++i) {
        const atoma_seq_desc &s = seqs[i];
        const Seq &q = info[(size_t)i];
        if (s.is_prompt) {
            for (int64_t j = q.context; j < q.seq_len; ++j) tokens[t + j - q.context] = s.token_ids[j];
        } else {
            tokens[t] = s.token_ids[s.length - 1];      // the last token of the sequence
        }
        const int64_t ntok = s.is_prompt ? q.query : 1;
        // positions context..sequence_length (worker.rs:341): as many as tokens in every case the reference supports
        for (int64_t j = 0; j < ntok; ++j) positions[t + j] = q.context + j;
        t += ntok;
        seq_lens[i] = (uint32_t)q.sliding_seq_len;       // 8.
        ctx_lens[i] = (uint32_t)q.context;
        q_start[i + 1] = q_start[i] + (uint32_t)q.query;  // 11. (integer cumsum; the reference's f32 cumsum is exact below 2^24)
        s_start[i + 1] = s_start[i] + (uint32_t)q.sliding_seq_len;
        uint32_t *row = bt + i * L.max_block_table_len;
        for (int64_t j = 0; j < q.bt_len; ++j) row[j] = s.block_table[q.bt_start + j];
        for (int64_t j = q.bt_len; j < L.max_block_table_len; ++j) row[j] = 0;           // make_tensor_with_pad(.., 0u32, ..)
        if (s.no_block_tables) {
            for (int64_t j = 0; j < q.slots; ++j) slots[sl + j] = PAD_SLOT_ID;
        } else {
            const int64_t start_index = has_sw ? std::max<int64_t>(0, q.query - sliding_window) : 0;   // worker.rs:383-388
            for (int64_t p = q.context; p < q.seq_len; ++p)
                slots[sl + p - q.context] = p < start_index ? PAD_SLOT_ID : (int64_t)s.block_table[p / block_size] * block_size + p % block_size;
        }
        sl += q.slots;
    }
    if (!device_buffer) return 0;
    // what the decode dispatcher cannot see on its own: whether this batch's decode sequences all have ONE length (paged_decode.hip, decode_pair)
    if (L.num_decode_tokens > 0) note_decode_lengths(min_decode_seq_len, L.max_decode_seq_len, L.num_decode_tokens);
    return check_hip(hipMemcpyAsync(device_buffer, host_staging, (size_t)L.total_bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)),
                     "prepare_inputs: hipMemcpyAsync") ? 0 : -1;
}
