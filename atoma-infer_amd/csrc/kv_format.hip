// KV block container: a self-describing image of a set of KV-cache pages of every layer, for moving blocks between engines
// or to disk (SURVEY.md 8f item 4 "KV block (wire/on-disk) format").  The reference has no such format -- its swap path
// copies raw pages between two caches of the same process (csrc/src/cache_manager.rs:18-128, worker.rs:602-632) -- so this is
// an addition on the same path: swap-out into ONE contiguous, checksummed buffer instead of 2 x layers separate tensors.
//
//   header (128 bytes, little endian)            atoma_kv_block_header
//   block ids      int64[num_blocks]             the LOGICAL ids the sender gave the blocks (the receiver maps them to its pages)
//   scales         f32[2][num_layers][h_k]       only when dtype = fp8 (k scales of every layer, then v scales)
//   zero padding                                 up to the next multiple of 256 bytes (so that every page of the payload is
//                                                16-byte aligned for the gather / scatter kernel's vector path)
//   payload        [layer][K | V][block][page]   raw page bytes, tensor-major: one gather launch fills it
//
// pack = the swap_blocks gather kernel (GPU -> pinned host over PCIe, or per-page memcpy for pageable memory) + the header
// + a 64-bit checksum of the WHOLE image (the header with its checksum field zeroed, then everything after it); a reader
// re-derives page_bytes / payload_offset / total_bytes from the validated geometry and refuses a header that disagrees, so no
// size or offset it uses comes from unchecked bytes.  Version 2 (version 1 = round 2: unpadded payload, header outside the
// checksum) is refused.
// Pure byte movement: bit-exact, checked against oracle/kv_format_oracle.py.
#include "common.h"
#include <string.h>

extern "C" int atoma_swap_blocks_multi(const void *const *srcs, void *const *dsts, int64_t num_tensors, const int64_t *mapping,
                                       int64_t num_pairs, int64_t block_size_in_bytes, int kind, void *stream);

namespace atoma {

static const char KV_MAGIC[8] = {'A', 'T', 'O', 'M', 'A', 'K', 'V', '1'};

// 64-bit multiply-mix over 8-byte words (+ the tail bytes): cheap on the host, order-sensitive, not cryptographic
static uint64_t kv_mix_words(uint64_t h, const unsigned char *p, size_t bytes) {   // bytes % 8 == 0
    for (size_t i = 0; i < bytes; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 32;
    }
    return h;
}

// checksum of an image: the header (with its checksum field zeroed) followed by the body
static uint64_t kv_checksum(const atoma_kv_block_header &header, const void *body, size_t bytes) {
    atoma_kv_block_header hz = header;
    hz.checksum = 0;
    const unsigned char *p = static_cast<const unsigned char *>(body);
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)(bytes + sizeof hz);
    h = kv_mix_words(h, reinterpret_cast<const unsigned char *>(&hz), sizeof hz);
    const size_t whole = bytes & ~(size_t)7;
    h = kv_mix_words(h, p, whole);
    uint64_t tail = 0;
    for (size_t j = 0; whole + j < bytes; ++j) tail |= (uint64_t)p[whole + j] << (8 * j);
    h = (h ^ tail) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h;
}

static int elt_bytes(int dtype) { return dtype == ATOMA_U8 ? 1 : ((dtype == ATOMA_F16 || dtype == ATOMA_BF16) ? 2 : 0); }

// a * b and a + b on non-negative int64 with overflow reported (header fields are untrusted 32 / 64-bit values)
static bool mul_ok(int64_t a, int64_t b, int64_t *out) { return !__builtin_mul_overflow(a, b, out); }
static bool add_ok(int64_t a, int64_t b, int64_t *out) { return !__builtin_add_overflow(a, b, out); }

struct KvLayout { int64_t page, payload_offset, total; };

// sizes of an image from its geometry; false when the shape is invalid or any product leaves int64
static bool kv_layout(int64_t num_layers, int64_t num_kv_heads, int64_t head_dim, int64_t block_size, int64_t num_blocks, int dtype, KvLayout *out) {
    const int eb = elt_bytes(dtype);
    if (!eb || num_layers <= 0 || num_kv_heads <= 0 || head_dim <= 0 || block_size <= 0 || num_blocks < 0) return false;
    int64_t page, ids, scales = 0, off, tensors, payload, total;
    if (!mul_ok(block_size, num_kv_heads, &page) || !mul_ok(page, head_dim, &page) || !mul_ok(page, eb, &page)) return false;
    if (!mul_ok(num_blocks, 8, &ids)) return false;
    if (dtype == ATOMA_U8 && (!mul_ok(num_layers, num_kv_heads, &scales) || !mul_ok(scales, 8, &scales))) return false;
    if (!add_ok((int64_t)sizeof(atoma_kv_block_header), ids, &off) || !add_ok(off, scales, &off) || !add_ok(off, 255, &off)) return false;
    off &= ~(int64_t)255;
    if (!mul_ok(num_layers, 2, &tensors) || !mul_ok(tensors, num_blocks, &payload) || !mul_ok(payload, page, &payload)) return false;
    if (!add_ok(off, payload, &total)) return false;
    out->page = page; out->payload_offset = off; out->total = total;
    return true;
}

}  // namespace atoma

extern "C" {

int64_t atoma_kv_blocks_packed_size(int64_t num_layers, int64_t num_kv_heads, int64_t head_dim, int64_t block_size, int64_t num_blocks, int dtype) {
    atoma::KvLayout lay;
    return atoma::kv_layout(num_layers, num_kv_heads, head_dim, block_size, num_blocks, dtype, &lay) ? lay.total : -1;
}

// k_caches / v_caches: HOST arrays of num_layers DEVICE pointers [nb, block_size, h_k, d]; block_ids: HOST int64[num_blocks]
// (pages to take); scales (fp8 only): HOST f32 [num_layers][h_k] each.  `out` is host memory (pinned = one gather kernel over
// PCIe).  Synchronises `stream` (the checksum needs the bytes).
int atoma_kv_pack_blocks(const void *const *k_caches, const void *const *v_caches, int64_t num_layers, int64_t num_kv_heads, int64_t head_dim,
                         int64_t block_size, const int64_t *block_ids, int64_t num_blocks, int dtype, const float *k_scales, const float *v_scales,
                         void *out, int64_t out_capacity, void *stream) {
    using namespace atoma;
    clear_error();
    KvLayout lay;
    if (!kv_layout(num_layers, num_kv_heads, head_dim, block_size, num_blocks, dtype, &lay)) {
        set_error("kv_pack_blocks: invalid shape or dtype (f16, bf16 or u8 = fp8 e4m3fn)");
        return -1;
    }
    const int64_t need = lay.total, page = lay.page;
    if (!out || out_capacity < need) { set_error("kv_pack_blocks: output buffer too small (atoma_kv_blocks_packed_size)"); return -1; }
    if (dtype == ATOMA_U8 && (!k_scales || !v_scales)) { set_error("kv_pack_blocks: an fp8 cache needs its scales"); return -1; }
    char *base = static_cast<char *>(out);
    atoma_kv_block_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, KV_MAGIC, 8);
    h.version = 2; h.dtype = (uint32_t)dtype;
    h.num_layers = (uint32_t)num_layers; h.num_kv_heads = (uint32_t)num_kv_heads; h.head_dim = (uint32_t)head_dim; h.block_size = (uint32_t)block_size;
    h.num_blocks = (uint64_t)num_blocks; h.page_bytes = (uint64_t)page; h.total_bytes = (uint64_t)need;
    char *ids = base + sizeof h;
    memcpy(ids, block_ids, 8 * (size_t)num_blocks);
    char *sc = ids + 8 * num_blocks;
    if (dtype == ATOMA_U8) {
        memcpy(sc, k_scales, (size_t)num_layers * num_kv_heads * 4);
        memcpy(sc + num_layers * num_kv_heads * 4, v_scales, (size_t)num_layers * num_kv_heads * 4);
        sc += 2 * num_layers * num_kv_heads * 4;
    }
    memset(sc, 0, (size_t)(lay.payload_offset - (sc - base)));           // the alignment padding is part of the checksummed image
    sc = base + lay.payload_offset;
    h.payload_offset = (uint64_t)lay.payload_offset;
    if (num_blocks > 0) {
        std::vector<const void *> srcs((size_t)(2 * num_layers));
        std::vector<void *> dsts((size_t)(2 * num_layers));
        for (int64_t l = 0; l < num_layers; ++l) {
            srcs[(size_t)(2 * l)] = k_caches[l];
            srcs[(size_t)(2 * l + 1)] = v_caches[l];
            dsts[(size_t)(2 * l)] = sc + (2 * l) * num_blocks * page;
            dsts[(size_t)(2 * l + 1)] = sc + (2 * l + 1) * num_blocks * page;
        }
        std::vector<int64_t> map((size_t)(2 * num_blocks));
        for (int64_t i = 0; i < num_blocks; ++i) { map[(size_t)(2 * i)] = block_ids[i]; map[(size_t)(2 * i + 1)] = i; }
        if (atoma_swap_blocks_multi(srcs.data(), dsts.data(), 2 * num_layers, map.data(), num_blocks, page, ATOMA_SWAP_GPU_TO_CPU, stream) != 0) return -1;
        if (!check_hip(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "kv_pack_blocks sync")) return -1;
    }
    h.checksum = kv_checksum(h, base + sizeof h, (size_t)(need - (int64_t)sizeof h));
    memcpy(base, &h, sizeof h);
    return 0;
}

// Reads the header of a packed image (validating magic, version, sizes and checksum) into *header.
int atoma_kv_read_header(const void *packed, int64_t bytes, atoma_kv_block_header *header) {
    using namespace atoma;
    clear_error();
    if (!packed || !header || bytes < (int64_t)sizeof(atoma_kv_block_header)) { set_error("kv block image: too short for a header"); return -1; }
    atoma_kv_block_header h;
    memcpy(&h, packed, sizeof h);
    if (memcmp(h.magic, KV_MAGIC, 8) != 0) { set_error("kv block image: bad magic"); return -1; }
    if (h.version != 2) { set_error("kv block image: unsupported version"); return -1; }
    // every size / offset a reader uses is re-derived from the geometry; a header that states anything else is refused
    KvLayout lay;
    if (h.num_blocks > (uint64_t)INT64_MAX ||
        !kv_layout(h.num_layers, h.num_kv_heads, h.head_dim, h.block_size, (int64_t)h.num_blocks, (int)h.dtype, &lay) ||
        (uint64_t)lay.total != h.total_bytes || (uint64_t)lay.page != h.page_bytes || (uint64_t)lay.payload_offset != h.payload_offset || bytes < lay.total) {
        set_error("kv block image: sizes in the header do not add up / image truncated");
        return -1;
    }
    const int64_t need = lay.total;
    if (kv_checksum(h, static_cast<const char *>(packed) + sizeof h, (size_t)(need - (int64_t)sizeof h)) != h.checksum) {
        set_error("kv block image: checksum mismatch");
        return -1;
    }
    *header = h;
    return 0;
}

// Scatters block i of the image into page dst_block_ids[i] of every layer's caches (which must have the image's geometry and
// dtype).  Stream-ordered; the image must stay alive until the stream has passed the copy.  fp8: the image's scales are
// returned through k_scales_out / v_scales_out (HOST f32 [num_layers][h_k]) when given.
int atoma_kv_unpack_blocks(const void *packed, int64_t bytes, void *const *k_caches, void *const *v_caches, int64_t num_layers, int64_t num_kv_heads,
                           int64_t head_dim, int64_t block_size, int dtype, const int64_t *dst_block_ids, int64_t num_blocks, float *k_scales_out,
                           float *v_scales_out, void *stream) {
    using namespace atoma;
    atoma_kv_block_header h;
    if (atoma_kv_read_header(packed, bytes, &h) != 0) return -1;
    if ((int64_t)h.num_layers != num_layers || (int64_t)h.num_kv_heads != num_kv_heads || (int64_t)h.head_dim != head_dim || (int64_t)h.block_size != block_size ||
        (int)h.dtype != dtype || (int64_t)h.num_blocks != num_blocks) {
        set_error("kv_unpack_blocks: the image's geometry / dtype / block count differs from the receiver's");
        return -1;
    }
    const char *base = static_cast<const char *>(packed);
    if (dtype == ATOMA_U8) {
        const char *sc = base + sizeof h + 8 * num_blocks;
        if (k_scales_out) memcpy(k_scales_out, sc, (size_t)num_layers * num_kv_heads * 4);
        if (v_scales_out) memcpy(v_scales_out, sc + num_layers * num_kv_heads * 4, (size_t)num_layers * num_kv_heads * 4);
    }
    if (num_blocks == 0) return 0;
    const int64_t page = (int64_t)h.page_bytes;            // == the value derived from the geometry (atoma_kv_read_header)
    const char *payload = base + h.payload_offset;
    std::vector<const void *> srcs((size_t)(2 * num_layers));
    std::vector<void *> dsts((size_t)(2 * num_layers));
    for (int64_t l = 0; l < num_layers; ++l) {
        srcs[(size_t)(2 * l)] = payload + (2 * l) * num_blocks * page;
        srcs[(size_t)(2 * l + 1)] = payload + (2 * l + 1) * num_blocks * page;
        dsts[(size_t)(2 * l)] = k_caches[l];
        dsts[(size_t)(2 * l + 1)] = v_caches[l];
    }
    std::vector<int64_t> map((size_t)(2 * num_blocks));
    for (int64_t i = 0; i < num_blocks; ++i) { map[(size_t)(2 * i)] = i; map[(size_t)(2 * i + 1)] = dst_block_ids[i]; }
    return atoma_swap_blocks_multi(srcs.data(), dsts.data(), 2 * num_layers, map.data(), num_blocks, page, ATOMA_SWAP_CPU_TO_GPU, stream);
}

}  // extern "C"
