// Error slot, device queries and the reference's host-side split heuristic.
#include "common.h"

#include <math.h>
#include <mutex>

namespace atoma {

static thread_local std::string g_error;

void set_error(const std::string &msg) { g_error = msg; }
void clear_error() { g_error.clear(); }
bool has_error() { return !g_error.empty(); }

bool set_decode_option(const std::string &name, int value);  // paged_decode.hip

int device_num_cus() {
    // The reference queries cudaDeviceGetAttribute on EVERY attention call
    // (/root/reference/csrc/src/lib.rs:1545,2201-2233); cache it per device instead.
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// /root/reference/csrc/src/lib.rs:2122-2167, f32 arithmetic as there.
int num_splits_heuristic(int64_t batch_nheads_mblocks, int64_t num_sms, int64_t num_n_blocks, int64_t max_splits) {
    if ((float)batch_nheads_mblocks >= 0.8f * (float)num_sms) return 1;
    max_splits = std::min(max_splits, std::min(num_sms, num_n_blocks));
    float max_eff = 0.f;
    std::vector<float> eff;
    eff.reserve((size_t)(max_splits > 0 ? max_splits : 0));
    auto eligible = [&](int64_t s) { return s == 1 || cdiv(num_n_blocks, s) != cdiv(num_n_blocks, s - 1); };
    for (int64_t s = 1; s <= max_splits; ++s) {
        if (!eligible(s)) { eff.push_back(0.f); continue; }
        const float n_waves = (float)(batch_nheads_mblocks * s) / (float)num_sms;
        const float e = n_waves / ceilf(n_waves);
        if (e > max_eff) max_eff = e;
        eff.push_back(e);
    }
    for (int64_t s = 1; s <= max_splits; ++s)
        if (eligible(s) && eff[(size_t)s - 1] >= 0.85f * max_eff) return (int)s;
    return 1;
}

}  // namespace atoma

extern "C" {

const char *atoma_last_error(void) { return atoma::g_error.c_str(); }
void atoma_clear_error(void) { atoma::clear_error(); }

int atoma_num_splits_heuristic(int64_t batch_nheads_mblocks, int64_t num_sms, int64_t num_n_blocks,
                               int64_t max_splits) {
    return atoma::num_splits_heuristic(batch_nheads_mblocks, num_sms, num_n_blocks, max_splits);
}

// /root/reference/csrc/src/lib.rs:2169-2199 with CUs in place of SMs.
int atoma_compute_num_splits(int64_t batch_size, int64_t num_heads, int64_t head_size, int64_t max_seqlen_k,
                             int64_t max_seqlen_q, int num_cus) {
    const int64_t block_n = head_size <= 64 ? 256 : (head_size <= 128 ? 128 : 64);
    const int64_t n_blocks = atoma::cdiv(max_seqlen_k, block_n);
    const int64_t m_blocks = atoma::cdiv(max_seqlen_q, 64);
    if (num_cus <= 0) num_cus = atoma::device_num_cus();
    return atoma::num_splits_heuristic(batch_size * num_heads * m_blocks, (int64_t)num_cus * 2, n_blocks, 128);
}

int atoma_set_option(const char *name, int value) {
    atoma::clear_error();
    if (name && atoma::set_decode_option(name, value)) return 0;
    atoma::set_error(std::string("atoma_set_option: unknown option ") + (name ? name : "(null)"));
    return -1;
}

int atoma_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int atoma_num_cus(int device) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
