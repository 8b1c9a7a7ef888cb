// placeholder until the MFMA prefill kernel lands (next commit): nothing is routed here yet.
#include "attn_params.h"
namespace atoma {
bool prefill_mfma_supported(const AttnParams &) { return false; }
void launch_prefill_mfma(const AttnParams &, bool, hipStream_t) {}
}  // namespace atoma
