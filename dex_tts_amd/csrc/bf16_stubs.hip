// Placeholder until the bf16 MFMA kernels land: the fp32 path never reaches these (dex_ctx_set_precision
// rejects DEX_PREC_BF16), and reaching them by mistake must fail loudly, not fall back silently.
#include <cstdio>
#include <cstdlib>
#include "kernels.h"
namespace dex {
void launch_attention_bf16(const AttnP&, hipStream_t) { fprintf(stderr, "dexamd: bf16 attention not built\n"); abort(); }
}  // namespace dex
