// Launcher of the decode attention on the matrix cores (attn_mfma.h).
#include "attn_mfma.h"

namespace tl {

// head_dim 128, a whole GQA group per workgroup, and a wave's 32 tokens of a stage inside one page
bool attn_decode_mfma_applicable(const AttnDecodeArgs &a, int head_dim, int rq) {
    return head_dim == 128 && rq == AD_RQ && a.page_shift >= 5 && a.tokens_per_split % 32 == 0;
}

void launch_attn_decode_mfma(const AttnDecodeArgs &a, dim3 grid, hipStream_t st) {
    const bool kv8 = a.key_scales != nullptr;  // FP8 pages (kv8.h)
    if (a.qkv_partial != nullptr) {
        if (kv8) hipLaunchKernelGGL((attn_decode_mfma_kernel<true, true>), grid, dim3(256), attn_mfma_lds_bytes(true), st, a);
        else hipLaunchKernelGGL((attn_decode_mfma_kernel<true, false>), grid, dim3(256), attn_mfma_lds_bytes(true), st, a);
    } else {
        if (kv8) hipLaunchKernelGGL((attn_decode_mfma_kernel<false, true>), grid, dim3(256), attn_mfma_lds_bytes(false), st, a);
        else hipLaunchKernelGGL((attn_decode_mfma_kernel<false, false>), grid, dim3(256), attn_mfma_lds_bytes(false), st, a);
    }
}

}  // namespace tl
