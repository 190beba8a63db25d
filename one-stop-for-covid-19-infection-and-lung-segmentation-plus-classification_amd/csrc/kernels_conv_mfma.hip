// placeholder: MFMA kernels land here
#include "common.h"
bool mfma_conv3x3_supported(int, int) { return false; }
int32_t k_conv3x3_mfma_fwd(unet_ctx* ctx, const float*, const float*, const float*, const float*, float*, int, int, int, int, int, int, hipStream_t) { UNET_FAIL(ctx, UNET_E_SHAPE, "mfma conv not built"); }
size_t mfma_wgrad_ws_bytes(int, int, int, int, int) { return 0; }
int32_t k_conv3x3_mfma_wgrad(unet_ctx* ctx, const float*, const float*, float*, float*, void*, size_t, int, int, int, int, int, hipStream_t) { UNET_FAIL(ctx, UNET_E_SHAPE, "mfma wgrad not built"); }
