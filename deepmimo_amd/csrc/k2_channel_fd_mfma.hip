// Stage 2, frequency domain - split-precision MFMA kernel (variant 2).  Placeholder until the
// kernel lands: reports "unsupported" so that variant 0 (automatic) uses the fp32 vector kernel.
#include "dmx_common.h"

namespace dmx {

bool fd_mfma_supported(const dmx_params&, const WsView&) { return false; }

int launch_channels_fd_mfma(const dmx_params&, const WsView&, int64_t, int64_t, float2*, hipStream_t) {
    set_error("MFMA variant not built");
    return DMX_ERR_SHAPE;
}

}  // namespace dmx
