// bf16-MFMA ("fast") precision path -- placeholder until the kernels land.
#include "common.hpp"
#include "kernels.hpp"

namespace genie {
struct Workspace;
int st_block_bf16(const genie_cfg&, const genie_layer_weights&, float*, Workspace&, int, hipStream_t) {
    set_error("GENIE_PREC_BF16 is not built yet");
    return GENIE_E_UNSUPPORTED;
}
int readout_bf16(const genie_cfg&, const genie_weights&, const float*, int, int, int, int, float*, hipStream_t) {
    set_error("GENIE_PREC_BF16 is not built yet");
    return GENIE_E_UNSUPPORTED;
}
}  // namespace genie
