// mlp_bwd_nhm2.hip — instantiations of the fused-MLP backward kernel with 2 hidden->hidden matrices (hidden = 64).
#include "mlp_bwd.h"

namespace LNH_MLP_NS {

int lnh_mlp_backward_nhm2(uint32_t in_ks, const MlpBwdArgs &a, hipStream_t s) {
    switch (in_ks) {
        case 1: return launch_mlp_backward<1, 4, 2>(a, s);
        case 2: return launch_mlp_backward<2, 4, 2>(a, s);
        case 3: return launch_mlp_backward<3, 4, 2>(a, s);
        case 4: return launch_mlp_backward<4, 4, 2>(a, s);
    }
    lnh_set_error("fused MLP backward: input_dim > 128 is not instantiated");
    return LNH_ERR_UNSUPPORTED;
}

}  // namespace LNH_MLP_NS
