// mlp_bwd_h32.hip — instantiations of the fused-MLP backward kernel for hidden = 32 (HT = 2), all depths.
#include "mlp_bwd.h"

namespace LNH_MLP_NS {

int lnh_mlp_backward_h32(uint32_t in_ks, uint32_t nhm, const MlpBwdArgs &a, hipStream_t s) {
    switch (in_ks * 10 + nhm) {
        case 10: return launch_mlp_backward<1, 2, 0>(a, s);
        case 11: return launch_mlp_backward<1, 2, 1>(a, s);
        case 12: return launch_mlp_backward<1, 2, 2>(a, s);
        case 20: return launch_mlp_backward<2, 2, 0>(a, s);
        case 21: return launch_mlp_backward<2, 2, 1>(a, s);
        case 22: return launch_mlp_backward<2, 2, 2>(a, s);
        case 30: return launch_mlp_backward<3, 2, 0>(a, s);
        case 31: return launch_mlp_backward<3, 2, 1>(a, s);
        case 32: return launch_mlp_backward<3, 2, 2>(a, s);
        case 40: return launch_mlp_backward<4, 2, 0>(a, s);
        case 41: return launch_mlp_backward<4, 2, 1>(a, s);
        case 42: return launch_mlp_backward<4, 2, 2>(a, s);
    }
    lnh_set_error("fused MLP backward: input_dim > 128 is not instantiated");
    return LNH_ERR_UNSUPPORTED;
}

}  // namespace LNH_MLP_NS
