// mlp_bwd_nhm0.hip — instantiations of the fused-MLP backward kernel with 0 hidden->hidden matrices (hidden = 64).
#include "mlp_bwd.h"

namespace LNH_MLP_NS {

// One-hidden-layer nets up to 64 inputs take the wave-independent kernel (measured 166 us vs 196 us for the
// workgroup-cooperative one on the sigma net); wider inputs would not fit its register budget.
int lnh_mlp_backward_nhm0(uint32_t in_ks, const MlpBwdArgs &a, hipStream_t s) {
    switch (in_ks) {
        case 1: return launch_mlp_backward_wi<1, 4>(a, s);
        case 2: return launch_mlp_backward_wi<2, 4>(a, s);
        case 3: return launch_mlp_backward<3, 4, 0>(a, s);
        case 4: return launch_mlp_backward<4, 4, 0>(a, s);
    }
    lnh_set_error("fused MLP backward: input_dim > 128 is not instantiated");
    return LNH_ERR_UNSUPPORTED;
}

// sigma net of the LiDAR field: level-major feature input, strided gradient rows (mlp_common.h DensityIO)
int lnh_density_mlp_backward_launch(const MlpBwdArgs &a, hipStream_t s) {
    return launch_mlp_backward_wi<1, 4, DensityIO>(a, s);
}

}  // namespace LNH_MLP_NS
