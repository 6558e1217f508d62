// mlp_bwd_nhm0.hip — instantiations of the fused-MLP backward kernel with 0 hidden->hidden matrices (hidden = 64).
#include "mlp_bwd.h"

// 0 = workgroup-cooperative (LDS transposes), 1 = wave-independent (operand-swap transposes).  Tuning switch only.
static int g_variant = 1;
extern "C" __attribute__((visibility("default"))) void lnh_debug_mlp_bwd_variant(int v) { g_variant = v; }

int lnh_mlp_backward_nhm0(uint32_t in_ks, const MlpBwdArgs &a, hipStream_t s) {
    switch (in_ks) {
        case 1: return g_variant ? launch_mlp_backward_wi<1, 4>(a, s) : launch_mlp_backward<1, 4, 0>(a, s);
        case 2: return g_variant ? launch_mlp_backward_wi<2, 4>(a, s) : launch_mlp_backward<2, 4, 0>(a, s);
        case 3: return launch_mlp_backward<3, 4, 0>(a, s);
        case 4: return launch_mlp_backward<4, 4, 0>(a, s);
    }
    lnh_set_error("fused MLP backward: input_dim > 128 is not instantiated");
    return LNH_ERR_UNSUPPORTED;
}

// sigma net of the LiDAR field: level-major feature input, strided gradient rows (mlp_common.h DensityIO)
int lnh_density_mlp_backward_launch(const MlpBwdArgs &a, hipStream_t s) {
    return g_variant ? launch_mlp_backward_wi<1, 4, DensityIO>(a, s) : launch_mlp_backward<1, 4, 0, DensityIO>(a, s);
}
