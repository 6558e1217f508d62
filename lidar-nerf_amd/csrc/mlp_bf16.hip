// mlp_bf16.hip — mlp.hip with bf16 MFMA operands: entry points lnh_mlp_forward_bf16 / lnh_mlp_backward_bf16 /
// lnh_density_mlp_forward_bf16 / lnh_density_mlp_backward_bf16 (mlp_common.h, "Element type").
#define LNH_MLP_BF16 1
#include "mlp.hip"
