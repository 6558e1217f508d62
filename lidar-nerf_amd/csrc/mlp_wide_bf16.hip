// mlp_wide_bf16.hip — mlp_wide.hip with bf16 MFMA operands (lnh_mlp_backward_data_bf16; lnh_mlp_forward_bf16 calls its forward).
#define LNH_MLP_BF16 1
#include "mlp_wide.hip"
