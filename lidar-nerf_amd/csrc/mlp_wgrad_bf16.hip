// mlp_wgrad_bf16.hip — the same translation unit with bf16 MFMA operands (mlp_common.h, "Element type").
#define LNH_MLP_BF16 1
#include "mlp_wgrad.hip"
