// lidar_color_bf16.hip — lidar_color.hip with bf16 MFMA operands (mlp_common.h, "Element type").
#define LNH_MLP_BF16 1
#include "lidar_color.hip"
