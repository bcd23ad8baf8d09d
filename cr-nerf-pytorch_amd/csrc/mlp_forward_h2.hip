// Stand-alone NeRF_sigma forward on the h2 core: mlp_forward_x3.hip's kernel built with two fp16 pieces per operand (mlp_core_x3.h, CRNERF_X_NP = 2:
// three MFMAs per product instead of six, packs from crnerf_pack_mlp_weights_h2).  Entry: crnerf_mlp_forward_f32h2.
#define CRNERF_X_NP 2
#define mlp_forward_x3_kernel mlp_forward_h2_kernel
#define launch_mlp_forward_x3 launch_mlp_forward_h2
#define gather_embedded_x3 gather_embedded_h2
#include "mlp_forward_x3.hip"
