// Fused volumetric renderer on the h2 core: render_fused_x3.hip's kernels built with two fp16 pieces per operand (mlp_core_x3.h, CRNERF_X_NP = 2).
// Inference entry only (crnerf_render_rays_f32h2); the training twins stay on the x3 / fp32 cores.
#define CRNERF_X_NP 2
#define render_rays_x3_kernel render_rays_h2_kernel
#define render_rays_x3_rng_kernel render_rays_h2_rng_kernel
#define render_rays_x3_body render_rays_h2_body
#define launch_render_rays_x3 launch_render_rays_h2
#define RenderParamsX RenderParamsH
#define NoHookX NoHookH
#include "render_fused_x3.hip"
