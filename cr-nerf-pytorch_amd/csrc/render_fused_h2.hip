// Fused volumetric renderer on the h2 core: render_fused_x3.hip's kernels built with two fp16 pieces per operand (CRNERF_X_NP = 2).
// Inference (crnerf_render_rays_f32h2) on the lazy-epilogue core mlp_core_h2.h; the training twin (crnerf_render_rays_train_f32h2) on the
// core's training form mlp_core_h2t.h -- same saved state as the fp32 / f32x3 twins.
#define CRNERF_X_NP 2
#define render_rays_x3_kernel render_rays_h2_kernel
#define render_rays_x3_rng_kernel render_rays_h2_rng_kernel
#define render_rays_train_x3_kernel render_rays_train_h2_kernel
#define render_rays_train_x3_rng_kernel render_rays_train_h2_rng_kernel
#define render_rays_x3_body render_rays_h2_body
#define launch_render_rays_x3 launch_render_rays_h2
#define RenderParamsX RenderParamsH
#define NoHookX NoHookH
#define TrainHookX TrainHookH
#include "render_fused_x3.hip"
