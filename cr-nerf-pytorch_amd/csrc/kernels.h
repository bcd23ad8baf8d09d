// Internal launcher declarations shared by the .hip translation units and the C-ABI (abi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crnerf {

// thread-local last-error slot behind crnerf_last_error(); returns code
int set_error(int code, const char* msg);
int check_launch(const char* what);
// compute units of the current device (hipDeviceAttributeMultiprocessorCount, cached per device): the persistent kernels launch
// one workgroup per CU
int num_cus();
// Dynamic work distribution of the persistent renderers (launches with more ray quads than workgroups): a slot of two device
// counters {next quad, finished workgroups}, zero between launches (the last workgroup to finish resets its slot).  Slots rotate
// so that launches in flight on different streams do not share one.  Why: the eight XCDs do not run at the same clock under
// MFMA load (tools/wg_times.py: workgroup lifetimes differ by up to 10 % between XCDs), so a static quad -> workgroup map makes
// every launch as slow as its slowest XCD.
constexpr int SCHED_SLOTS = 64;
unsigned int* sched_slot(const void* symbol);   // symbol: the translation unit's `__device__ unsigned int [SCHED_SLOTS][2]` array
// raises `fn`'s dynamic-LDS limit to `bytes` once per (function, device) instead of on every launch; returns 0 or an error code
int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what);

struct MlpTensors {  // device pointers to the 24 tensors of one NeRF_sigma (models/nerf.py:137-154)
  const float* w[8];   // xyz_encoding_{1..8}.0.weight
  const float* b[8];   // xyz_encoding_{1..8}.0.bias
  const float* w_final; const float* b_final;  // xyz_encoding_final
  const float* w_sigma; const float* b_sigma;  // static_sigma.0
  const float* w_dir;   const float* b_dir;    // dir_encoding.0
  const float* w_rgb;   const float* b_rgb;    // static_rgb.0
};

int launch_pack_mlp(const MlpTensors& t, void* packed, hipStream_t stream);
int launch_pack_mlpT(const MlpTensors& t, void* packed, hipStream_t stream);
size_t mlp_train_acts_bytes(long P);
size_t mlp_train_scratch_bytes(long P);
// The saved state's RANGE WORD (round 6): behind the rows and relu bits of a pass's P points sits one 256-byte line whose first word
// holds the bits of the largest |operand| -- activations and embedded inputs -- the h2 forward twin multiplied in this pass (every workgroup raises
// it with an atomic max; Inf = a point left fp16's range).  The f16x2 weight gradients take the power of two of their ACTIVATION operand from it
// (mlp_train16.hip act_scale_h); 0 = not tracked (the fp32 / f32x3 twins wrote the rows): scale 1, as up to round 5.  Every producer of saved rows
// zeroes it first (zero_acts_range).
constexpr size_t ACTS_RANGE_BYTES = 256;
__host__ __device__ inline size_t acts_rows_bytes(long P) { return (size_t)10 * P * 256 * 4 + (size_t)10 * P * 32; }
__host__ __device__ inline uint32_t* acts_range_word(const void* acts, long P) { return (uint32_t*)((char*)acts + acts_rows_bytes(P)); }
inline int zero_acts_range(void* acts, long P, hipStream_t st) {
  return hipMemsetAsync(acts_range_word(acts, P), 0, ACTS_RANGE_BYTES, st) == hipSuccess ? 0 : set_error(-1, "saved state: hipMemsetAsync(range word) failed");
}
int launch_mlp_forward_train(const void* packed, const float* x, float* out, float* acts, long P, hipStream_t stream);
// flags: bit 0 = weight gradients of every Linear except static_sigma from bf16-rounded operands (CRNERF_BWD_WGRAD_BF16, include/crnerf.h)
int launch_mlp_backward(const void* packedT, const float* x, const float* out, const float* d_out, const float* acts, void* scratch,
                        float* const* grads, long P, hipStream_t stream, int flags = 0, const void* packedT_x3 = nullptr, const void* packedT_h2 = nullptr);
int launch_mlp_wgrads(const float* x, const float* acts, const float* deltas, const float* d_rgb, const float* d_sig, float* ws, float* const* grads,
                      long P, hipStream_t stream, int wb, const uint32_t* dmax = nullptr, const uint32_t* amax = nullptr);
// mixed-precision training twins (mlp_gemm_bf16.hip): per-layer bf16-MFMA GEMMs; activations, deltas and the embedded input travel as bf16
size_t gemm_packed_bytes();
size_t mlp_train_mixed_acts_bytes(long P);
size_t mlp_train_mixed_scratch_bytes(long P);
int launch_pack_mlp_gemm(const MlpTensors& t, void* packed, hipStream_t st);
int launch_mlp_forward_train_mixed(const MlpTensors& t, const void* packed, const float* x, float* out, void* acts, long P, hipStream_t st);
int launch_mlp_backward_mixed(const MlpTensors& t, const void* packed, const float* x, const float* out, const float* d_out, const void* acts,
                              void* scratch, float* const* grads, long P, hipStream_t st, int acts_layout = 0);
// dst[m * ldc + n] = sum_c partial[c][m][n] (+ db[m] = sum_c bias_partial[c][m]): the deterministic partial-sum reduction of the wgrad kernels
int launch_wgrad_reduce(const float* partial, int nchunk, int M, int N, float* dst, int ldc, const float* bias_partial, float* db, hipStream_t st);
int launch_ray_directions(float fx, float fy, float cx, float cy, int H, int W, float* dirs, hipStream_t stream);
int launch_rays_from_directions(const float* dirs, const float* c2w_host, long n, float* rays_o, float* rays_d, hipStream_t stream);
int launch_generate_rays(const float* intr4_host, const float* c2w_host, int H, int W, float near, float far, float* rays, hipStream_t stream);
size_t encoder_workspace_bytes(int H, int W);
int launch_encoder_forward(const float* img, int H, int W, const float* const* w, void* workspace, float* out, hipStream_t st);
// dst[m*ldc + n] = sum_p D[p][m] * A[p][n] (and db[m] = sum_p D[p][m] when db != null); ws: wgrad_workspace_floats()
size_t wgrad_workspace_floats(long P, int M, int N, int bf16 = 0);   // bf16: the mode wgrad() will be called with (3 = f16x2: narrow jobs run on half chunks)
// bf16 != 0: full 256 x 256 tiles multiply bf16-rounded operands on the bf16 MFMA (fp32 accumulate); other shapes stay fp32
int wgrad(const float* D, int ldd, int M, const float* A, int lda, int N, float* dst, int ldc, float* db, long P, float* ws,
          hipStream_t st, int bf16 = 0, const uint32_t* dmax = nullptr, const uint32_t* amax = nullptr);
// several such products in ONE launch + ONE reduction (small batches: a launch per job is mostly ramp-up and drain); at most 16 jobs
// bf16: 0 fp32 operands, 1 bf16-rounded, 2 "bf16x3", 3 "f16x2" (full tiles; dmax = the bits of max |D| over the tensor, see wgrad_h2_kernel)
struct WgradSpec { const float* D; int ldd, M; const float* A; int lda, N; float* dst; int ldc; float* db; float weight; int bf16; long P; const uint32_t* dmax = nullptr;
                   const uint32_t* amax = nullptr; };   // amax: the saved state's range word (acts_range_word) or null -- the f16x2 activation scale
float wgrad_job_weight(int M, int N);                              // per-point cost of a job relative to a full 256 x 256 block
size_t wgrad_batch_ws_floats(const WgradSpec* specs, int n);       // partial-sum workspace the plan for these jobs needs
int wgrad_batch(const WgradSpec* specs, int n, float* ws, size_t ws_floats, hipStream_t st);
int launch_posenc(const float* x, float* out, long n, int n_freqs, hipStream_t stream);
int launch_embed_points(const float* rays, const float* z, const float* dir_emb, float* x, long R, int N, hipStream_t stream);
int launch_composite(const float* raw, const float* z, const float* noise, float noise_std, float* weights,
                     float* feature, float* depth, long R, int N, hipStream_t stream);
int launch_composite_backward(const float* raw, const float* z, const float* noise, float noise_std, const float* d_feature,
                              const float* d_depth, const float* d_weights, float* d_raw, long R, int N, hipStream_t stream);
int launch_sample_pdf_merge(const float* z_coarse, const float* weights_coarse, const float* u, long u_stride, float* z_fine_sorted,
                            float* z_samples, long R, int Nc, int Ni, hipStream_t stream);

struct RenderArgs {
  const void* packed_coarse;
  const void* packed_fine;     // may be null when n_importance == 0
  const float* rays;           // [R,8]
  const float* view_dir;       // [R,3] or null -> rays[:,3:6]
  const float* z_coarse;       // [R,Nc] or null -> computed from near/far (perturb == 0)
  const float* z_steps;        // [Nc] linspace(0,1,Nc) table or null -> in-kernel formula
  const float* u;              // [R,Ni] (u_stride = Ni) / shared [Ni] table (u_stride = 0) or null -> in-kernel linspace
  long u_stride;
  const float* noise_coarse;   // [R,Nc] or null
  const float* noise_fine;     // [R,Nc+Ni] or null
  float noise_std;
  int use_disp;
  long R;
  int Nc, Ni;
  float* weights_coarse;       // [R,Nc]
  float* feature_coarse;       // [R,64]
  float* depth_coarse;         // [R]
  float* weights_fine;         // [R,Nc+Ni]
  float* feature_fine;         // [R,64]
  float* depth_fine;           // [R]
  float* z_fine;               // [R,Nc+Ni] optional (null to skip)
  // in-kernel random draws (include/crnerf.h, ABI 2; philox.h)
  unsigned long long rng_seed = 0;
  long rng_ray_offset = 0;
  int rng_flags = 0;
  float perturb = 0.0f;
  float* z_coarse_out = nullptr;
  float* noise_coarse_out = nullptr;
  float* noise_fine_out = nullptr;
  // crnerf_render_rays_f32x3_repair: only ray quads whose feature_coarse / feature_fine hold a NaN are rendered (again)
  int repair = 0;
  // crnerf_render_rays_bf16_fine: weights_coarse is an INPUT (rendered by another core); sample_pdf + merge + the fine pass only
  int fine_only = 0;
  // training twin (crnerf_render_rays_train_f32; all null for inference): saved activations + raw MLP outputs per pass
  void* train_acts_coarse = nullptr;   // crnerf_mlp_train_acts_bytes(R*Nc)
  void* train_acts_fine = nullptr;     // crnerf_mlp_train_acts_bytes(R*(Nc+Ni))
  float* train_raw_coarse = nullptr;   // [R*Nc,65]
  float* train_raw_fine = nullptr;     // [R*(Nc+Ni),65]
};
int launch_render_rays16(const RenderArgs& a, hipStream_t stream);
int launch_mlp_forward16(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream);
int launch_pack_mlp_bf16(const MlpTensors& t, void* packed, hipStream_t stream);
int launch_pack_mlp_x3(const MlpTensors& t, void* packed, hipStream_t stream);
int launch_mlp_forward_x3(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream, int repair);
int launch_render_rays_x3(const RenderArgs& a, hipStream_t stream);
int launch_pack_mlp_x3t(const MlpTensors& t, void* packed, hipStream_t stream);
// "h2" core (mlp_core_x3.h built with two fp16 pieces per operand): packs, module forward, fused renderer (inference)
int launch_pack_mlp_h2(const MlpTensors& t, void* packed, hipStream_t stream, bool check = true);
int pack_h2_status(const void* packed, hipStream_t stream);
int launch_mlp_forward_h2(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream, int repair);
int launch_render_rays_h2(const RenderArgs& a, hipStream_t stream);
int launch_pack_mlp_h2t(const MlpTensors& t, void* packed, hipStream_t stream);
// dmax: null, or ACT_SLOTS + 1 words the kernel raises (atomic max) to the bits of the largest |delta| it stored in every slot and in d_rgb (what
// the f16x2 weight gradients range their delta operands with, mlp_train16.hip wgrad_h2_kernel); the caller zeroes them
int launch_mlp_dgrad_h2(const void* packedT_h2, const float* out, const float* d_out, const float* acts, float* deltas, float* d_rgb, float* d_sig, long P,
                        hipStream_t stream, uint32_t* dmax = nullptr);
// only_if: device word; the kernel leaves at once when it is 0 (the f32x3 stand-in of an h2 data gradient whose pack was refused); null: always
int launch_mlp_dgrad_x3(const void* packedT_x3, const float* out, const float* d_out, const float* acts, float* deltas, float* d_rgb, float* d_sig, long P,
                        hipStream_t stream, const int* only_if = nullptr);
int launch_mlp_forward_bf16p(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream);   // pair core: the module entry
int launch_render_rays_bf16p(const RenderArgs& a, hipStream_t stream);
int launch_rng_fill(float* out, long R, int n, unsigned long long seed, int stream_id, long ray_offset, hipStream_t stream);   // one ray per wave PAIR, 32-point tiles, two waves per SIMD (render_fused_bf16p.hip)


size_t content_backward_workspace_floats(long HW);
int launch_content_backward(const float* content, long HW, const float* W, const float* rgb, long rgb_stride, const float* d_rgb, long d_stride,
                            float* workspace, float* d_content, float* dW, float* db, hipStream_t stream);
size_t encoder_train_saved_bytes(int H, int W, int n_out = 1024);
size_t encoder_train_scratch_bytes(int H, int W, int n_out = 1024);
int launch_encoder_forward_train(const float* img, int H, int W, const float* const* w, void* saved, float* out, hipStream_t st);
// the same over a band of rows of an image of Hg rows (encoder_train.hip EncBand): rows [row0, row0 + H) in, output rows [o0, o1) of the 32 x 32 map out
int launch_encoder_forward_train_band(const float* img, int H, int W, int Hg, int row0, int o0, int o1, const float* const* w, void* saved, float* out, hipStream_t st);
int launch_encoder_backward_band(int H, int W, int Hg, int row0, int o0, int o1, const float* const* w, const void* saved, const float* out, const float* d_out,
                                 void* scratch, float* const* grads, float* d_img, hipStream_t st);
int launch_encoder_backward(int H, int W, const float* const* w, const void* saved, const float* out, const float* d_out, void* scratch,
                            float* const* grads, float* d_img, hipStream_t st);

// ---- transient-mask network operators (cgnet.hip)
struct ConvGeom { int cin, cout, H, W, Ho, Wo, k, stride, pad, dil, depthwise; int accum = 0; };   // accum: the data gradient is ADDED to d_x (chain only)
int launch_cg_conv_forward(const ConvGeom& g, const float* x, const float* w, float* y, hipStream_t st);
int launch_cg_conv_backward(const ConvGeom& g, const float* x, const float* w, const float* dy, float* dx, float* dw, hipStream_t st);
int launch_cg_bn_prelu_forward(const float* x, const float* gamma, const float* beta, const float* alpha, float* mean, float* invstd, float* var_unbiased,
                               float* y, int C, int HW, float eps, int training, hipStream_t st, float* run_mean = nullptr, float* run_var = nullptr,
                               long long* n_tracked = nullptr, float momentum = 0.0f);
int launch_cg_bn_prelu_backward(const float* x, const float* gamma, const float* beta, const float* alpha, const float* mean, const float* invstd,
                                const float* dy, float* dx, float* dgamma, float* dbeta, float* dalpha, int C, int HW, int training, hipStream_t st);
int launch_cg_avgpool(const float* in, float* out, int C, int H, int W, int backward, hipStream_t st);
int launch_cg_fglo_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* stats, float* y, int C, int R, int HW,
                           hipStream_t st, const float* residual = nullptr);
int launch_cg_add_inplace(float* dst, const float* src, int n, hipStream_t st);
// F_loc / F_sur (depth-wise 3x3, dilation 1 / dil) -> cat -> BatchNorm + PReLU as one launch each way (the training chain, cgnet_chain.hip)
int launch_cg_dwpair_bn_prelu_forward(const float* y, const float* w_loc, const float* w_sur, int n, int H, int W, int dil, float* cat, const float* gamma,
                                      const float* beta, const float* alpha, float* mean, float* invstd, float* var_unbiased, float* z, float eps,
                                      float* run_mean, float* run_var, long long* n_tracked, float momentum, hipStream_t st);
int launch_cg_dwpair_bn_prelu_backward(const float* y, const float* w_loc, const float* w_sur, int n, int H, int W, int dil, const float* cat, const float* gamma,
                                       const float* beta, const float* alpha, const float* mean, const float* invstd, const float* dz, float* d_cat,
                                       float* dgamma, float* dbeta, float* dalpha, float* dw_loc, float* dw_sur, float* d_y, hipStream_t st);

int launch_cg_fglo_backward(const float* x, const float* w1, const float* w2, const float* stats, const float* dy, float* scratch, float* dx, float* dw1,
                            float* db1, float* dw2, float* db2, int C, int R, int HW, hipStream_t st);
int launch_cg_bilinear(const float* in, const long* idx, float* out, long n, int h, int w, int Ho, int Wo, int sigmoid, hipStream_t st);
int launch_cg_bilinear_backward(const float* out, const float* d_out, const long* idx, float* d_in, long n, int h, int w, int Ho, int Wo, int sigmoid,
                                hipStream_t st);

// the whole network of a training step as two calls (cgnet_chain.hip); parameter / BatchNorm order documented there
constexpr int CGNET_PARAMS = 76, CGNET_BNS = 14;
struct CgNetArgs {
  int cin, H, W;
  const float* const* params;                       // [CGNET_PARAMS]
  float* const* run_mean; float* const* run_var;    // [CGNET_BNS] running buffers, updated by the forward
  long long* const* tracked;                        // [CGNET_BNS] num_batches_tracked (array or entries may be null)
  float momentum, eps;
};
size_t cgnet_arena_floats(int cin, int H, int W);   // floats of `saved` (forward) and of `scratch` (backward)
int launch_cgnet_forward_train(const CgNetArgs& a, const float* image, float* saved, float* mask, hipStream_t st);
int launch_cgnet_backward(const CgNetArgs& a, const float* image, const float* saved, const float* mask, const float* d_mask, float* scratch,
                          float* const* grads, hipStream_t st);

// ---- training-side neighbours (train_aux.hip)
struct LossArgs {
  const float* rgb_c; const float* rgb_f; const float* tgt; const float* mask;   // [R,3] strided, [R,3] or null, [R,3], [R] or null
  const float* a; const float* a_rand; const float* a_rec; const float* c_wo; const float* c_with;
  long R, n_a, n_rec, n_c;
  long rc_sr, rc_sc, rf_sr, rf_sc, tg_sr, tg_sc;   // element strides (row, channel)
  int mse_a;
};
struct LossScales { float s[7]; };                 // loss_k = s[k] * sum_k
struct LossGrads { float *d_rgb_c, *d_rgb_f, *d_mask, *d_a, *d_a_rec, *d_c_wo, *d_c_with; };   // contiguous, any may be null
size_t loss_workspace_bytes();
int launch_loss_forward(const LossArgs& a, const LossScales& sc, float* losses, void* workspace, hipStream_t stream);
int launch_loss_backward(const LossArgs& a, const LossScales& sc, const float* upstream, const LossGrads& g, hipStream_t stream);
struct BatchArgs {
  const float* all_rays; long ray_stride; const float* all_rgbs; long row_offset;
  int img_w, img_h, side;
  const float* w_lin; const float* h_lin;          // [side] linspace tables (host-built like the reference's)
  float scale, h_offset, w_offset;
  float* rays; long* ts; float* rgbs; long* rgb_idx; float* uv;
};
int launch_grid_batch(const BatchArgs& a, hipStream_t stream);
constexpr int ADAM_MAX_TENSORS = 448;             // gradient pointers ride in the kernel arguments (3.5 KB of the 4 KB)
constexpr int ADAM_BLOCK_ELEMS = 4096;            // elements one workgroup updates
struct AdamHyper { float step_size, beta1, beta2, eps, weight_decay, bias_correction2_sqrt; };
int launch_adam_step(float* p, float* m, float* v, const int* blocks, int n_blocks, const float* const* grads, int n_tensors,
                     const AdamHyper& h, hipStream_t stream);

}  // namespace crnerf
