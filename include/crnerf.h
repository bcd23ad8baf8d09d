/*
 * crnerf.h -- C ABI of the MI355X-native CR-NeRF rendering hot path (libcrnerf_hip.so).
 *
 * The reference (YifYang993/CR-NeRF-PyTorch) has no FFI layer: its boundary for this path is the
 * Python call render_rays_cross_ray(...) plus the nn.Module protocol of NeRF_sigma / PosEmbedding /
 * style_net.  Each entry point below replaces the eager-PyTorch op cluster cited next to it; the
 * Python shim in cr-nerf-pytorch_amd/models/ presents the reference signatures on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a caller-owned, contiguous fp32 buffer unless stated;
 *   - nothing is allocated inside; scratch comes from caller-provided workspaces whose size is
 *     returned by the *_bytes() queries;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue work;
 *   - return value: 0 on success, negative error code otherwise; crnerf_last_error() returns the
 *     thread-local message of the last failing call; nothing throws;
 *   - the network shape is the one the reference ships and hard-wires (models/nerf.py:117,
 *     opt.py:46-48,93): D=8, W=256, skip at layer 5, N_emb_xyz=15, N_emb_dir=4, nerf_out_dim=64.
 */
#ifndef CRNERF_H
#define CRNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRNERF_ABI_VERSION 2    /* 2: crnerf_render_args grew the in-kernel random-draw fields (appended), crnerf_rng_fill_f32 */

#define CRNERF_OK 0
#define CRNERF_ERR_NULL (-1)     /* required pointer is NULL */
#define CRNERF_ERR_SHAPE (-2)    /* unsupported size */
#define CRNERF_ERR_CONFIG (-3)   /* inconsistent arguments */
#define CRNERF_ERR_RANGE (-4)    /* crnerf_pack_mlp_weights_h2: a weight does not fit the h2 core's range (use the f32x3 / f32 entry points) */
#define CRNERF_ERR_HIP (-10)     /* HIP runtime / launch failure */

int crnerf_abi_version(void);
const char* crnerf_last_error(void);

/* Tensor order of one NeRF_sigma state_dict (models/nerf.py:137-154):
 *   [2i], [2i+1]  xyz_encoding_{i+1}.0.weight / .bias   i = 0..7   ([256,93] [256,256]x3 [256,349] [256,256]x3)
 *   [16],[17]     xyz_encoding_final.weight / .bias     [256,256]
 *   [18],[19]     static_sigma.0.weight / .bias         [1,256]
 *   [20],[21]     dir_encoding.0.weight / .bias         [128,283]
 *   [22],[23]     static_rgb.0.weight / .bias           [64,128]                                        */
#define CRNERF_MLP_TENSORS 24

/* Bytes of one packed model (consts + MFMA fragment stream). */
size_t crnerf_packed_mlp_bytes(void);

/* Re-pack the 24 nn.Linear tensors into the kernels' layout (padded K = 96/352/288, A-operand
 * fragment order).  Replaces nothing in the reference (it feeds at::addmm the raw tensors);
 * called once per weight update.  `tensors` is a HOST array of 24 device pointers. */
int crnerf_pack_mlp_weights(const float* const* tensors, void* packed, void* stream);

/* PosEmbedding.forward, models/nerf.py:17-30 (logscale freqs 2^0..2^(F-1)): x[n,3] -> out[n,6F+3]. */
int crnerf_posenc_f32(const float* x, float* out, int64_t n, int n_freqs, void* stream);

/* The MLP input of one pass of render_rays_cross_ray, models/rendering.py:100-114 (xyz_ = rays_o + rays_d * z_vals, embedding_xyz,
 * the repeat of dir_embedded, the cat): rays[R,8] (o, d, near, far), z[R,N], dir_emb[R,27] = crnerf_posenc_f32(view dirs, 4)
 * -> x[R*N,120], point index = ray*N + sample.  Used by the training backward, which rebuilds x instead of storing it. */
int crnerf_embed_points_f32(const float* rays, const float* z, const float* dir_emb, float* x, int64_t n_rays, int32_t n_samples,
                            void* stream);

/* NeRF_sigma.forward, models/nerf.py:157-182: x[n,120] -> out[n,65] (64 features then sigma);
 * sigma_only != 0: x[n,93] -> out[n,1] (models/nerf.py:159-160,173-174). */
int crnerf_mlp_forward_f32(const void* packed, const float* x, float* out, int64_t n, int sigma_only, void* stream);

/* ---- training twins of NeRF_sigma.forward (in the reference: PyTorch autograd over the 11 addmm nodes of
 * models/nerf.py:157-182).  Forward that keeps the layer activations, then backward = data gradient on a
 * transposed weight stream + weight/bias gradients as point-reduction GEMMs.
 *   acts    : crnerf_mlp_train_acts_bytes(n) bytes, written by forward_train, read by backward
 *   scratch : crnerf_mlp_train_scratch_bytes(n) bytes (layer deltas + reduction partials)
 *   grads   : HOST array of 24 device pointers (tensor order above), each OVERWRITTEN with d(sum(out*d_out))/d(tensor)
 * out / d_out are [n,65] (64 features then sigma).  x is not differentiated (embeddings are inputs). */
size_t crnerf_packed_mlp_t_bytes(void);
int crnerf_pack_mlp_weights_t(const float* const* tensors, void* packed_t, void* stream);
size_t crnerf_mlp_train_acts_bytes(int64_t n);
size_t crnerf_mlp_train_scratch_bytes(int64_t n);
int crnerf_mlp_forward_train_f32(const void* packed, const float* x, float* out, void* acts, int64_t n, void* stream);
int crnerf_mlp_backward_f32(const void* packed_t, const float* x, const float* out, const float* d_out, const void* acts, void* scratch,
                            float* const* grads, int64_t n, void* stream);
/* Same with option flags (0 = the call above).  CRNERF_BWD_WGRAD_BF16: the weight gradients of every nn.Linear except static_sigma
 * are computed from the SAME fp32 deltas and activations / embeddings rounded to bf16 (RNE) in registers, on the bf16 MFMA with fp32 accumulation over the points -- an opt-in mixed
 * precision with no counterpart in the reference (its autograd is fp32): data gradients, biases, every other tensor and the loss
 * stay exact fp32; each affected dW entry carries an unbiased rounding noise of ~2^-8 / sqrt(points) relative to its terms. */
#define CRNERF_BWD_WGRAD_BF16 1
/* CRNERF_BWD_WGRAD_BF16X3: fp32-ACCURATE weight gradients on the bf16 matrix cores.  Each fp32 operand of the 256 x 256 blocks is split in
 * registers into three bf16 pieces (x = x1 + x2 + x3, 24 mantissa bits) and every product is the sum of the six leading piece products, exact
 * in fp32 and accumulated in fp32; what is dropped is <= 3 x 2^-24 of a product -- one fp32 rounding.  Applies to every weight-gradient
 * block of the eleven nn.Linear (static_sigma: only when the jobs share one batched launch, n <= 2^18); bias gradients, data gradients
 * and everything else: the exact fp32 path.  Exclusive with CRNERF_BWD_WGRAD_BF16. */
#define CRNERF_BWD_WGRAD_BF16X3 2
/* CRNERF_BWD_WGRAD_F16X2 (crnerf_mlp_backward_h2_f32 only; exclusive with the two above): CRNERF_BWD_WGRAD_BF16X3 with the full 256 x 256 blocks
 * -- eight of the thirteen products of a backward -- formed from TWO fp16 pieces per operand and three piece products (what is dropped is <= 2^-22
 * of a product: the h2 core's arithmetic): half the matrix instructions.  Both operands go in under ONE power of two per tensor group: a delta
 * tensor under 2^(13 - e) of its largest entry, which the h2 data gradient of the same call leaves in the scratch; the activation / embedded-input
 * operand under 2^(14 - e) of the largest |operand| of the whole pass, which crnerf_render_rays_train_f32h2 leaves behind the saved rows (the
 * "range word": the last 256 bytes of crnerf_mlp_train_acts_bytes(n); zero when the fp32 / f32x3 twin wrote the rows -- the operand then goes
 * in unscaled and values below 2^-14 keep fewer than 22 bits, an ABSOLUTE piece error of 2^-25).  With the word an operand 2^-39 of the pass's
 * largest still lands on a piece bit; what is dropped is then <= 2^-22 of a product relative to the largest operands of its tensor, not more.  A
 * workgroup that meets an operand outside fp16's range (rows of a ray the forward had to repair; the f32x3 stand-in ran and left no range) redoes
 * its chunk of points as CRNERF_BWD_WGRAD_BF16X3 would have, bit for bit. */
#define CRNERF_BWD_WGRAD_F16X2 4
/* Phases (crnerf_mlp_backward_h2_f32 / _x3_f32 / _ex_f32; neither bit or both = the whole backward in one call).  CRNERF_BWD_PHASE_DGRAD: the data
 * gradient only -- the layer deltas, d_rgb, d_sig and (F16X2) the delta range words are left in `scratch`; `x` and `grads` are not touched and may be
 * NULL.  CRNERF_BWD_PHASE_WGRAD: the weight / bias gradients only, from a `scratch` that a PHASE_DGRAD call with the same n, acts and weight-gradient
 * mode has filled; packed_t_* / out / d_out are not read and may be NULL.  The two halves may run on different streams (the caller orders them with an
 * event): the weight gradients of one ray chunk read rows at the HBM read rate while the data gradient of the next writes its rows -- DESIGN 3.5. */
#define CRNERF_BWD_PHASE_DGRAD 8
#define CRNERF_BWD_PHASE_WGRAD 16
int crnerf_mlp_backward_ex_f32(const void* packed_t, const float* x, const float* out, const float* d_out, const void* acts, void* scratch,
                               float* const* grads, int64_t n, int flags, void* stream);

/* ---- opt-in mixed-precision training twins (no counterpart in the reference, whose autograd is fp32).  Same interface
 * as the fp32 twins above -- x[n,120], out[n,65], grads in fp32 -- but every nn.Linear except static_sigma runs as one points x
 * features GEMM on the bf16 MFMA: operands rounded to bf16 (RNE), fp32 accumulation, fp32 biases / activations / sigma head; and
 * what is kept between the passes is bf16: `acts` (crnerf_mlp_train_mixed_acts_bytes: the ten activation rows, relu bits, the
 * embedded input) and the layer deltas in `scratch` (crnerf_mlp_train_mixed_scratch_bytes) -- 5.7 + 5.3 KB per point against
 * 10.6 + 10.5 KB of the fp32 twins.  Forward = the arithmetic of crnerf_mlp_forward_bf16; backward = data gradients through the
 * transposed matrices, weight gradients from the stored rows, bias gradients as column sums of the stored (rounded) deltas.
 * `tensors` = the 24 fp32 tensors (biases and the sigma head are read from them), `packed_mixed` = crnerf_pack_mlp_weights_mixed
 * of the same tensors.  Both buffers are opaque to the caller (rows are stored in a kernel-internal feature order). */
size_t crnerf_mlp_train_mixed_acts_bytes(int64_t n);
size_t crnerf_mlp_train_mixed_scratch_bytes(int64_t n);
size_t crnerf_packed_mlp_mixed_bytes(void);
int crnerf_pack_mlp_weights_mixed(const float* const* tensors, void* packed_mixed, void* stream);
int crnerf_mlp_forward_train_mixed_f32(const float* const* tensors, const void* packed_mixed, const float* x, float* out, void* acts, int64_t n,
                                       void* stream);
int crnerf_mlp_backward_mixed_f32(const float* const* tensors, const void* packed_mixed, const float* x, const float* out, const float* d_out,
                                  const void* acts, void* scratch, float* const* grads, int64_t n, void* stream);
/* The same backward for either producer of `acts`: CRNERF_MIXED_ACTS_GEMM = crnerf_mlp_forward_train_mixed_f32 (as the call above),
 * CRNERF_MIXED_ACTS_FUSED = crnerf_render_rays_train_bf16 below (the fused renderer's training twin keeps its rows in the order its
 * registers hold them and the embedded input as the MLP multiplied it; x is not needed -- the embedded input is part of `acts`). */
#define CRNERF_MIXED_ACTS_GEMM 0
#define CRNERF_MIXED_ACTS_FUSED 1
int crnerf_mlp_backward_mixed_ex_f32(const float* const* tensors, const void* packed_mixed, const float* out, const float* d_out, const void* acts,
                                     void* scratch, float* const* grads, int64_t n, int acts_layout, void* stream);

/* Compositing part of the nested inference(), models/rendering.py:116-143:
 * raw[R,N,65], z[R,N], optional noise[R,N] (scaled by noise_std) -> weights[R,N], feature[R,64], depth[R]. */
int crnerf_composite_f32(const float* raw, const float* z, const float* noise, float noise_std, float* weights,
                         float* feature, float* depth, int64_t R, int N, void* stream);

/* Backward of the compositing above (what autograd derives from models/rendering.py:121-143 in the reference):
 * d_feature[R,64] (required), d_depth[R] / d_weights[R,N] (optional, NULL = zero) -> d_raw[R,N,65].
 * N <= 2035: a workgroup keeps five N-float rows per ray for its four rays in the 160 KiB LDS (larger N returns CRNERF_ERR_SHAPE). */
int crnerf_composite_backward_f32(const float* raw, const float* z, const float* noise, float noise_std, const float* d_feature,
                                  const float* d_depth, const float* d_weights, float* d_raw, int64_t R, int N, void* stream);

/* sample_pdf + merge, models/rendering.py:7-46 and :183-187:
 * z_coarse[R,Nc], weights_coarse[R,Nc] (the full coarse weights; [:,1:-1] is taken inside),
 * u: per-ray uniforms [R,Ni] (u_stride = Ni), one shared row [Ni] (u_stride = 0; e.g. torch.linspace(0,1,Ni)),
 * or NULL (in-kernel linspace) -> z_sorted[R,Nc+Ni]; z_samples[R,Ni] optional (NULL to skip). */
int crnerf_sample_pdf_merge_f32(const float* z_coarse, const float* weights_coarse, const float* u, int64_t u_stride, float* z_sorted,
                                float* z_samples, int64_t R, int Nc, int Ni, void* stream);

/* render_rays_cross_ray, models/rendering.py:50-196, fully fused (coarse -> sample_pdf -> fine). */
typedef struct crnerf_render_args {
  const void* packed_coarse;    /* crnerf_pack_mlp_weights output for models['coarse'] */
  const void* packed_fine;      /* ... for models['fine']; may be NULL when n_importance == 0 */
  const float* rays;            /* [R,8] = o(3) d(3) near far, rendering.py:152-153 */
  const float* view_dir;        /* [R,3] or NULL -> rays[:,3:6], rendering.py:155 */
  const float* z_coarse;        /* [R,Nc] or NULL -> rendering.py:161-165 computed in-kernel (perturb == 0) */
  const float* z_steps;         /* [Nc] = torch.linspace(0,1,Nc) (rendering.py:160) or NULL -> same formula in-kernel */
  const float* u;               /* rendering.py:26-31: [R,Ni] uniforms (u_stride = Ni), one shared row (u_stride = 0,
                                   e.g. torch.linspace(0,1,Ni) when det) or NULL -> in-kernel linspace */
  int64_t u_stride;
  const float* noise_coarse;    /* [R,Nc] standard-normal or NULL, rendering.py:125 */
  const float* noise_fine;      /* [R,Nc+Ni] or NULL */
  float noise_std;
  int32_t use_disp;
  int64_t n_rays;
  int32_t n_samples;            /* Nc in [2,256] */
  int32_t n_importance;         /* Ni in [0,256] */
  float* weights_coarse;        /* [R,Nc] */
  float* feature_coarse;        /* [R,64] */
  float* depth_coarse;          /* [R] */
  float* weights_fine;          /* [R,Nc+Ni]  (ignored when Ni == 0) */
  float* feature_fine;          /* [R,64] */
  float* depth_fine;            /* [R] */
  float* z_fine;                /* [R,Nc+Ni] optional debug/test output, NULL to skip */
  /* ---- in-kernel random draws (ABI 2; crnerf_render_rays_f32 / _f32x3 / _f32h2 and their _train_ twins; the bf16 entry points and the *_out pointers
   * below on kernels that do not write them return CRNERF_ERR_CONFIG).  rng_flags == 0: everything
   * above is used as given.  Draws are Philox4x32-10 keyed on rng_seed, counter = (sample, stream, rng_ray_offset + ray): a pure
   * function of the GLOBAL ray index, so ray chunks and the backward's recomputation see the same numbers.  The same draws as
   * tensors: crnerf_rng_fill_f32 (feeding them through z_coarse / u / noise_* gives bit-identical results). */
  uint64_t rng_seed;
  int64_t rng_ray_offset;       /* index of rays[0] in the caller's batch */
  int32_t rng_flags;            /* CRNERF_RNG_* below */
  float perturb;                /* jitter amplitude, rendering.py:175 (CRNERF_RNG_JITTER) */
  float* z_coarse_out;          /* [R,Nc] optional: the coarse depths used (the backward composites at them) */
  float* noise_coarse_out;      /* [R,Nc] optional: the standard-normal draws used by the coarse pass (CRNERF_RNG_NOISE) */
  float* noise_fine_out;        /* [R,Nc+Ni] optional */
} crnerf_render_args;
#define CRNERF_RNG_JITTER 1     /* stratified jitter of the coarse depths, rendering.py:169-176: z_coarse must be NULL */
#define CRNERF_RNG_U 2          /* sample_pdf's uniforms (det = False), rendering.py:30: u must be NULL */
#define CRNERF_RNG_NOISE 4      /* density noise randn * noise_std, rendering.py:125: noise_coarse / noise_fine must be NULL */
/* The draws of stream `stream` (0 jitter uniforms, 1 sample_pdf uniforms, 2 coarse noise, 3 fine noise) as a tensor:
 * out[r * n + s] = draw(seed, stream, ray_offset + r, s); streams 0 / 1 are U[0,1) on the 2^-24 grid, 2 / 3 standard normal. */
int crnerf_rng_fill_f32(float* out, int64_t n_rays, int n, uint64_t seed, int stream, int64_t ray_offset, void* stream_handle);
int crnerf_render_rays_f32(const crnerf_render_args* args, void* stream);

/* Training twin of the call above (the reference trains THROUGH render_rays_cross_ray under autograd, rendering.py:100-143):
 * the same fused launch -- positional encoding, both NeRF_sigma passes, compositing, sample_pdf + merge -- that additionally
 * keeps what the backward twins need: per pass the layer activations in the crnerf_mlp_forward_train_f32 layout
 * (acts_*: crnerf_mlp_train_acts_bytes(R*N) bytes, point index = ray*N + sample) and the raw MLP outputs raw_*[R*N,65].
 * args->z_fine is REQUIRED when n_importance > 0 (the backward composites at those depths).  Backward = per pass
 * crnerf_composite_backward_f32 (raw, z, noise -> d_raw) then crnerf_mlp_backward_f32 (x = the embedded points, which the
 * caller rebuilds from rays and z with crnerf_posenc_f32 -- they are never stored between forward and backward). */
int crnerf_render_rays_train_f32(const crnerf_render_args* args, void* acts_coarse, void* acts_fine, float* raw_coarse,
                                 float* raw_fine, void* stream);

/* ---- bf16 matrix-core variants (BASELINE config 3: "1x MI355X bf16"; SURVEY 8b minimum export set
 * crnerf_mlp_forward_{f32,bf16} / crnerf_render_rays_{f32,bf16}).  Same reference functions, mixed precision:
 * the operands of every nn.Linear of NeRF_sigma except static_sigma -- weights and input activations, including the
 * two positional embeddings -- are rounded to bf16 (round-to-nearest-even); products accumulate in fp32; biases,
 * ReLU, Softplus, Sigmoid, the sigma head (fp32 weights on the un-rounded output of xyz_encoding_8), ray geometry,
 * compositing, sample_pdf and the z merge stay fp32.  I/O tensors are fp32 exactly as in the _f32 entry points.
 * The packed buffer is a different layout (fp32 consts + bf16 fragments): pack with crnerf_pack_mlp_weights_bf16. */
size_t crnerf_packed_mlp_bf16_bytes(void);
int crnerf_pack_mlp_weights_bf16(const float* const* tensors, void* packed_bf16, void* stream);
/* NeRF_sigma.forward, models/nerf.py:157-182: x[n,120] -> out[n,65]; sigma_only != 0: x[n,93] -> out[n,1]. */
int crnerf_mlp_forward_bf16(const void* packed_bf16, const float* x, float* out, int64_t n, int sigma_only, void* stream);
/* render_rays_cross_ray, models/rendering.py:50-196, fully fused; args->packed_{coarse,fine} are bf16 packs. */
int crnerf_render_rays_bf16(const crnerf_render_args* args, void* stream);
/* The FINE half of render_rays_cross_ray on the bf16 matrix cores (models/rendering.py:183-194: sample_pdf on the coarse weights, the z merge, the
 * fine model over the N_samples + N_importance merged depths, compositing) for a coarse pass rendered elsewhere -- precision "bf16_hc" renders it
 * with crnerf_render_rays_f32h2 / _f32x3_repair (n_importance = 0), so that the fine DEPTHS are the fp32 reference's and only the fine network's
 * own products are bf16.  args->weights_coarse [R, n_samples] is READ; args->packed_fine is a bf16 pack; packed_coarse, feature_coarse and
 * depth_coarse are ignored (may be NULL); z_coarse / z_steps / u / noise_fine as in crnerf_render_rays_bf16; writes weights_fine, feature_fine,
 * depth_fine and (optional) z_fine.  Nothing per point goes through HBM. */
int crnerf_render_rays_bf16_fine(const crnerf_render_args* args, void* stream);

/* Training twin of crnerf_render_rays_bf16 for the opt-in mixed-precision training mode (no counterpart in the reference): the same
 * fused launch on the bf16 matrix cores that additionally keeps, per pass, what crnerf_mlp_backward_mixed_ex_f32(..., CRNERF_MIXED_ACTS_FUSED)
 * needs -- acts_*: crnerf_mlp_train_mixed_acts_bytes(R*N) bytes (bf16 activation rows, relu-activity bits, the embedded input; point index
 * = ray*N + sample), raw_*[R*N,65] fp32 MLP outputs -- written from the registers the values are born in: no embedded-input tensor, no
 * per-layer round trip of the activations.  args->z_fine is REQUIRED when n_importance > 0.  R*(n_samples + n_importance) < 2^23 per call.
 * Random draws come as tensors (z_coarse / u / noise_*); rng_flags must be 0.  Backward = per pass crnerf_composite_backward_f32 then
 * crnerf_mlp_backward_mixed_ex_f32. */
int crnerf_render_rays_train_bf16(const crnerf_render_args* args, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine,
                                  void* stream);

/* ---- "f32x3": the fp32 entry points evaluated on the bf16 matrix cores (no counterpart in the reference; opt-in).  Same functions, same
 * fp32 inputs / outputs / biases / activations / sigma head / embeddings as crnerf_mlp_forward_f32 and crnerf_render_rays_f32; only the
 * products of the eleven nn.Linear are formed differently: each fp32 operand is split into three bf16 pieces (w = w1 + w2 + w3, 24 mantissa
 * bits; weights at pack time, activations in registers) and a product is the sum of the six leading piece products, each exact in fp32 and
 * accumulated in fp32 -- the dropped terms are <= 3 x 2^-24 of a product, one fp32 rounding.  The results meet the fp32 entry points'
 * tolerances against the oracle (tests/test_gpu_x3.py); they are not bit-identical to the fp32 MFMA's.  The split is scale-free (bf16 has fp32's
 * exponent range); non-finite operands give NaN where the fp32 MFMA gives +-inf.  packed = crnerf_pack_mlp_weights_x3. */
size_t crnerf_packed_mlp_x3_bytes(void);
int crnerf_pack_mlp_weights_x3(const float* const* tensors, void* packed_x3, void* stream);
int crnerf_mlp_forward_f32x3(const void* packed_x3, const float* x, float* out, int64_t n, int sigma_only, void* stream);
/* render_rays_cross_ray, models/rendering.py:50-196, fully fused; args->packed_{coarse,fine} are x3 packs; rng_flags as in
 * crnerf_render_rays_f32 (the same counters: one seed, one set of draws, whichever kernel renders). */
int crnerf_render_rays_f32x3(const crnerf_render_args* args, void* stream);
/* ---- "f32h2": the same idea with TWO fp16 pieces per operand and THREE piece products (w2 a1 + w1 a2 + w1 a1; the dropped w2 a2 is <= 2^-24 of the
 * product): x = h1 + h2, h1 = fp16(x), h2 = fp16(x - h1) -- 11 + 11 mantissa bits and the sign of h2, one fp32 rounding while h2 is a normal fp16
 * number, an absolute error <= 2^-25 below that (the matrix cores honour fp16 subnormals).  Half the MFMAs and two thirds of the weight stream of
 * f32x3.  NOT scale-free: the packed weights are scaled by 2^8 (exactly undone in registers) so that their second pieces stay normal, and the
 * operands must fit fp16's range -- |weight| < 255, |activation| < 65,504 -- or the outputs are inf / nan (loud, not silently wrong).  Inference
 * entry points only; held to the fp32 entry points' goldens and tolerances (tests/test_gpu_h2.py).  packed = crnerf_pack_mlp_weights_h2. */
size_t crnerf_packed_mlp_h2_bytes(void);
/* Returns CRNERF_ERR_RANGE when a weight is not a finite number with |w| < 255.  To know that, this call WAITS for `stream` (one 4-byte read-back:
 * the only entry point of the library that synchronises; it cannot be stream-captured).  Packing happens once per set of weights. */
int crnerf_pack_mlp_weights_h2(const float* const* tensors, void* packed_h2, void* stream);
int crnerf_mlp_forward_f32h2(const void* packed_h2, const float* x, float* out, int64_t n, int sigma_only, void* stream);
int crnerf_render_rays_f32h2(const crnerf_render_args* args, void* stream);   /* args->packed_* are h2 packs; rng_flags as in crnerf_render_rays_f32 */
/* ---- "auto" = f32h2 with the scale-free f32x3 core as its safety net (the reference, models/nerf.py:157-182, has no range limit).  The h2 core
 * turns the outputs of a point whose activations leave fp16's range into NaN; a ray with such a point comes out of crnerf_render_rays_f32h2 with
 * NaN features.  The *_repair entry points take the SAME arguments as the h2 call that went before (outputs included) with f32x3 packs of the same
 * weights, and re-evaluate on the x3 core exactly the units that hold a NaN: ray quads (4 consecutive rays) whose feature_coarse / feature_fine
 * start with a NaN, 128-point groups with a NaN sigma.  One workgroup per unit, which leaves at once when there is nothing to repair (~5 us per
 * 1,024 rays); asynchronous, no host round trip, same in-kernel random draws.  After the pair of calls no output is NaN unless the fp32 function
 * itself is.  (A pack refused with CRNERF_ERR_RANGE: call the f32x3 entry points instead -- crnerf_amd's precision="auto" does both.) */
int crnerf_render_rays_f32x3_repair(const crnerf_render_args* args, void* stream);
int crnerf_mlp_forward_f32x3_repair(const void* packed_x3, const float* x, float* out, int64_t n, int sigma_only, void* stream);
/* Training twin: crnerf_render_rays_train_f32 on the x3 core -- the same saved state (acts_*: crnerf_mlp_train_acts_bytes(R*N) bytes in the
 * layout of crnerf_mlp_forward_train_f32, raw_*[R*N,65]), so the fp32 backward twins (crnerf_composite_backward_f32 ->
 * crnerf_mlp_backward[_ex]_f32, or crnerf_mlp_backward_x3_f32) follow unchanged.  args->packed_* are x3 packs; random draws as tensors or
 * in-kernel (rng_flags); R * (n_samples + n_importance) < 3.9 M per call. */
int crnerf_render_rays_train_f32x3(const crnerf_render_args* args, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine,
                                   void* stream);
/* crnerf_mlp_backward_ex_f32 with the DATA gradient on the x3 core (the layer deltas from three-piece splits of the transposed weights and of the
 * deltas; six bf16 MFMAs per product, fp32 accumulation): same acts / scratch buffers and layouts, same flags for the weight gradients.
 * packed_t_x3 = crnerf_pack_mlp_weights_t_x3 (crnerf_packed_mlp_t_x3_bytes).  n < 3.9 M points per call (CRNERF_ERR_SHAPE-class error beyond). */
size_t crnerf_packed_mlp_t_x3_bytes(void);
int crnerf_pack_mlp_weights_t_x3(const float* const* tensors, void* packed_t_x3, void* stream);
int crnerf_mlp_backward_x3_f32(const void* packed_t_x3, const float* x, const float* out, const float* d_out, const void* acts, void* scratch,
                               float* const* grads, int64_t n, int flags, void* stream);
/* ---- training on the h2 core (round 4): the f32x3 training twins' contracts with two fp16 pieces per operand and three MFMAs per product.
 * crnerf_render_rays_train_f32h2: args->packed_* are h2 packs; same saved state as crnerf_render_rays_train_f32 / _f32x3.  A ray with a point
 * whose activations left fp16's range comes out with NaN features (as from crnerf_render_rays_f32h2) and garbage saved rows;
 * crnerf_render_rays_train_f32x3_repair -- the SAME arguments with f32x3 packs of the same weights -- renders exactly those ray quads again on the
 * scale-free core, saved rows, raw rows and *_out draws included (one workgroup per quad, leaves at once when nothing is NaN; no host round trip).
 * crnerf_mlp_backward_h2_f32: crnerf_mlp_backward_x3_f32 with the data gradient on the h2 core; packed_t_h2 = crnerf_pack_mlp_weights_t_h2 (the
 * transposed weights as two fp16 pieces of 2^8 w; its range flag stays on the device, see crnerf_pack_mlp_weights_h2_async below).  Every
 * point's delta vector is rescaled by an exact power of two per layer so that its largest entry sits in [2^7, 2^8): gradients of any magnitude
 * (1e-30 ... 1e30) go through at fp32 accuracy, there is no range failure on this side.  n < 3.9 M points per call. */
int crnerf_render_rays_train_f32h2(const crnerf_render_args* args, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine,
                                   void* stream);
int crnerf_render_rays_train_f32x3_repair(const crnerf_render_args* args, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine,
                                          void* stream);
size_t crnerf_packed_mlp_t_h2_bytes(void);
int crnerf_pack_mlp_weights_t_h2(const float* const* tensors, void* packed_t_h2, void* stream);
int crnerf_mlp_backward_h2_f32(const void* packed_t_h2, const void* packed_t_x3, const float* x, const float* out, const float* d_out, const void* acts,
                               void* scratch, float* const* grads, int64_t n, int flags, void* stream);
/* The range verdict without a host round trip (training packs every step).  crnerf_pack_mlp_weights_h2_async packs like crnerf_pack_mlp_weights_h2 but
 * never waits for the stream and never returns CRNERF_ERR_RANGE: a weight outside |w| < 255 leaves a flag word in the pack itself, and so does
 * crnerf_pack_mlp_weights_t_h2.  The h2 kernels read it on the device: crnerf_render_rays[_train]_f32h2 then render nothing and mark every ray NaN
 * (the *_f32x3_repair call that precision "auto" issues next renders them all on the scale-free core), crnerf_mlp_backward_h2_f32 leaves its data
 * gradient to the f32x3 kernel when packed_t_x3 (optional, may be NULL) is given -- that kernel is launched behind the h2 one and exits at once unless
 * the flag is set.  crnerf_pack_h2_status reads the flag back (this one waits for the stream): 0 or CRNERF_ERR_RANGE. */
int crnerf_pack_mlp_weights_h2_async(const float* const* tensors, void* packed_h2, void* stream);
int crnerf_pack_h2_status(const void* packed_h2, void* stream);

/* Appearance encoder: encoder_sameoutputsize.forward, models/linearStyleTransfer.py:208-276.
 * image[3,H,W] (NCHW, values in [0,1]) -> out[1024,64], the pixel-major 32x32 style grid the decoder consumes.
 * weights = HOST array of 14 device pointers: conv1.weight, conv1.bias, ..., conv7.weight, conv7.bias (reference layouts). */
#define CRNERF_ENCODER_TENSORS 14
size_t crnerf_encoder_workspace_bytes(int H, int W);
int crnerf_encoder_forward_f32(const float* image, int H, int W, const float* const* weights, void* workspace, float* out, void* stream);

/* Ray generation on the device.  HOST pointers: intrinsics = {fx, fy, cx, cy}, c2w = 3x4 row-major.
 * get_ray_directions datasets/ray_utils.py:5-26 -> directions[H,W,3]; get_rays :29-52 -> rays_o/rays_d [n,3];
 * generate_rays = both + the near/far columns of datasets/PhototourismDataset.py:17-22 -> rays[H*W,8]. */
int crnerf_ray_directions_f32(int H, int W, float fx, float fy, float cx, float cy, float* directions, void* stream);
int crnerf_rays_from_directions_f32(const float* directions, const float* c2w_host, int64_t n, float* rays_o, float* rays_d, void* stream);
int crnerf_generate_rays_f32(const float* intrinsics_host, const float* c2w_host, int H, int W, float near, float far, float* rays,
                             void* stream);

/* Cross-ray transformation + decoder: style_net.forward models/linearStyleTransfer.py:284-291,
 * MulLayer.forward :58-94, CNN.forward :28-37, NeuralRenderer.forward nerf_decoder_stylenerf.py:279-291.
 * Feature grids are pixel-major x[HW,64]; split at the two global reductions (see crossray.hip). */
size_t crnerf_crossray_workspace_bytes(void);
/* per-channel sums over pixels (divide by the GLOBAL pixel count to get MulLayer's mean, :59-65/:67-73) */
int crnerf_crossray_chansum_f32(const float* x, int64_t HW, float* sum64, void* workspace, void* stream);
/* cnn[6] = HOST array of device pointers: convs.0.weight[128,64], .bias, convs.2.weight[64,128], .bias,
 * convs.4.weight[32,64], .bias.  gram_sum[32*32] = sum over pixels of f(x-mean) f(x-mean)^T  (CNN.forward :29-34
 * before the division by h*w). */
int crnerf_crossray_gram_f32(const float* x, int64_t HW, const float* mean64, const float* const* cnn, float* gram_sum,
                             void* workspace, void* stream);
/* out[1024] = fc(gram_sum / count)   (CNN.forward :34-37) */
int crnerf_crossray_matrix_f32(const float* gram_sum, double count, const float* fc_w, const float* fc_b, float* out,
                               void* stream);
/* lin[6] = HOST array of device pointers: compress.weight[32,64], .bias, unzip.weight[64,32], .bias,
 * feat_2_rgb_list.0.weight[3,64], .bias.  Folds MulLayer :76-89 + the 1x1 rgb conv into affine[195] =
 * A[3][64] then v[3].  s_matrix == NULL selects the type=="content" path (:285-287: decoder only). */
int crnerf_crossray_fold_f32(const float* s_matrix, const float* c_matrix, const float* c_mean64, const float* s_mean64,
                             const float* const* lin, float* affine, void* stream);
/* Single-GPU convenience: the whole of style_net.forward (chansum -> gram -> matrix -> fold -> apply for the
 * content grid and the style grid) enqueued from one call.  weights = HOST array of the 22 parameter tensors in
 * state_dict order: multi_net.snet.{convs.0,convs.2,convs.4,fc}.{weight,bias}, multi_net.cnet.(same),
 * multi_net.compress.{weight,bias}, multi_net.unzip.{weight,bias}, decoder.feat_2_rgb_list.0.{weight,bias}.
 * style == NULL selects type=="content".  rgb[c*plane_stride + px]. */
#define CRNERF_DECODER_TENSORS 22
int crnerf_crossray_decode_f32(const float* content, int64_t HW, const float* style, int64_t HWs, const float* const* weights,
                               void* workspace, float* rgb, int64_t plane_stride, void* stream);
/* Ray-sharded decode: the same math in three calls placed around the two all-reduces (multi-GPU, SURVEY 8e option B).
 * xchg[1088] device floats: [0:64] channel sums, [64:1088] Gram sums of the CONTENT grid.
 *   phase 0 writes the local channel sums into xchg[0:64]                              -> caller all-reduces xchg[0:64]
 *   phase 1 reads the global sums (+ count_global), writes local Gram sums to xchg[64:] -> caller all-reduces xchg[64:1088]
 *   phase 2 reads the global Gram, writes rgb for the local pixels.
 * HW_local may be 0 (a rank without pixels); the style grid is replicated on every rank. */
int crnerf_crossray_decode_sharded_f32(const float* content, int64_t HW_local, const float* style, int64_t HWs, const float* const* weights,
                                       int phase, float* xchg, double count_global, void* workspace, float* rgb, int64_t plane_stride,
                                       void* stream);
/* Peer-window all-reduce: an alternative carrier for the two reductions above (the reference has no counterpart: it is
 * single-GPU at inference, train_mask_grid_sample.py:441-450 uses Lightning DDP only for gradients).  Every rank creates one
 * window in its own HBM, hands the 64-byte HIP IPC handle to the other processes of the node (any host channel), opens theirs,
 * and then reduces n <= 1024 floats IN PLACE with one single-workgroup kernel per rank: push to every window, flag, bounded
 * wait, sum in rank order (bit-identical on all ranks).  `windows` = HOST array of world_size device pointers, windows[rank] = own.
 * `epoch` = 1, 2, 3, ... the same on all ranks for the same reduction, +1 per call.  A peer that does not arrive within
 * timeout_us leaves NaN in `data` and 1 + its rank in the window's status word (crnerf_peer_window_status; synchronises). */
/* ---- streams that own a share of the GPU's compute units.  The training backward runs its write-bound data gradient and its read-bound weight
 * gradients side by side (CRNERF_BWD_PHASE_*); every such kernel takes a whole CU per workgroup (512 registers per wave), so "side by side"
 * means a partition of the CUs, which HIP offers per stream.  crnerf_stream_create_cu_share: a stream whose kernels run on a fraction of every XCD's
 * CUs -- those with index (within the XCD's logical numbering) in [first, first + count) out of crnerf_cus_per_xcd().  Destroy with
 * crnerf_stream_destroy (synchronises the stream).  Any entry point of this library takes such a stream as its `stream` argument. */
int crnerf_cus_per_xcd(void);
int crnerf_stream_create_cu_share(void** stream, int first, int count);
int crnerf_stream_destroy(void* stream);
#define CRNERF_PEER_MAX_RANKS 8
#define CRNERF_PEER_MAX_FLOATS 1024
#define CRNERF_PEER_HANDLE_BYTES 64
size_t crnerf_peer_window_bytes(void);
int crnerf_peer_window_create(void** window, void* handle_out64);
int crnerf_peer_window_open(const void* handle64, void** window);
int crnerf_peer_window_close(void* opened_window);
int crnerf_peer_window_destroy(void* own_window);
int crnerf_peer_window_status(void* own_window, int* status);
int crnerf_peer_allreduce_f32(float* data, int n, void* const* windows, int rank, int world_size, uint32_t epoch, int64_t timeout_us,
                              void* stream);
/* Backward of crnerf_crossray_decode_f32 (in the reference: autograd through style_net.forward): d_rgb[c*d_plane_stride + px]
 * -> d_content[HW,64], d_style[HWs,64] and grads[22] (same order as `weights`, each OVERWRITTEN).
 * workspace: crnerf_crossray_backward_workspace_bytes(HW, HWs). */
size_t crnerf_crossray_backward_workspace_bytes(int64_t HW, int64_t HWs);
int crnerf_crossray_decode_backward_f32(const float* content, int64_t HW, const float* style, int64_t HWs, const float* const* weights,
                                        const float* d_rgb, int64_t d_plane_stride, void* workspace, float* d_content, float* d_style,
                                        float* const* grads, void* stream);
/* The same backward for a content grid that is ray-sharded over the ranks of a group (round 6; the training twin of
 * crnerf_crossray_decode_sharded_f32): three phases around two all-reduces, driven by the caller (parallel.DecodeShardedFn).
 *   fwd_xchg[64 + 1024] : the forward's GLOBAL channel sums and Gram sums (what its two all-reduces delivered); count_global = the global pixel count
 *   xb[384]             : exchange buffer.  phase 0 leaves this rank's dA [3,64] (ld 64) | dv [3] in xb[0:256] | xb[256:320] -> all-reduce xb[0:320];
 *                         phase 1 leaves the column sums of the centred chain's input gradient in xb[320:384] -> all-reduce; phase 2 finishes.
 * The workspace (crnerf_crossray_backward_workspace_bytes(HW, HWs), HW = this rank's pixels) carries the state between the phases.  grads[8..13] (the
 * content chain's three convolutions) come out as this rank's PART -- sums over its pixels; every other gradient and d_style are whole on every rank. */
int crnerf_crossray_decode_backward_sharded_f32(const float* content, int64_t HW, const float* style, int64_t HWs, const float* const* weights,
                                                const float* d_rgb, int64_t d_plane_stride, void* workspace, float* d_content, float* d_style,
                                                float* const* grads, int phase, const float* fwd_xchg, double count_global, float* xb, void* stream);
/* Backward of the decoder-only call style_net.forward(content, None, type="content") (linearStyleTransfer.py:285-287;
 * forward = crnerf_crossray_decode_f32 with style == NULL): rgb = sigmoid(W x + b) with W = decoder.feat_2_rgb_list.0.weight
 * [3,64].  rgb / d_rgb planar with their strides -> d_content[HW,64], d_w[3,64], d_b[3]. */
size_t crnerf_decoder_content_backward_workspace_bytes(int64_t HW);
int crnerf_decoder_content_backward_f32(const float* content, int64_t HW, const float* rgb_w, const float* rgb, int64_t rgb_plane_stride,
                                        const float* d_rgb, int64_t d_plane_stride, void* workspace, float* d_content, float* d_w, float* d_b,
                                        void* stream);
/* rgb[c*plane_stride + px] = sigmoid(A[c] . x[px] + v[c]) */
int crnerf_crossray_apply_f32(const float* x, int64_t HW, const float* affine, float* rgb, int64_t plane_stride,
                              void* stream);

/* Training twins of the appearance encoder (the reference trains enc_a through PyTorch autograd,
 * train_mask_grid_sample.py:95-97, :160, :219): a forward that keeps every layer output in `saved`, and the backward:
 * grads[14] in the order of `weights`, d_image[3,H,W] optional (NULL to skip; needed where the encoder reads the
 * re-rendered image).  `out` is the forward's output (its sign carries the last LeakyReLU's derivative). */
size_t crnerf_encoder_train_saved_bytes(int H, int W);
size_t crnerf_encoder_train_scratch_bytes(int H, int W);
int crnerf_encoder_forward_train_f32(const float* image, int H, int W, const float* const* weights, void* saved, float* out, void* stream);
int crnerf_encoder_backward_f32(int H, int W, const float* const* weights, const void* saved, const float* out, const float* d_out,
                                void* scratch, float* const* grads, float* d_image, void* stream);
/* The same twins over a BAND of rows of an image (round 6: in ray-parallel training the encoder passes over the re-rendered H_image x W images are
 * split into row bands, one per rank -- DESIGN 4).  image_rows[3,H,W] = rows [row0, row0 + H) of the image (contiguous; row0 a multiple of 4); the
 * seven layers run on the band as on an image of H rows, so the band must bring a halo of >= 12 rows beyond the rows whose outputs are wanted
 * wherever it is cut inside the image (10 rows = the layers' receptive reach; what the reflection padding gets wrong at a cut edge then never
 * reaches them).  Only the final AdaptiveAvgPool2d(32) knows the whole image: out[(o1 - o0) * 32, 64] = rows [o0, o1) of the 32 x 32 style grid,
 * with the whole image's pooling windows (which must lie inside the band).  Backward: d_out[(o1 - o0) * 32, 64] -> grads[14] (this band's part
 * of the weight gradients: the bands' results ADD UP to the whole image's) and d_image_rows[3,H,W] (optional): what the band's output rows
 * contribute to the gradient of the image rows it read -- halo rows included, the caller sums overlapping bands.  n_out_rows = o1 - o0. */
size_t crnerf_encoder_train_band_saved_bytes(int H, int W, int n_out_rows);
size_t crnerf_encoder_train_band_scratch_bytes(int H, int W, int n_out_rows);
int crnerf_encoder_forward_train_band_f32(const float* image_rows, int H, int W, int H_image, int row0, int o0, int o1, const float* const* weights, void* saved,
                                          float* out, void* stream);
int crnerf_encoder_backward_band_f32(int H, int W, int H_image, int row0, int o0, int o1, const float* const* weights, const void* saved, const float* out,
                                     const float* d_out, void* scratch, float* const* grads, float* d_image_rows, void* stream);

/* ---- training-side neighbours of the path ------------------------------------------------------
 * CRNeRFLoss.forward, losses.py:49-78 (mask_regularize :80-91, _l2_regularize :93-96).  The seven terms, in the
 * order losses[0..6] = kl_a, rec_a_random, c_l, content_constraint, r_ms, r_md, f_l; a term whose inputs are absent
 * (NULL pointers) is 0 and must be left out of the dict by the caller, as the reference does.  rgb tensors are [R,3]
 * with explicit element strides (row, channel), so the decoder's planar [3,R] output is consumed in place.
 * mask: out_mask[R] (it is .detach()-ed inside c_l only, losses.py:63 vs :70). */
typedef struct crnerf_loss_args {
  const float* rgb_coarse; int64_t rgb_coarse_row_stride, rgb_coarse_chan_stride;
  const float* rgb_fine;   int64_t rgb_fine_row_stride, rgb_fine_chan_stride;      /* NULL: no 'rgb_fine' */
  const float* targets;    int64_t targets_row_stride, targets_chan_stride;
  const float* mask;                                                               /* NULL: no 'out_mask' */
  const float* a_embedded; int64_t n_a;                                            /* NULL: no 'a_embedded' */
  const float* a_embedded_random; const float* a_embedded_random_rec; int64_t n_rec;
  const float* content_wo; const float* content_with; int64_t n_content;
  int64_t n_rays;
  int32_t mse_on_appearance;
  float coef, weight_kl, weight_rec_a, weight_content;   /* CRNeRFLoss.coef, hparams.weightKL / weightRecA / weightcontent */
  float mask_size_weight;                                 /* ExponentialAnnealingWeight.getWeight(global_step), losses.py:38-39 */
  float mask_digit_weight;                                /* hparams.maskrd */
} crnerf_loss_args;
typedef struct crnerf_loss_grads {                        /* contiguous outputs; any may be NULL */
  float* d_rgb_coarse; float* d_rgb_fine; float* d_mask;  /* [R,3] [R,3] [R] */
  float* d_a_embedded; float* d_a_embedded_random_rec; float* d_content_wo; float* d_content_with;
} crnerf_loss_grads;
#define CRNERF_LOSS_TERMS 7
size_t crnerf_loss_workspace_bytes(void);
int crnerf_loss_f32(const crnerf_loss_args* args, float* losses, void* workspace, void* stream);
/* upstream[7] (device): d total / d losses[k] -- what autograd hands back for the seven scalars */
int crnerf_loss_backward_f32(const crnerf_loss_args* args, const float* upstream, const crnerf_loss_grads* grads, void* stream);

/* Grid-sample batcher: PhototourismDataset.__getitem__ (train), datasets/phototourism_mask_grid_sample.py:241-275.
 * Cuts a side x side lattice batch out of image `row_offset / (w*h)` of the flat, HBM-resident buffers all_rays[N,
 * ray_stride >= 9] (o, d, near, far, image id) and all_rgbs[N,3].  w_lin / h_lin: the reference's two linspace
 * tables (torch.linspace(0, 1-1/w, side), :249-250); scale / offsets: its three uniform draws (:255-257), made by the
 * host so that the RNG stream is the reference's.  Outputs: rays[side^2,8], ts[side^2] (int64), rgbs[side^2,3],
 * rgb_idx[side^2] (int64, pixel index inside the image), uv_sample[side^2,2] = (h, w) lattice coordinates. */
typedef struct crnerf_batch_args {
  const float* all_rays; int64_t ray_stride; const float* all_rgbs; int64_t row_offset;
  int32_t img_w, img_h, side;
  const float* w_lin; const float* h_lin;
  float scale, h_offset, w_offset;
  float* rays; int64_t* ts; float* rgbs; int64_t* rgb_idx; float* uv_sample;
} crnerf_batch_args;
int crnerf_grid_sample_batch_f32(const crnerf_batch_args* args, void* stream);

/* -------- the optimiser step: torch.optim.Adam(get_parameters(models), lr, eps=1e-8, weight_decay) -- utils/__init__.py:24-33,
 * stepped once per batch (train_mask_grid_sample.py:249-252).  One launch over flat parameter / first-moment / second-moment
 * buffers that share offsets; gradients stay where they are (grads[t] = device pointer of tensor t's gradient, null = that
 * parameter has no gradient this step and is skipped, as torch skips `p.grad is None`).  blocks[n_blocks][4] (device, int32):
 * {tensor index, flat offset, element count, offset inside the tensor} per workgroup -- built once by the host.
 * step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) for step t (1-based), computed by the host in
 * double like torch/optim/adam.py.  n_tensors <= crnerf_adam_max_tensors() per call. */
int crnerf_adam_max_tensors(void);
int crnerf_adam_step_f32(float* params, float* exp_avg, float* exp_avg_sq, const int32_t* blocks, int32_t n_blocks,
                         const float* const* grads, int32_t n_tensors, float step_size, float beta1, float beta2, float eps,
                         float weight_decay, float bias_correction2_sqrt, void* stream);

/* Operators of the transient-mask network: Context_Guided_Network(classes=1, M=2, N=2, input_channel=3),
 * models/lightweight_seg.py:274-368, applied once per step to the 1/8-scale photo (train_mask_grid_sample.py:170-176).
 * All tensors NCHW fp32, batch 1, contiguous; every backward OVERWRITES its gradient outputs. */
typedef struct crnerf_conv_geom {           /* nn.Conv2d(cin, cout, k, stride, padding=pad, dilation=dil, bias=False); */
  int32_t cin, cout, H, W;                  /* groups = 1, or groups = cin = cout when depthwise != 0                   */
  int32_t k, stride, pad, dil, depthwise;   /* (Conv / ChannelWiseConv / ChannelWiseDilatedConv, lightweight_seg.py:12-140) */
} crnerf_conv_geom;
/* x[cin,H,W], w[cout, cin or 1, k, k] -> y[cout,Ho,Wo], Ho = (H + 2 pad - dil (k-1) - 1) / stride + 1 */
int crnerf_conv2d_f32(const crnerf_conv_geom* geom, const float* x, const float* w, float* y, void* stream);
/* d_x may be NULL (first layer) */
int crnerf_conv2d_backward_f32(const crnerf_conv_geom* geom, const float* x, const float* w, const float* d_y, float* d_x, float* d_w,
                               void* stream);
/* BatchNorm2d(C, eps) followed by PReLU(C) (ConvBNPReLU / BNPReLU, lightweight_seg.py:12-52).  training != 0: batch
 * statistics -- mean[C], invstd[C] (biased variance) and var_unbiased[C] are WRITTEN (the host applies the momentum
 * update of running_mean / running_var with them); training == 0: mean / invstd are READ (running statistics). */
int crnerf_bn_prelu_f32(const float* x, const float* gamma, const float* beta, const float* alpha, float* mean, float* invstd,
                        float* var_unbiased, float* y, int C, int64_t HW, float eps, int training, void* stream);
/* The training-mode call that ALSO applies nn.BatchNorm2d's momentum update in the same launch (lightweight_seg.py BatchNorm2d defaults:
 * momentum 0.1): running = running * (1 - momentum) + momentum * batch statistic (unbiased variance), num_batches_tracked += 1 (may be NULL). */
int crnerf_bn_prelu_train_f32(const float* x, const float* gamma, const float* beta, const float* alpha, float* mean, float* invstd,
                              float* var_unbiased, float* y, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float momentum, int C, int64_t HW, float eps, void* stream);
int crnerf_bn_prelu_backward_f32(const float* x, const float* gamma, const float* beta, const float* alpha, const float* mean,
                                 const float* invstd, const float* d_y, float* d_x, float* d_gamma, float* d_beta, float* d_alpha, int C,
                                 int64_t HW, int training, void* stream);
/* nn.AvgPool2d(3, stride=2, padding=1) (InputInjection, lightweight_seg.py:258-270): x[C,H,W] -> y[C,(H-1)/2+1,(W-1)/2+1];
 * backward != 0: `in` is d_y and `out` d_x[C,H,W] (H, W always the un-pooled size). */
int crnerf_avgpool3s2_f32(const float* in, float* out, int C, int H, int W, int backward, void* stream);
/* FGlo (lightweight_seg.py:143-162): y = x * sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2), W1[R,C], W2[C,R], C <= 256, R <= 64.
 * stats[2C+R] (means, hidden, gates) is written by the forward and read by the backward; scratch: 2C floats. */
int crnerf_fglo_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* stats, float* y, int C, int R,
                    int64_t HW, void* stream);
int crnerf_fglo_backward_f32(const float* x, const float* w1, const float* w2, const float* stats, const float* d_y, float* scratch,
                             float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2, int C, int R, int64_t HW, void* stream);
/* F.interpolate(in[1,1,h,w], size=(Ho,Wo), mode='bilinear', align_corners=False) evaluated at the n output pixels idx[]
 * (row-major index into Ho x Wo; NULL = all pixels in order, n = Ho*Wo), optionally followed by sigmoid:
 * the network's final upsample + sigmoid (lightweight_seg.py:366-367) and the full-resolution mask read at the batch's
 * pixels, interpolate(...)[rgb_idx] (train_mask_grid_sample.py:172-175).  The backward zeroes d_in[h,w] first. */
int crnerf_bilinear_gather_f32(const float* in, int h, int w, int Ho, int Wo, const int64_t* idx, int64_t n, int sigmoid, float* out,
                               void* stream);
int crnerf_bilinear_gather_backward_f32(const float* out, const float* d_out, int h, int w, int Ho, int Wo, const int64_t* idx, int64_t n,
                                        int sigmoid, float* d_in, void* stream);

/* The whole Context_Guided_Network(classes=1, M=2, N=2, input_channel=cin) of a TRAINING step as two calls (lightweight_seg.py:274-368,
 * train_mask_grid_sample.py:170-176): the operators above enqueued back to back, channel concatenations written in place, gradient sums
 * accumulated by the producing kernels.  image[cin,H,W] -> mask[H,W] = sigmoid(upsample(classifier(...))).  BatchNorm runs on batch
 * statistics and updates running_mean / running_var / num_batches_tracked (momentum, eps shared by the 14 layers, as the reference builds
 * them).  params[crnerf_cgnet_param_count()] and bn buffers [crnerf_cgnet_bn_count()] in the order csrc/cgnet_chain.hip documents (the
 * reference's state_dict order).  saved / scratch: crnerf_cgnet_arena_bytes(cin, H, W) each; `saved` and `mask` are read by the backward.
 * The backward OVERWRITES grads[i] (same shapes as params[i]); the image receives no gradient (it is data). */
int crnerf_cgnet_param_count(void);
int crnerf_cgnet_bn_count(void);
size_t crnerf_cgnet_arena_bytes(int32_t cin, int32_t H, int32_t W);
int crnerf_cgnet_forward_train_f32(const float* image, int32_t cin, int32_t H, int32_t W, const float* const* params, float* const* running_mean,
                                   float* const* running_var, int64_t* const* num_batches_tracked, float momentum, float eps, void* saved,
                                   float* mask, void* stream);
int crnerf_cgnet_backward_f32(const float* image, int32_t cin, int32_t H, int32_t W, const float* const* params, const void* saved,
                              const float* mask, const float* d_mask, void* scratch, float* const* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRNERF_H */
