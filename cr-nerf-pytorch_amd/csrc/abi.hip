// extern "C" surface of libcrnerf_hip.so -- see include/crnerf.h for the contract of every symbol.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/crnerf.h"
#include "crossray.h"
#include "kernels.h"
#include "layout.h"

namespace crnerf {

static thread_local char g_err[512] = "";

int set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return CRNERF_ERR_HIP;
}

int num_cus() {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// CU-share streams (include/crnerf.h).  The CU mask of hipExtStreamCreateWithCUMask is indexed in the driver's logical CU order, which on a
// multi-XCD part interleaves the XCDs (bit b = CU b / n_xcd of XCD b % n_xcd): a contiguous range of logical CUs of EVERY XCD is then the
// bit range [first * n_xcd, (first + count) * n_xcd).  n_xcd = 8 on gfx950 (256 CUs = 8 x 32).
static const int N_XCD = 8;
extern "C" int crnerf_cus_per_xcd(void) { return num_cus() / N_XCD; }
extern "C" int crnerf_stream_create_cu_share(void** stream, int first, int count) {
  if (!stream) return set_error(-1, "stream_create_cu_share: stream is NULL");
  const int per = num_cus() / N_XCD;
  if (first < 0 || count <= 0 || first + count > per) return set_error(-3, "stream_create_cu_share: [first, first + count) must lie inside crnerf_cus_per_xcd()");
  uint32_t mask[16] = {0};
  for (int b = first * N_XCD; b < (first + count) * N_XCD; ++b) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((num_cus() + 31) / 32), mask);
  if (e != hipSuccess) {
    static thread_local char msg[160];
    snprintf(msg, sizeof msg, "stream_create_cu_share: hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return set_error(-1, msg);
  }
  *stream = (void*)st;
  return 0;
}
extern "C" int crnerf_stream_destroy(void* stream) {
  if (!stream) return 0;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : set_error(-1, "stream_destroy: hipStreamDestroy failed");
}

unsigned int* sched_slot(const void* symbol) {
  static std::mutex mu;
  static int cursor = 0;
  void* base = nullptr;
  if (hipGetSymbolAddress(&base, symbol) != hipSuccess || !base) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  const int k = cursor++ % SCHED_SLOTS;
  return (unsigned int*)base + 2 * k;
}

int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what) {
  struct Slot { const void* fn; int dev; size_t bytes; };
  static Slot slots[128];
  static int n_slots = 0;
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  Slot* s = nullptr;
  for (int i = 0; i < n_slots; ++i)
    if (slots[i].fn == fn && slots[i].dev == dev) { s = &slots[i]; break; }
  if (s && s->bytes >= bytes) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    char msg[160];
    snprintf(msg, sizeof(msg), "hipFuncSetAttribute(%s, %zu bytes of LDS) failed", what, bytes);
    return set_error(-10, msg);
  }
  if (!s && n_slots < 128) s = &slots[n_slots++];
  if (s) { s->fn = fn; s->dev = dev; s->bytes = bytes; }
  return 0;
}

}  // namespace crnerf

using namespace crnerf;

#define REQUIRE(p, name) \
  if (!(p)) return set_error(CRNERF_ERR_NULL, name " is NULL")

extern "C" {

int crnerf_abi_version(void) { return CRNERF_ABI_VERSION; }
const char* crnerf_last_error(void) { return g_err; }
size_t crnerf_packed_mlp_bytes(void) { return PACKED_BYTES; }
size_t crnerf_crossray_workspace_bytes(void) { return CROSSRAY_WORKSPACE_BYTES; }

int crnerf_pack_mlp_weights(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors");
  REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights: a tensor pointer is NULL");
  MlpTensors t;
  for (int i = 0; i < 8; ++i) { t.w[i] = tensors[2 * i]; t.b[i] = tensors[2 * i + 1]; }
  t.w_final = tensors[16]; t.b_final = tensors[17];
  t.w_sigma = tensors[18]; t.b_sigma = tensors[19];
  t.w_dir = tensors[20]; t.b_dir = tensors[21];
  t.w_rgb = tensors[22]; t.b_rgb = tensors[23];
  return launch_pack_mlp(t, packed, (hipStream_t)stream);
}

static MlpTensors to_tensors(const float* const* tensors) {
  MlpTensors t;
  for (int i = 0; i < 8; ++i) { t.w[i] = tensors[2 * i]; t.b[i] = tensors[2 * i + 1]; }
  t.w_final = tensors[16]; t.b_final = tensors[17];
  t.w_sigma = tensors[18]; t.b_sigma = tensors[19];
  t.w_dir = tensors[20]; t.b_dir = tensors[21];
  t.w_rgb = tensors[22]; t.b_rgb = tensors[23];
  return t;
}

size_t crnerf_packed_mlp_t_bytes(void) { return PACKEDT_BYTES; }
size_t crnerf_mlp_train_acts_bytes(int64_t n) { return mlp_train_acts_bytes((long)n); }
size_t crnerf_mlp_train_scratch_bytes(int64_t n) { return mlp_train_scratch_bytes((long)n); }

int crnerf_pack_mlp_weights_t(const float* const* tensors, void* packed_t, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed_t, "packed_t");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_t: a tensor pointer is NULL");
  return launch_pack_mlpT(to_tensors(tensors), packed_t, (hipStream_t)stream);
}

int crnerf_mlp_forward_train_f32(const void* packed, const float* x, float* out, void* acts, int64_t n, void* stream) {
  if (n == 0) return 0;
  REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out"); REQUIRE(acts, "acts");
  return launch_mlp_forward_train(packed, x, out, (float*)acts, (long)n, (hipStream_t)stream);
}

// Flag and pointer checks shared by the three crnerf_mlp_backward_*_f32 entries.  `modes`: the weight-gradient modes this entry accepts (exclusive);
// CRNERF_BWD_PHASE_DGRAD / _WGRAD choose which half runs and with it which pointers are read.
static int check_backward_args(const char* who, int flags, int modes, const void* packed_t, const float* x, const float* out, const float* d_out,
                               const void* acts, const void* scratch, float* const* grads) {
  static thread_local char msg[160];
  const int phases = CRNERF_BWD_PHASE_DGRAD | CRNERF_BWD_PHASE_WGRAD;
  if (flags & ~(modes | phases)) { snprintf(msg, sizeof msg, "%s: unknown flag bits", who); return set_error(CRNERF_ERR_CONFIG, msg); }
  const int m = flags & modes;
  if ((m & (m - 1)) != 0) { snprintf(msg, sizeof msg, "%s: the weight-gradient modes are exclusive", who); return set_error(CRNERF_ERR_CONFIG, msg); }
  const bool dgrad = (flags & phases) != CRNERF_BWD_PHASE_WGRAD, wgrad = (flags & phases) != CRNERF_BWD_PHASE_DGRAD;
  const char* missing = !acts ? "acts" : !scratch ? "scratch" : (dgrad && !packed_t) ? "packed_t" : (dgrad && !out) ? "out" : (dgrad && !d_out) ? "d_out" :
                        (wgrad && !x) ? "x" : (wgrad && !grads) ? "grads" : nullptr;
  if (missing) { snprintf(msg, sizeof msg, "%s: %s is NULL", who, missing); return set_error(CRNERF_ERR_NULL, msg); }
  if (wgrad)
    for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
      if (!grads[i]) { snprintf(msg, sizeof msg, "%s: a gradient pointer is NULL", who); return set_error(CRNERF_ERR_NULL, msg); }
  return 0;
}

int crnerf_mlp_backward_f32(const void* packed_t, const float* x, const float* out, const float* d_out, const void* acts, void* scratch,
                            float* const* grads, int64_t n, void* stream) {
  return crnerf_mlp_backward_ex_f32(packed_t, x, out, d_out, acts, scratch, grads, n, 0, stream);
}

int crnerf_mlp_backward_ex_f32(const void* packed_t, const float* x, const float* out, const float* d_out, const void* acts, void* scratch,
                               float* const* grads, int64_t n, int flags, void* stream) {
  if (n == 0) return 0;
  if (int rc = check_backward_args("mlp_backward_ex", flags, CRNERF_BWD_WGRAD_BF16 | CRNERF_BWD_WGRAD_BF16X3, packed_t, x, out, d_out, acts, scratch, grads)) return rc;
  return launch_mlp_backward(packed_t, x, out, d_out, (const float*)acts, scratch, grads, (long)n, (hipStream_t)stream, flags);
}

int crnerf_posenc_f32(const float* x, float* out, int64_t n, int n_freqs, void* stream) {
  if (n == 0) return 0;
  REQUIRE(x, "x");
  REQUIRE(out, "out");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "posenc: negative n");
  return launch_posenc(x, out, (long)n, n_freqs, (hipStream_t)stream);
}

int crnerf_embed_points_f32(const float* rays, const float* z, const float* dir_emb, float* x, int64_t n_rays, int32_t n_samples, void* stream) {
  if (n_rays == 0 || n_samples == 0) return 0;
  REQUIRE(rays, "rays"); REQUIRE(z, "z"); REQUIRE(dir_emb, "dir_emb"); REQUIRE(x, "x");
  if (n_rays < 0 || n_samples < 0) return set_error(CRNERF_ERR_SHAPE, "embed_points: negative size");
  return launch_embed_points(rays, z, dir_emb, x, (long)n_rays, (int)n_samples, (hipStream_t)stream);
}

int crnerf_mlp_forward_f32(const void* packed, const float* x, float* out, int64_t n, int sigma_only, void* stream) {
  if (n == 0) return 0;
  REQUIRE(packed, "packed");
  REQUIRE(x, "x");
  REQUIRE(out, "out");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "mlp_forward: negative n");
  return launch_mlp_forward16(packed, x, out, (long)n, sigma_only, (hipStream_t)stream);
}

int crnerf_composite_f32(const float* raw, const float* z, const float* noise, float noise_std, float* weights, float* feature,
                         float* depth, int64_t R, int N, void* stream) {
  if (R == 0) return 0;
  REQUIRE(raw, "raw"); REQUIRE(z, "z"); REQUIRE(weights, "weights"); REQUIRE(feature, "feature"); REQUIRE(depth, "depth");
  if (R < 0) return set_error(CRNERF_ERR_SHAPE, "composite: negative R");
  return launch_composite(raw, z, noise, noise_std, weights, feature, depth, (long)R, N, (hipStream_t)stream);
}

int crnerf_composite_backward_f32(const float* raw, const float* z, const float* noise, float noise_std, const float* d_feature,
                                  const float* d_depth, const float* d_weights, float* d_raw, int64_t R, int N, void* stream) {
  if (R == 0) return 0;
  REQUIRE(raw, "raw"); REQUIRE(z, "z"); REQUIRE(d_feature, "d_feature"); REQUIRE(d_raw, "d_raw");
  if (R < 0) return set_error(CRNERF_ERR_SHAPE, "composite_backward: negative R");
  return launch_composite_backward(raw, z, noise, noise_std, d_feature, d_depth, d_weights, d_raw, (long)R, N, (hipStream_t)stream);
}

int crnerf_sample_pdf_merge_f32(const float* z_coarse, const float* weights_coarse, const float* u, int64_t u_stride, float* z_sorted,
                                float* z_samples, int64_t R, int Nc, int Ni, void* stream) {
  if (R == 0) return 0;
  REQUIRE(z_coarse, "z_coarse"); REQUIRE(weights_coarse, "weights_coarse"); REQUIRE(z_sorted, "z_sorted");
  if (R < 0) return set_error(CRNERF_ERR_SHAPE, "sample_pdf_merge: negative R");
  return launch_sample_pdf_merge(z_coarse, weights_coarse, u, (long)u_stride, z_sorted, z_samples, (long)R, Nc, Ni, (hipStream_t)stream);
}

// x3: 0 = no, 1 = the x3 core (three bf16 pieces), 2 = the h2 core (two fp16 pieces), 3 = the x3 core repairing NaN ray quads
static int render_rays_common(const crnerf_render_args* a, void* stream, bool bf16, void* acts_c = nullptr, void* acts_f = nullptr,
                              float* raw_c = nullptr, float* raw_f = nullptr, int x3 = 0) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  if (a->n_rays < 0) return set_error(CRNERF_ERR_SHAPE, "render_rays: negative n_rays");
  REQUIRE(a->packed_coarse, "packed_coarse"); REQUIRE(a->rays, "rays");
  REQUIRE(a->weights_coarse, "weights_coarse"); REQUIRE(a->feature_coarse, "feature_coarse"); REQUIRE(a->depth_coarse, "depth_coarse");
  if (a->n_importance > 0) {
    REQUIRE(a->packed_fine, "packed_fine");
    REQUIRE(a->weights_fine, "weights_fine"); REQUIRE(a->feature_fine, "feature_fine"); REQUIRE(a->depth_fine, "depth_fine");
  }
  RenderArgs r;
  r.packed_coarse = a->packed_coarse; r.packed_fine = a->packed_fine; r.rays = a->rays; r.view_dir = a->view_dir;
  r.z_coarse = a->z_coarse; r.z_steps = a->z_steps; r.u = a->u; r.u_stride = (long)a->u_stride; r.noise_coarse = a->noise_coarse; r.noise_fine = a->noise_fine;
  r.noise_std = a->noise_std; r.use_disp = a->use_disp; r.R = (long)a->n_rays; r.Nc = a->n_samples; r.Ni = a->n_importance;
  r.weights_coarse = a->weights_coarse; r.feature_coarse = a->feature_coarse; r.depth_coarse = a->depth_coarse;
  r.weights_fine = a->weights_fine; r.feature_fine = a->feature_fine; r.depth_fine = a->depth_fine; r.z_fine = a->z_fine;
  r.train_acts_coarse = acts_c; r.train_acts_fine = acts_f; r.train_raw_coarse = raw_c; r.train_raw_fine = raw_f;
  if (a->rng_flags) {
    if (a->rng_flags & ~(CRNERF_RNG_JITTER | CRNERF_RNG_U | CRNERF_RNG_NOISE)) return set_error(CRNERF_ERR_CONFIG, "render_rays: unknown rng_flags bits");
    if (bf16) return set_error(CRNERF_ERR_CONFIG, "render_rays: in-kernel random draws exist in the fp32, f32x3 and f32h2 kernels only");
    if ((a->rng_flags & CRNERF_RNG_JITTER) && a->z_coarse) return set_error(CRNERF_ERR_CONFIG, "render_rays: CRNERF_RNG_JITTER and z_coarse are exclusive");
    if ((a->rng_flags & CRNERF_RNG_U) && a->u) return set_error(CRNERF_ERR_CONFIG, "render_rays: CRNERF_RNG_U and u are exclusive");
    if ((a->rng_flags & CRNERF_RNG_NOISE) && (a->noise_coarse || a->noise_fine)) return set_error(CRNERF_ERR_CONFIG, "render_rays: CRNERF_RNG_NOISE and noise_* are exclusive");
    r.rng_seed = a->rng_seed; r.rng_ray_offset = (long)a->rng_ray_offset; r.rng_flags = a->rng_flags; r.perturb = a->perturb;
  }
  if ((a->z_coarse_out || a->noise_coarse_out || a->noise_fine_out) && bf16)   // (the bf16 kernels do not write them)
    return set_error(CRNERF_ERR_CONFIG, "render_rays: z_coarse_out / noise_*_out are written by the fp32, f32x3 and f32h2 kernels only");
  r.z_coarse_out = a->z_coarse_out; r.noise_coarse_out = a->noise_coarse_out; r.noise_fine_out = a->noise_fine_out;
  if (x3 == 2) return launch_render_rays_h2(r, (hipStream_t)stream);
  if (x3 == 3) r.repair = 1;
  if (x3) return launch_render_rays_x3(r, (hipStream_t)stream);
  if (bf16) return launch_render_rays_bf16p(r, (hipStream_t)stream);   // inference and (acts_c) the mixed-precision training twin
  return launch_render_rays16(r, (hipStream_t)stream);                 // inference and (acts_c) the fp32 training twin
}

int crnerf_render_rays_f32(const crnerf_render_args* a, void* stream) { return render_rays_common(a, stream, false); }
int crnerf_rng_fill_f32(float* out, int64_t n_rays, int n, uint64_t seed, int stream_id, int64_t ray_offset, void* stream) {
  if (n_rays == 0 || n == 0) return 0;
  REQUIRE(out, "out");
  if (n_rays < 0 || n < 0) return set_error(CRNERF_ERR_SHAPE, "rng_fill: negative size");
  if (stream_id < 0 || stream_id > 3) return set_error(CRNERF_ERR_CONFIG, "rng_fill: stream must be 0..3");
  return launch_rng_fill(out, (long)n_rays, n, seed, stream_id, (long)ray_offset, (hipStream_t)stream);
}

int crnerf_render_rays_bf16(const crnerf_render_args* a, void* stream) { return render_rays_common(a, stream, true); }

int crnerf_render_rays_bf16_fine(const crnerf_render_args* a, void* stream) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  if (a->n_rays < 0) return set_error(CRNERF_ERR_SHAPE, "render_rays_bf16_fine: negative n_rays");
  if (a->n_importance <= 0) return set_error(CRNERF_ERR_CONFIG, "render_rays_bf16_fine: n_importance must be > 0");
  REQUIRE(a->packed_fine, "packed_fine"); REQUIRE(a->rays, "rays"); REQUIRE(a->weights_coarse, "weights_coarse (input)");
  REQUIRE(a->weights_fine, "weights_fine"); REQUIRE(a->feature_fine, "feature_fine"); REQUIRE(a->depth_fine, "depth_fine");
  if (a->rng_flags || a->z_coarse_out || a->noise_coarse_out || a->noise_fine_out)
    return set_error(CRNERF_ERR_CONFIG, "render_rays_bf16_fine: no in-kernel random draws in the bf16 kernels");
  RenderArgs r;
  r.packed_coarse = a->packed_fine; r.packed_fine = a->packed_fine; r.rays = a->rays; r.view_dir = a->view_dir;
  r.z_coarse = a->z_coarse; r.z_steps = a->z_steps; r.u = a->u; r.u_stride = (long)a->u_stride; r.noise_coarse = nullptr; r.noise_fine = a->noise_fine;
  r.noise_std = a->noise_std; r.use_disp = a->use_disp; r.R = (long)a->n_rays; r.Nc = a->n_samples; r.Ni = a->n_importance;
  r.weights_coarse = a->weights_coarse; r.feature_coarse = nullptr; r.depth_coarse = nullptr;
  r.weights_fine = a->weights_fine; r.feature_fine = a->feature_fine; r.depth_fine = a->depth_fine; r.z_fine = a->z_fine;
  r.fine_only = 1;
  return launch_render_rays_bf16p(r, (hipStream_t)stream);
}

size_t crnerf_packed_mlp_mixed_bytes(void) { return gemm_packed_bytes(); }
size_t crnerf_mlp_train_mixed_acts_bytes(int64_t n) { return mlp_train_mixed_acts_bytes((long)n); }
size_t crnerf_mlp_train_mixed_scratch_bytes(int64_t n) { return mlp_train_mixed_scratch_bytes((long)n); }

int crnerf_pack_mlp_weights_mixed(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_mixed: a tensor pointer is NULL");
  return launch_pack_mlp_gemm(to_tensors(tensors), packed, (hipStream_t)stream);
}

int crnerf_mlp_forward_train_mixed_f32(const float* const* tensors, const void* packed, const float* x, float* out, void* acts, int64_t n,
                                       void* stream) {
  if (n == 0) return 0;
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out"); REQUIRE(acts, "acts");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "mlp_forward_train_mixed: negative n");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "mlp_forward_train_mixed: a tensor pointer is NULL");
  return launch_mlp_forward_train_mixed(to_tensors(tensors), packed, x, out, acts, (long)n, (hipStream_t)stream);
}

int crnerf_mlp_backward_mixed_f32(const float* const* tensors, const void* packed, const float* x, const float* out, const float* d_out,
                                  const void* acts, void* scratch, float* const* grads, int64_t n, void* stream) {
  if (n == 0) return 0;
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out"); REQUIRE(d_out, "d_out"); REQUIRE(acts, "acts");
  REQUIRE(scratch, "scratch"); REQUIRE(grads, "grads");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i] || !grads[i]) return set_error(CRNERF_ERR_NULL, "mlp_backward_mixed: a tensor / gradient pointer is NULL");
  return launch_mlp_backward_mixed(to_tensors(tensors), packed, x, out, d_out, acts, scratch, grads, (long)n, (hipStream_t)stream);
}

int crnerf_mlp_backward_mixed_ex_f32(const float* const* tensors, const void* packed, const float* out, const float* d_out, const void* acts,
                                     void* scratch, float* const* grads, int64_t n, int acts_layout, void* stream) {
  if (n == 0) return 0;
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed"); REQUIRE(out, "out"); REQUIRE(d_out, "d_out"); REQUIRE(acts, "acts");
  REQUIRE(scratch, "scratch"); REQUIRE(grads, "grads");
  if (acts_layout != CRNERF_MIXED_ACTS_GEMM && acts_layout != CRNERF_MIXED_ACTS_FUSED) return set_error(CRNERF_ERR_CONFIG, "mlp_backward_mixed_ex: unknown acts_layout");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i] || !grads[i]) return set_error(CRNERF_ERR_NULL, "mlp_backward_mixed_ex: a tensor / gradient pointer is NULL");
  return launch_mlp_backward_mixed(to_tensors(tensors), packed, nullptr, out, d_out, acts, scratch, grads, (long)n, (hipStream_t)stream, acts_layout);
}

int crnerf_render_rays_train_bf16(const crnerf_render_args* a, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine,
                                  void* stream) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  REQUIRE(acts_coarse, "acts_coarse"); REQUIRE(raw_coarse, "raw_coarse");
  if (a->n_importance > 0) { REQUIRE(acts_fine, "acts_fine"); REQUIRE(raw_fine, "raw_fine"); REQUIRE(a->z_fine, "z_fine"); }
  return render_rays_common(a, stream, true, acts_coarse, acts_fine, raw_coarse, raw_fine);
}

int crnerf_render_rays_train_f32(const crnerf_render_args* a, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine,
                                 void* stream) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  REQUIRE(acts_coarse, "acts_coarse"); REQUIRE(raw_coarse, "raw_coarse");
  if (a->n_importance > 0) { REQUIRE(acts_fine, "acts_fine"); REQUIRE(raw_fine, "raw_fine"); REQUIRE(a->z_fine, "z_fine"); }
  return render_rays_common(a, stream, false, acts_coarse, acts_fine, raw_coarse, raw_fine);
}

size_t crnerf_packed_mlp_t_x3_bytes(void) { return PACKEDXT_BYTES; }

int crnerf_pack_mlp_weights_t_x3(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_t_x3: a tensor pointer is NULL");
  return launch_pack_mlp_x3t(to_tensors(tensors), packed, (hipStream_t)stream);
}

int crnerf_mlp_backward_x3_f32(const void* packed_t_x3, const float* x, const float* out, const float* d_out, const void* acts, void* scratch,
                               float* const* grads, int64_t n, int flags, void* stream) {
  if (n == 0) return 0;
  if (int rc = check_backward_args("mlp_backward_x3", flags, CRNERF_BWD_WGRAD_BF16 | CRNERF_BWD_WGRAD_BF16X3, packed_t_x3, x, out, d_out, acts, scratch, grads)) return rc;
  return launch_mlp_backward(nullptr, x, out, d_out, (const float*)acts, scratch, grads, (long)n, (hipStream_t)stream, flags, packed_t_x3);
}

size_t crnerf_packed_mlp_t_h2_bytes(void) { return PACKEDHT_BYTES; }

int crnerf_pack_mlp_weights_t_h2(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_t_h2: a tensor pointer is NULL");
  return launch_pack_mlp_h2t(to_tensors(tensors), packed, (hipStream_t)stream);
}

int crnerf_mlp_backward_h2_f32(const void* packed_t_h2, const void* packed_t_x3, const float* x, const float* out, const float* d_out, const void* acts,
                               void* scratch, float* const* grads, int64_t n, int flags, void* stream) {
  if (n == 0) return 0;
  if (int rc = check_backward_args("mlp_backward_h2", flags, CRNERF_BWD_WGRAD_BF16 | CRNERF_BWD_WGRAD_BF16X3 | CRNERF_BWD_WGRAD_F16X2, packed_t_h2, x, out, d_out, acts,
                                   scratch, grads)) return rc;
  // a PHASE_WGRAD call carries no packs; the non-null marker keeps the f16x2 mode (its range words are in the scratch)
  const void* h2 = packed_t_h2 ? packed_t_h2 : (const void*)scratch;
  return launch_mlp_backward(nullptr, x, out, d_out, (const float*)acts, scratch, grads, (long)n, (hipStream_t)stream, flags, packed_t_x3, h2);
}

size_t crnerf_packed_mlp_x3_bytes(void) { return PACKEDX_BYTES; }

int crnerf_pack_mlp_weights_x3(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_x3: a tensor pointer is NULL");
  return launch_pack_mlp_x3(to_tensors(tensors), packed, (hipStream_t)stream);
}

int crnerf_mlp_forward_f32x3(const void* packed, const float* x, float* out, int64_t n, int sigma_only, void* stream) {
  if (n == 0) return 0;
  REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "mlp_forward_f32x3: negative n");
  return launch_mlp_forward_x3(packed, x, out, (long)n, sigma_only, (hipStream_t)stream, 0);
}

int crnerf_render_rays_f32x3(const crnerf_render_args* a, void* stream) { return render_rays_common(a, stream, false, nullptr, nullptr, nullptr, nullptr, 1); }

size_t crnerf_packed_mlp_h2_bytes(void) { return PACKEDH_BYTES; }

int crnerf_pack_mlp_weights_h2(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_h2: a tensor pointer is NULL");
  return launch_pack_mlp_h2(to_tensors(tensors), packed, (hipStream_t)stream);
}

int crnerf_pack_mlp_weights_h2_async(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_h2_async: a tensor pointer is NULL");
  return launch_pack_mlp_h2(to_tensors(tensors), packed, (hipStream_t)stream, false);
}

int crnerf_pack_h2_status(const void* packed_h2, void* stream) {
  REQUIRE(packed_h2, "packed_h2");
  return pack_h2_status(packed_h2, (hipStream_t)stream);
}

int crnerf_mlp_forward_f32h2(const void* packed, const float* x, float* out, int64_t n, int sigma_only, void* stream) {
  if (n == 0) return 0;
  REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "mlp_forward_f32h2: negative n");
  return launch_mlp_forward_h2(packed, x, out, (long)n, sigma_only, (hipStream_t)stream, 0);
}

int crnerf_render_rays_f32x3_repair(const crnerf_render_args* a, void* stream) { return render_rays_common(a, stream, false, nullptr, nullptr, nullptr, nullptr, 3); }
int crnerf_mlp_forward_f32x3_repair(const void* packed, const float* x, float* out, int64_t n, int sigma_only, void* stream) {
  if (n == 0) return 0;
  REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "mlp_forward_f32x3_repair: negative n");
  return launch_mlp_forward_x3(packed, x, out, (long)n, sigma_only, (hipStream_t)stream, 1);
}
int crnerf_render_rays_f32h2(const crnerf_render_args* a, void* stream) { return render_rays_common(a, stream, false, nullptr, nullptr, nullptr, nullptr, 2); }

int crnerf_render_rays_train_f32x3(const crnerf_render_args* a, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine, void* stream) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  REQUIRE(acts_coarse, "acts_coarse"); REQUIRE(raw_coarse, "raw_coarse");
  if (a->n_importance > 0) { REQUIRE(acts_fine, "acts_fine"); REQUIRE(raw_fine, "raw_fine"); REQUIRE(a->z_fine, "z_fine"); }
  return render_rays_common(a, stream, false, acts_coarse, acts_fine, raw_coarse, raw_fine, true);
}

int crnerf_render_rays_train_f32h2(const crnerf_render_args* a, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine, void* stream) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  REQUIRE(acts_coarse, "acts_coarse"); REQUIRE(raw_coarse, "raw_coarse");
  if (a->n_importance > 0) { REQUIRE(acts_fine, "acts_fine"); REQUIRE(raw_fine, "raw_fine"); REQUIRE(a->z_fine, "z_fine"); }
  return render_rays_common(a, stream, false, acts_coarse, acts_fine, raw_coarse, raw_fine, 2);
}

int crnerf_render_rays_train_f32x3_repair(const crnerf_render_args* a, void* acts_coarse, void* acts_fine, float* raw_coarse, float* raw_fine, void* stream) {
  REQUIRE(a, "args");
  if (a->n_rays == 0) return 0;
  REQUIRE(acts_coarse, "acts_coarse"); REQUIRE(raw_coarse, "raw_coarse");
  if (a->n_importance > 0) { REQUIRE(acts_fine, "acts_fine"); REQUIRE(raw_fine, "raw_fine"); REQUIRE(a->z_fine, "z_fine"); }
  return render_rays_common(a, stream, false, acts_coarse, acts_fine, raw_coarse, raw_fine, 3);
}

size_t crnerf_packed_mlp_bf16_bytes(void) { return PACKEDB_BYTES; }

int crnerf_pack_mlp_weights_bf16(const float* const* tensors, void* packed, void* stream) {
  REQUIRE(tensors, "tensors"); REQUIRE(packed, "packed");
  for (int i = 0; i < CRNERF_MLP_TENSORS; ++i)
    if (!tensors[i]) return set_error(CRNERF_ERR_NULL, "pack_mlp_weights_bf16: a tensor pointer is NULL");
  return launch_pack_mlp_bf16(to_tensors(tensors), packed, (hipStream_t)stream);
}

int crnerf_mlp_forward_bf16(const void* packed, const float* x, float* out, int64_t n, int sigma_only, void* stream) {
  if (n == 0) return 0;
  REQUIRE(packed, "packed"); REQUIRE(x, "x"); REQUIRE(out, "out");
  if (n < 0) return set_error(CRNERF_ERR_SHAPE, "mlp_forward_bf16: negative n");
  return launch_mlp_forward_bf16p(packed, x, out, (long)n, sigma_only, (hipStream_t)stream);
}

size_t crnerf_encoder_workspace_bytes(int H, int W) { return encoder_workspace_bytes(H, W); }

int crnerf_encoder_forward_f32(const float* image, int H, int W, const float* const* weights, void* workspace, float* out, void* stream) {
  REQUIRE(image, "image"); REQUIRE(weights, "weights"); REQUIRE(workspace, "workspace"); REQUIRE(out, "out");
  for (int i = 0; i < CRNERF_ENCODER_TENSORS; ++i)
    if (!weights[i]) return set_error(CRNERF_ERR_NULL, "encoder_forward: a weight pointer is NULL");
  return launch_encoder_forward(image, H, W, weights, workspace, out, (hipStream_t)stream);
}

int crnerf_ray_directions_f32(int H, int W, float fx, float fy, float cx, float cy, float* directions, void* stream) {
  REQUIRE(directions, "directions");
  return launch_ray_directions(fx, fy, cx, cy, H, W, directions, (hipStream_t)stream);
}

int crnerf_rays_from_directions_f32(const float* directions, const float* c2w_host, int64_t n, float* rays_o, float* rays_d, void* stream) {
  if (n == 0) return 0;
  REQUIRE(directions, "directions"); REQUIRE(c2w_host, "c2w"); REQUIRE(rays_o, "rays_o"); REQUIRE(rays_d, "rays_d");
  return launch_rays_from_directions(directions, c2w_host, (long)n, rays_o, rays_d, (hipStream_t)stream);
}

int crnerf_generate_rays_f32(const float* intrinsics_host, const float* c2w_host, int H, int W, float near, float far, float* rays,
                             void* stream) {
  REQUIRE(intrinsics_host, "intrinsics"); REQUIRE(c2w_host, "c2w"); REQUIRE(rays, "rays");
  return launch_generate_rays(intrinsics_host, c2w_host, H, W, near, far, rays, (hipStream_t)stream);
}

int crnerf_crossray_chansum_f32(const float* x, int64_t HW, float* sum64, void* workspace, void* stream) {
  REQUIRE(x, "x"); REQUIRE(sum64, "sum64"); REQUIRE(workspace, "workspace");
  return launch_crossray_chansum(x, (long)HW, sum64, (float*)workspace, (hipStream_t)stream);
}

int crnerf_crossray_gram_f32(const float* x, int64_t HW, const float* mean64, const float* const* cnn, float* gram_sum,
                             void* workspace, void* stream) {
  REQUIRE(x, "x"); REQUIRE(mean64, "mean64"); REQUIRE(cnn, "cnn"); REQUIRE(gram_sum, "gram_sum"); REQUIRE(workspace, "workspace");
  for (int i = 0; i < 6; ++i)
    if (!cnn[i]) return set_error(CRNERF_ERR_NULL, "crossray_gram: a cnn tensor pointer is NULL");
  CnnTensors w{cnn[0], cnn[1], cnn[2], cnn[3], cnn[4], cnn[5]};
  return launch_crossray_gram(x, (long)HW, mean64, w, gram_sum, (float*)workspace, (hipStream_t)stream);
}

int crnerf_crossray_matrix_f32(const float* gram_sum, double count, const float* fc_w, const float* fc_b, float* out, void* stream) {
  REQUIRE(gram_sum, "gram_sum"); REQUIRE(fc_w, "fc_w"); REQUIRE(fc_b, "fc_b"); REQUIRE(out, "out");
  return launch_crossray_matrix(gram_sum, count, fc_w, fc_b, out, (hipStream_t)stream);
}

int crnerf_crossray_fold_f32(const float* s_matrix, const float* c_matrix, const float* c_mean64, const float* s_mean64,
                             const float* const* lin, float* affine, void* stream) {
  REQUIRE(lin, "lin"); REQUIRE(affine, "affine");
  for (int i = 0; i < 6; ++i)
    if (!lin[i]) return set_error(CRNERF_ERR_NULL, "crossray_fold: a tensor pointer is NULL");
  if (s_matrix) { REQUIRE(c_matrix, "c_matrix"); REQUIRE(c_mean64, "c_mean64"); REQUIRE(s_mean64, "s_mean64"); }
  FoldTensors w{lin[0], lin[1], lin[2], lin[3], lin[4], lin[5]};
  return launch_crossray_fold(s_matrix, c_matrix, c_mean64, s_mean64, w, affine, (hipStream_t)stream);
}

int crnerf_crossray_decode_f32(const float* content, int64_t HW, const float* style, int64_t HWs, const float* const* w,
                               void* workspace, float* rgb, int64_t plane_stride, void* stream) {
  if (HW == 0) return 0;
  REQUIRE(content, "content"); REQUIRE(w, "weights"); REQUIRE(workspace, "workspace"); REQUIRE(rgb, "rgb");
  if (HW < 0 || HWs < 0) return set_error(CRNERF_ERR_SHAPE, "crossray_decode: negative size");
  for (int i = 0; i < CRNERF_DECODER_TENSORS; ++i)
    if (!w[i]) return set_error(CRNERF_ERR_NULL, "crossray_decode: a weight pointer is NULL");
  DecodeArgs d;
  d.content = content; d.HW = (long)HW; d.style = style; d.HWs = (long)HWs;
  d.snet = CnnTensors{w[0], w[1], w[2], w[3], w[4], w[5]}; d.snet_fc_w = w[6]; d.snet_fc_b = w[7];
  d.cnet = CnnTensors{w[8], w[9], w[10], w[11], w[12], w[13]}; d.cnet_fc_w = w[14]; d.cnet_fc_b = w[15];
  d.lin = FoldTensors{w[16], w[17], w[18], w[19], w[20], w[21]};
  d.workspace = workspace; d.rgb = rgb; d.plane_stride = (long)plane_stride;
  return launch_crossray_decode(d, (hipStream_t)stream);
}

int crnerf_crossray_decode_sharded_f32(const float* content, int64_t HW_local, const float* style, int64_t HWs, const float* const* w,
                                       int phase, float* xchg, double count_global, void* workspace, float* rgb, int64_t plane_stride,
                                       void* stream) {
  REQUIRE(style, "style"); REQUIRE(w, "weights"); REQUIRE(xchg, "xchg"); REQUIRE(workspace, "workspace");
  if (HW_local < 0 || phase < 0 || phase > 2) return set_error(CRNERF_ERR_SHAPE, "crossray_decode_sharded: bad size or phase");
  if (HW_local > 0) REQUIRE(content, "content");
  if (phase == 2 && HW_local > 0) REQUIRE(rgb, "rgb");
  for (int i = 0; i < CRNERF_DECODER_TENSORS; ++i)
    if (!w[i]) return set_error(CRNERF_ERR_NULL, "crossray_decode_sharded: a weight pointer is NULL");
  DecodeArgs d;
  d.content = content; d.HW = (long)HW_local; d.style = style; d.HWs = (long)HWs;
  d.snet = CnnTensors{w[0], w[1], w[2], w[3], w[4], w[5]}; d.snet_fc_w = w[6]; d.snet_fc_b = w[7];
  d.cnet = CnnTensors{w[8], w[9], w[10], w[11], w[12], w[13]}; d.cnet_fc_w = w[14]; d.cnet_fc_b = w[15];
  d.lin = FoldTensors{w[16], w[17], w[18], w[19], w[20], w[21]};
  d.workspace = workspace; d.rgb = rgb; d.plane_stride = (long)plane_stride;
  return launch_crossray_decode_sharded(d, phase, xchg, count_global, (hipStream_t)stream);
}

size_t crnerf_crossray_backward_workspace_bytes(int64_t HW, int64_t HWs) { return crossray_backward_workspace_floats((long)HW, (long)HWs) * sizeof(float); }

int crnerf_crossray_decode_backward_f32(const float* content, int64_t HW, const float* style, int64_t HWs, const float* const* w,
                                        const float* d_rgb, int64_t d_plane_stride, void* workspace, float* d_content, float* d_style,
                                        float* const* grads, void* stream) {
  REQUIRE(content, "content"); REQUIRE(style, "style"); REQUIRE(w, "weights"); REQUIRE(d_rgb, "d_rgb"); REQUIRE(workspace, "workspace");
  REQUIRE(d_content, "d_content"); REQUIRE(d_style, "d_style"); REQUIRE(grads, "grads");
  if (HW <= 0 || HWs <= 0) return set_error(CRNERF_ERR_SHAPE, "crossray_decode_backward: empty grid");
  for (int i = 0; i < CRNERF_DECODER_TENSORS; ++i)
    if (!w[i] || !grads[i]) return set_error(CRNERF_ERR_NULL, "crossray_decode_backward: a weight or gradient pointer is NULL");
  DecodeArgs d;
  d.content = content; d.HW = (long)HW; d.style = style; d.HWs = (long)HWs;
  d.snet = CnnTensors{w[0], w[1], w[2], w[3], w[4], w[5]}; d.snet_fc_w = w[6]; d.snet_fc_b = w[7];
  d.cnet = CnnTensors{w[8], w[9], w[10], w[11], w[12], w[13]}; d.cnet_fc_w = w[14]; d.cnet_fc_b = w[15];
  d.lin = FoldTensors{w[16], w[17], w[18], w[19], w[20], w[21]};
  d.workspace = nullptr; d.rgb = nullptr; d.plane_stride = 0;
  return launch_crossray_decode_backward(d, d_rgb, (long)d_plane_stride, (float*)workspace, d_content, d_style, grads, (hipStream_t)stream);
}

int crnerf_crossray_decode_backward_sharded_f32(const float* content, int64_t HW, const float* style, int64_t HWs, const float* const* w, const float* d_rgb,
                                                int64_t d_plane_stride, void* workspace, float* d_content, float* d_style, float* const* grads, int phase,
                                                const float* fwd_xchg, double count_global, float* xb, void* stream) {
  REQUIRE(content, "content"); REQUIRE(style, "style"); REQUIRE(w, "weights"); REQUIRE(d_rgb, "d_rgb"); REQUIRE(workspace, "workspace");
  REQUIRE(d_content, "d_content"); REQUIRE(d_style, "d_style"); REQUIRE(grads, "grads"); REQUIRE(fwd_xchg, "fwd_xchg"); REQUIRE(xb, "xb");
  if (HW <= 0 || HWs <= 0) return set_error(CRNERF_ERR_SHAPE, "crossray_decode_backward_sharded: empty grid (every rank must hold pixels)");
  if (phase < 0 || phase > 2) return set_error(CRNERF_ERR_CONFIG, "crossray_decode_backward_sharded: phase must be 0, 1 or 2");
  for (int i = 0; i < CRNERF_DECODER_TENSORS; ++i)
    if (!w[i] || !grads[i]) return set_error(CRNERF_ERR_NULL, "crossray_decode_backward_sharded: a weight or gradient pointer is NULL");
  DecodeArgs d;
  d.content = content; d.HW = (long)HW; d.style = style; d.HWs = (long)HWs;
  d.snet = CnnTensors{w[0], w[1], w[2], w[3], w[4], w[5]}; d.snet_fc_w = w[6]; d.snet_fc_b = w[7];
  d.cnet = CnnTensors{w[8], w[9], w[10], w[11], w[12], w[13]}; d.cnet_fc_w = w[14]; d.cnet_fc_b = w[15];
  d.lin = FoldTensors{w[16], w[17], w[18], w[19], w[20], w[21]};
  d.workspace = nullptr; d.rgb = nullptr; d.plane_stride = 0;
  return launch_crossray_decode_backward_sharded(d, d_rgb, (long)d_plane_stride, (float*)workspace, d_content, d_style, grads, phase, fwd_xchg, count_global, xb,
                                                 (hipStream_t)stream);
}

int crnerf_crossray_apply_f32(const float* x, int64_t HW, const float* affine, float* rgb, int64_t plane_stride, void* stream) {
  if (HW == 0) return 0;
  REQUIRE(x, "x"); REQUIRE(affine, "affine"); REQUIRE(rgb, "rgb");
  return launch_crossray_apply(x, (long)HW, affine, rgb, (long)plane_stride, (hipStream_t)stream);
}


size_t crnerf_decoder_content_backward_workspace_bytes(int64_t HW) { return content_backward_workspace_floats((long)HW) * sizeof(float); }

int crnerf_decoder_content_backward_f32(const float* content, int64_t HW, const float* rgb_w, const float* rgb, int64_t rgb_plane_stride,
                                        const float* d_rgb, int64_t d_plane_stride, void* workspace, float* d_content, float* d_w, float* d_b,
                                        void* stream) {
  if (HW == 0) return 0;
  REQUIRE(content, "content"); REQUIRE(rgb_w, "rgb_w"); REQUIRE(rgb, "rgb"); REQUIRE(d_rgb, "d_rgb"); REQUIRE(workspace, "workspace");
  REQUIRE(d_content, "d_content"); REQUIRE(d_w, "d_w"); REQUIRE(d_b, "d_b");
  if (HW < 0) return set_error(CRNERF_ERR_SHAPE, "decoder_content_backward: negative HW");
  return launch_content_backward(content, (long)HW, rgb_w, rgb, (long)rgb_plane_stride, d_rgb, (long)d_plane_stride, (float*)workspace, d_content,
                                 d_w, d_b, (hipStream_t)stream);
}

size_t crnerf_encoder_train_saved_bytes(int H, int W) { return encoder_train_saved_bytes(H, W); }
size_t crnerf_encoder_train_scratch_bytes(int H, int W) { return encoder_train_scratch_bytes(H, W); }

int crnerf_encoder_forward_train_f32(const float* image, int H, int W, const float* const* weights, void* saved, float* out, void* stream) {
  REQUIRE(image, "image"); REQUIRE(weights, "weights"); REQUIRE(saved, "saved"); REQUIRE(out, "out");
  for (int i = 0; i < CRNERF_ENCODER_TENSORS; ++i)
    if (!weights[i]) return set_error(CRNERF_ERR_NULL, "encoder_forward_train: a weight pointer is NULL");
  return launch_encoder_forward_train(image, H, W, weights, saved, out, (hipStream_t)stream);
}

size_t crnerf_encoder_train_band_saved_bytes(int H, int W, int n_out_rows) { return encoder_train_saved_bytes(H, W, n_out_rows * 32); }
size_t crnerf_encoder_train_band_scratch_bytes(int H, int W, int n_out_rows) { return encoder_train_scratch_bytes(H, W, n_out_rows * 32); }

int crnerf_encoder_forward_train_band_f32(const float* image_rows, int H, int W, int H_image, int row0, int o0, int o1, const float* const* weights, void* saved,
                                          float* out, void* stream) {
  REQUIRE(image_rows, "image_rows"); REQUIRE(weights, "weights"); REQUIRE(saved, "saved"); REQUIRE(out, "out");
  for (int i = 0; i < CRNERF_ENCODER_TENSORS; ++i)
    if (!weights[i]) return set_error(CRNERF_ERR_NULL, "encoder_forward_train_band: a weight pointer is NULL");
  return launch_encoder_forward_train_band(image_rows, H, W, H_image, row0, o0, o1, weights, saved, out, (hipStream_t)stream);
}

int crnerf_encoder_backward_band_f32(int H, int W, int H_image, int row0, int o0, int o1, const float* const* weights, const void* saved, const float* out,
                                     const float* d_out, void* scratch, float* const* grads, float* d_image_rows, void* stream) {
  REQUIRE(weights, "weights"); REQUIRE(saved, "saved"); REQUIRE(out, "out"); REQUIRE(d_out, "d_out"); REQUIRE(scratch, "scratch"); REQUIRE(grads, "grads");
  if (H < 8 || W < 8) return set_error(CRNERF_ERR_SHAPE, "encoder_backward_band: band must be at least 8x8");
  for (int i = 0; i < CRNERF_ENCODER_TENSORS; ++i)
    if (!weights[i] || !grads[i]) return set_error(CRNERF_ERR_NULL, "encoder_backward_band: a weight or gradient pointer is NULL");
  return launch_encoder_backward_band(H, W, H_image, row0, o0, o1, weights, saved, out, d_out, scratch, grads, d_image_rows, (hipStream_t)stream);
}

int crnerf_encoder_backward_f32(int H, int W, const float* const* weights, const void* saved, const float* out, const float* d_out, void* scratch,
                                float* const* grads, float* d_image, void* stream) {
  REQUIRE(weights, "weights"); REQUIRE(saved, "saved"); REQUIRE(out, "out"); REQUIRE(d_out, "d_out"); REQUIRE(scratch, "scratch"); REQUIRE(grads, "grads");
  if (H < 8 || W < 8) return set_error(CRNERF_ERR_SHAPE, "encoder_backward: image must be at least 8x8");
  for (int i = 0; i < CRNERF_ENCODER_TENSORS; ++i)
    if (!weights[i] || !grads[i]) return set_error(CRNERF_ERR_NULL, "encoder_backward: a weight or gradient pointer is NULL");
  return launch_encoder_backward(H, W, weights, saved, out, d_out, scratch, grads, d_image, (hipStream_t)stream);
}

static bool to_loss_args(const crnerf_loss_args* a, LossArgs& k, LossScales& sc) {
  k.rgb_c = a->rgb_coarse; k.rgb_f = a->rgb_fine; k.tgt = a->targets; k.mask = a->mask;
  k.a = a->a_embedded; k.a_rand = a->a_embedded_random; k.a_rec = a->a_embedded_random_rec; k.c_wo = a->content_wo; k.c_with = a->content_with;
  k.R = (long)a->n_rays; k.n_a = a->a_embedded ? (long)a->n_a : 0; k.n_rec = a->a_embedded_random_rec ? (long)a->n_rec : 0;
  k.n_c = a->content_wo ? (long)a->n_content : 0;
  k.rc_sr = (long)a->rgb_coarse_row_stride; k.rc_sc = (long)a->rgb_coarse_chan_stride;
  k.rf_sr = (long)a->rgb_fine_row_stride; k.rf_sc = (long)a->rgb_fine_chan_stride;
  k.tg_sr = (long)a->targets_row_stride; k.tg_sc = (long)a->targets_chan_stride;
  k.mse_a = a->mse_on_appearance;
  const double c = a->coef, R = (double)a->n_rays;
  sc.s[0] = k.n_a ? (float)(c * a->weight_kl / (double)k.n_a) : 0.0f;
  sc.s[1] = k.n_rec ? (float)(c * a->weight_rec_a / (double)k.n_rec) : 0.0f;
  sc.s[2] = (float)(c * 0.5 / (3.0 * R));
  sc.s[3] = k.n_c ? (float)(c * a->weight_content / (double)k.n_c) : 0.0f;
  const bool reg = a->mask && a->rgb_fine;                    // r_ms / r_md exist with rgb_fine AND out_mask only (losses.py:67-69)
  sc.s[4] = reg ? (float)(c * a->mask_size_weight / R) : 0.0f;
  sc.s[5] = reg ? (float)(c * a->mask_digit_weight / R) : 0.0f;
  sc.s[6] = a->rgb_fine ? (float)(c * 0.5 / (3.0 * R)) : 0.0f;
  return true;
}

size_t crnerf_loss_workspace_bytes(void) { return loss_workspace_bytes(); }

int crnerf_loss_f32(const crnerf_loss_args* a, float* losses, void* workspace, void* stream) {
  REQUIRE(a, "args"); REQUIRE(losses, "losses"); REQUIRE(workspace, "workspace");
  if (a->n_rays <= 0) return set_error(CRNERF_ERR_SHAPE, "loss: n_rays must be positive");
  REQUIRE(a->rgb_coarse, "rgb_coarse"); REQUIRE(a->targets, "targets");
  if (a->a_embedded_random_rec && !a->a_embedded_random) return set_error(CRNERF_ERR_NULL, "loss: a_embedded_random_rec without a_embedded_random");
  if ((a->content_wo == nullptr) != (a->content_with == nullptr)) return set_error(CRNERF_ERR_NULL, "loss: content_wo and content_with come as a pair");
  LossArgs k; LossScales sc;
  to_loss_args(a, k, sc);
  return launch_loss_forward(k, sc, losses, workspace, (hipStream_t)stream);
}

int crnerf_loss_backward_f32(const crnerf_loss_args* a, const float* upstream, const crnerf_loss_grads* g, void* stream) {
  REQUIRE(a, "args"); REQUIRE(upstream, "upstream"); REQUIRE(g, "grads");
  if (a->n_rays <= 0) return set_error(CRNERF_ERR_SHAPE, "loss_backward: n_rays must be positive");
  REQUIRE(a->rgb_coarse, "rgb_coarse"); REQUIRE(a->targets, "targets");
  LossArgs k; LossScales sc;
  to_loss_args(a, k, sc);
  LossGrads kg{g->d_rgb_coarse, g->d_rgb_fine, g->d_mask, g->d_a_embedded, g->d_a_embedded_random_rec, g->d_content_wo, g->d_content_with};
  return launch_loss_backward(k, sc, upstream, kg, (hipStream_t)stream);
}

int crnerf_grid_sample_batch_f32(const crnerf_batch_args* a, void* stream) {
  REQUIRE(a, "args");
  if (a->side <= 0) return 0;
  REQUIRE(a->all_rays, "all_rays"); REQUIRE(a->all_rgbs, "all_rgbs"); REQUIRE(a->w_lin, "w_lin"); REQUIRE(a->h_lin, "h_lin");
  REQUIRE(a->rays, "rays"); REQUIRE(a->ts, "ts"); REQUIRE(a->rgbs, "rgbs"); REQUIRE(a->rgb_idx, "rgb_idx"); REQUIRE(a->uv_sample, "uv_sample");
  if (a->img_w <= 0 || a->img_h <= 0 || a->ray_stride < 9) return set_error(CRNERF_ERR_SHAPE, "grid_sample_batch: bad image size or ray_stride < 9");
  BatchArgs k{a->all_rays, (long)a->ray_stride, a->all_rgbs, (long)a->row_offset, a->img_w, a->img_h, a->side, a->w_lin, a->h_lin,
              a->scale, a->h_offset, a->w_offset, a->rays, (long*)a->ts, a->rgbs, (long*)a->rgb_idx, a->uv_sample};
  return launch_grid_batch(k, (hipStream_t)stream);
}

int crnerf_adam_max_tensors(void) { return ADAM_MAX_TENSORS; }

int crnerf_adam_step_f32(float* params, float* exp_avg, float* exp_avg_sq, const int32_t* blocks, int32_t n_blocks, const float* const* grads,
                         int32_t n_tensors, float step_size, float beta1, float beta2, float eps, float weight_decay, float bias_correction2_sqrt,
                         void* stream) {
  if (n_blocks <= 0) return 0;
  REQUIRE(params, "params"); REQUIRE(exp_avg, "exp_avg"); REQUIRE(exp_avg_sq, "exp_avg_sq"); REQUIRE(blocks, "blocks"); REQUIRE(grads, "grads");
  if (n_tensors <= 0 || n_tensors > ADAM_MAX_TENSORS) return set_error(CRNERF_ERR_SHAPE, "adam_step: n_tensors outside 1 ... crnerf_adam_max_tensors()");
  if (!(bias_correction2_sqrt > 0.0f) || !(eps >= 0.0f)) return set_error(CRNERF_ERR_CONFIG, "adam_step: bias_correction2_sqrt must be positive and eps non-negative");
  const AdamHyper h{step_size, beta1, beta2, eps, weight_decay, bias_correction2_sqrt};
  return launch_adam_step(params, exp_avg, exp_avg_sq, blocks, n_blocks, grads, n_tensors, h, (hipStream_t)stream);
}

static int to_geom(const crnerf_conv_geom* a, ConvGeom& g) {
  if (a->cin <= 0 || a->cout <= 0 || a->H <= 0 || a->W <= 0 || a->k <= 0 || a->stride <= 0 || a->dil <= 0 || a->pad < 0)
    return set_error(CRNERF_ERR_SHAPE, "conv2d: non-positive geometry");
  if (a->depthwise && a->cin != a->cout) return set_error(CRNERF_ERR_SHAPE, "conv2d: depth-wise needs cin == cout");
  const int Ho = (a->H + 2 * a->pad - a->dil * (a->k - 1) - 1) / a->stride + 1, Wo = (a->W + 2 * a->pad - a->dil * (a->k - 1) - 1) / a->stride + 1;
  if (Ho <= 0 || Wo <= 0) return set_error(CRNERF_ERR_SHAPE, "conv2d: empty output");
  g = ConvGeom{a->cin, a->cout, a->H, a->W, Ho, Wo, a->k, a->stride, a->pad, a->dil, a->depthwise ? 1 : 0};
  return 0;
}

int crnerf_conv2d_f32(const crnerf_conv_geom* geom, const float* x, const float* w, float* y, void* stream) {
  REQUIRE(geom, "geom"); REQUIRE(x, "x"); REQUIRE(w, "w"); REQUIRE(y, "y");
  ConvGeom g;
  if (int e = to_geom(geom, g)) return e;
  return launch_cg_conv_forward(g, x, w, y, (hipStream_t)stream);
}

int crnerf_conv2d_backward_f32(const crnerf_conv_geom* geom, const float* x, const float* w, const float* d_y, float* d_x, float* d_w, void* stream) {
  REQUIRE(geom, "geom"); REQUIRE(x, "x"); REQUIRE(w, "w"); REQUIRE(d_y, "d_y"); REQUIRE(d_w, "d_w");
  ConvGeom g;
  if (int e = to_geom(geom, g)) return e;
  return launch_cg_conv_backward(g, x, w, d_y, d_x, d_w, (hipStream_t)stream);
}

int crnerf_bn_prelu_f32(const float* x, const float* gamma, const float* beta, const float* alpha, float* mean, float* invstd, float* var_unbiased,
                        float* y, int C, int64_t HW, float eps, int training, void* stream) {
  REQUIRE(x, "x"); REQUIRE(gamma, "gamma"); REQUIRE(beta, "beta"); REQUIRE(alpha, "alpha"); REQUIRE(mean, "mean"); REQUIRE(invstd, "invstd"); REQUIRE(y, "y");
  if (training) REQUIRE(var_unbiased, "var_unbiased");
  if (C <= 0 || HW <= 0 || HW > (1 << 30)) return set_error(CRNERF_ERR_SHAPE, "bn_prelu: C and HW must be positive");
  return launch_cg_bn_prelu_forward(x, gamma, beta, alpha, mean, invstd, var_unbiased, y, C, (int)HW, eps, training, (hipStream_t)stream);
}

int crnerf_bn_prelu_train_f32(const float* x, const float* gamma, const float* beta, const float* alpha, float* mean, float* invstd, float* var_unbiased,
                              float* y, float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, int C, int64_t HW, float eps,
                              void* stream) {
  REQUIRE(x, "x"); REQUIRE(gamma, "gamma"); REQUIRE(beta, "beta"); REQUIRE(alpha, "alpha"); REQUIRE(mean, "mean"); REQUIRE(invstd, "invstd"); REQUIRE(y, "y");
  REQUIRE(var_unbiased, "var_unbiased"); REQUIRE(running_mean, "running_mean"); REQUIRE(running_var, "running_var");
  if (C <= 0 || HW <= 0 || HW > (1 << 30)) return set_error(CRNERF_ERR_SHAPE, "bn_prelu_train: C and HW must be positive");
  if (!(momentum >= 0.0f && momentum <= 1.0f)) return set_error(CRNERF_ERR_CONFIG, "bn_prelu_train: momentum must be in [0, 1]");
  return launch_cg_bn_prelu_forward(x, gamma, beta, alpha, mean, invstd, var_unbiased, y, C, (int)HW, eps, 1, (hipStream_t)stream, running_mean, running_var,
                                    (long long*)num_batches_tracked, momentum);
}

int crnerf_bn_prelu_backward_f32(const float* x, const float* gamma, const float* beta, const float* alpha, const float* mean, const float* invstd,
                                 const float* d_y, float* d_x, float* d_gamma, float* d_beta, float* d_alpha, int C, int64_t HW, int training,
                                 void* stream) {
  REQUIRE(x, "x"); REQUIRE(gamma, "gamma"); REQUIRE(beta, "beta"); REQUIRE(alpha, "alpha"); REQUIRE(mean, "mean"); REQUIRE(invstd, "invstd");
  REQUIRE(d_y, "d_y"); REQUIRE(d_x, "d_x"); REQUIRE(d_gamma, "d_gamma"); REQUIRE(d_beta, "d_beta"); REQUIRE(d_alpha, "d_alpha");
  if (C <= 0 || HW <= 0 || HW > (1 << 30)) return set_error(CRNERF_ERR_SHAPE, "bn_prelu_backward: C and HW must be positive");
  return launch_cg_bn_prelu_backward(x, gamma, beta, alpha, mean, invstd, d_y, d_x, d_gamma, d_beta, d_alpha, C, (int)HW, training, (hipStream_t)stream);
}

int crnerf_avgpool3s2_f32(const float* in, float* out, int C, int H, int W, int backward, void* stream) {
  REQUIRE(in, "in"); REQUIRE(out, "out");
  if (C <= 0 || H <= 0 || W <= 0) return set_error(CRNERF_ERR_SHAPE, "avgpool3s2: non-positive shape");
  return launch_cg_avgpool(in, out, C, H, W, backward, (hipStream_t)stream);
}

int crnerf_fglo_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* stats, float* y, int C, int R, int64_t HW,
                    void* stream) {
  REQUIRE(x, "x"); REQUIRE(w1, "w1"); REQUIRE(b1, "b1"); REQUIRE(w2, "w2"); REQUIRE(b2, "b2"); REQUIRE(stats, "stats"); REQUIRE(y, "y");
  if (C <= 0 || R <= 0 || HW <= 0 || HW > (1 << 22)) return set_error(CRNERF_ERR_SHAPE, "fglo: non-positive shape");
  return launch_cg_fglo_forward(x, w1, b1, w2, b2, stats, y, C, R, (int)HW, (hipStream_t)stream);
}

int crnerf_fglo_backward_f32(const float* x, const float* w1, const float* w2, const float* stats, const float* d_y, float* scratch, float* d_x,
                             float* d_w1, float* d_b1, float* d_w2, float* d_b2, int C, int R, int64_t HW, void* stream) {
  REQUIRE(x, "x"); REQUIRE(w1, "w1"); REQUIRE(w2, "w2"); REQUIRE(stats, "stats"); REQUIRE(d_y, "d_y"); REQUIRE(scratch, "scratch"); REQUIRE(d_x, "d_x");
  REQUIRE(d_w1, "d_w1"); REQUIRE(d_b1, "d_b1"); REQUIRE(d_w2, "d_w2"); REQUIRE(d_b2, "d_b2");
  if (C <= 0 || C > 256 || R <= 0 || R > 64 || HW <= 0 || HW > (1 << 22)) return set_error(CRNERF_ERR_SHAPE, "fglo_backward: bad shape");
  return launch_cg_fglo_backward(x, w1, w2, stats, d_y, scratch, d_x, d_w1, d_b1, d_w2, d_b2, C, R, (int)HW, (hipStream_t)stream);
}

int crnerf_bilinear_gather_f32(const float* in, int h, int w, int Ho, int Wo, const int64_t* idx, int64_t n, int sigmoid, float* out, void* stream) {
  if (n == 0) return 0;
  REQUIRE(in, "in"); REQUIRE(out, "out");
  if (h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0 || n < 0) return set_error(CRNERF_ERR_SHAPE, "bilinear_gather: non-positive shape");
  if (!idx && n != (int64_t)Ho * Wo) return set_error(CRNERF_ERR_SHAPE, "bilinear_gather: idx == NULL means every output pixel (n = Ho*Wo)");
  return launch_cg_bilinear(in, (const long*)idx, out, (long)n, h, w, Ho, Wo, sigmoid, (hipStream_t)stream);
}

int crnerf_bilinear_gather_backward_f32(const float* out, const float* d_out, int h, int w, int Ho, int Wo, const int64_t* idx, int64_t n, int sigmoid,
                                        float* d_in, void* stream) {
  REQUIRE(d_in, "d_in");
  if (h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0 || n < 0) return set_error(CRNERF_ERR_SHAPE, "bilinear_gather_backward: non-positive shape");
  if (n > 0) { REQUIRE(d_out, "d_out"); if (sigmoid) REQUIRE(out, "out"); }
  if (!idx && n != (int64_t)Ho * Wo) return set_error(CRNERF_ERR_SHAPE, "bilinear_gather_backward: idx == NULL means every output pixel");
  return launch_cg_bilinear_backward(out, d_out, (const long*)idx, d_in, (long)n, h, w, Ho, Wo, sigmoid, (hipStream_t)stream);
}

int crnerf_cgnet_param_count(void) { return CGNET_PARAMS; }
int crnerf_cgnet_bn_count(void) { return CGNET_BNS; }
size_t crnerf_cgnet_arena_bytes(int32_t cin, int32_t H, int32_t W) {
  if (cin <= 0 || H <= 0 || W <= 0) return 0;
  return cgnet_arena_floats(cin, H, W) * sizeof(float);
}

static int cgnet_args(const char* who, int32_t cin, int32_t H, int32_t W, const float* const* params) {
  if (cin <= 0 || H < 1 || W < 1 || (long)H * W > (1L << 24)) return set_error(CRNERF_ERR_SHAPE, "cgnet: image must be cin >= 1 channels of 1 ... 2^24 pixels");
  REQUIRE(params, "params");
  for (int i = 0; i < CGNET_PARAMS; ++i)
    if (!params[i]) return set_error(CRNERF_ERR_NULL, who);
  return 0;
}

int crnerf_cgnet_forward_train_f32(const float* image, int32_t cin, int32_t H, int32_t W, const float* const* params, float* const* running_mean,
                                   float* const* running_var, int64_t* const* num_batches_tracked, float momentum, float eps, void* saved,
                                   float* mask, void* stream) {
  REQUIRE(image, "image"); REQUIRE(saved, "saved"); REQUIRE(mask, "mask"); REQUIRE(running_mean, "running_mean"); REQUIRE(running_var, "running_var");
  if (int rc = cgnet_args("cgnet_forward_train: a params[] entry is NULL", cin, H, W, params)) return rc;
  for (int i = 0; i < CGNET_BNS; ++i)
    if (!running_mean[i] || !running_var[i]) return set_error(CRNERF_ERR_NULL, "cgnet_forward_train: a running_mean / running_var entry is NULL");
  if (!(eps > 0.0f) || !(momentum >= 0.0f && momentum <= 1.0f)) return set_error(CRNERF_ERR_CONFIG, "cgnet_forward_train: eps must be positive and momentum in [0, 1]");
  const CgNetArgs a{cin, H, W, params, running_mean, running_var, (long long* const*)num_batches_tracked, momentum, eps};
  return launch_cgnet_forward_train(a, image, (float*)saved, mask, (hipStream_t)stream);
}

int crnerf_cgnet_backward_f32(const float* image, int32_t cin, int32_t H, int32_t W, const float* const* params, const void* saved, const float* mask,
                              const float* d_mask, void* scratch, float* const* grads, void* stream) {
  REQUIRE(image, "image"); REQUIRE(saved, "saved"); REQUIRE(mask, "mask"); REQUIRE(d_mask, "d_mask"); REQUIRE(scratch, "scratch"); REQUIRE(grads, "grads");
  if (int rc = cgnet_args("cgnet_backward: a params[] entry is NULL", cin, H, W, params)) return rc;
  for (int i = 0; i < CGNET_PARAMS; ++i)
    if (!grads[i]) return set_error(CRNERF_ERR_NULL, "cgnet_backward: a grads[] entry is NULL");
  const CgNetArgs a{cin, H, W, params, nullptr, nullptr, nullptr, 0.0f, 1e-3f};
  return launch_cgnet_backward(a, image, (const float*)saved, mask, d_mask, (float*)scratch, grads, (hipStream_t)stream);
}

}  // extern "C"
