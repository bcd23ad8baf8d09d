// "f32x3" fused volumetric renderer: the one-ray-per-wave renderer (round 1's render_fused.hip structure) on the x3 core (mlp_core_x3.h: fp32-accurate products from three-piece bf16
// splits on the bf16 matrix cores); everything around the MLP is the same fp32 code.
// Fused volumetric renderer: the whole of render_rays_cross_ray (models/rendering.py:50-196) for
// one ray per wavefront in ONE launch -- coarse depths, positional encoding, coarse NeRF_sigma,
// compositing, sample_pdf, z merge, fine NeRF_sigma, compositing.  Per ray the kernel reads 32 B
// (rays[r,0:8]) and writes weights/feature/depth; the [P,93] embeddings, the [P,256] activations
// and the [P,65] raw MLP output never exist in HBM (the reference materialises all three).
//
// Work decomposition: workgroup = 4 waves = 4 rays in lockstep on the shared LDS weight ring
// (mlp_core.h); a ray's N samples are walked in 32-sample tiles by its wave, carrying the running
// transmittance, so compositing needs no cross-wave traffic.  Grid-stride over ray quads.
#include <hip/hip_runtime.h>
#include "kernels.h"
#if defined(CRNERF_X_NP) && CRNERF_X_NP == 2
#include "mlp_core_h2t.h"
#else
#include "mlp_core_x3.h"
#endif
#include "posenc.h"
#include "ray_ops.h"

namespace crnerf {

struct RenderParamsX {
  const char* packed0;
  const char* packed1;
  const float* rays;
  const float* view_dir;
  const float* z_coarse;
  const float* z_steps;
  const float* u;
  long u_stride;
  const float* noise_c;
  const float* noise_f;
  float noise_std;
  int use_disp;
  long R;
  int Nc, Ni;
  int iters;
  float* weights_c;
  float* feature_c;
  float* depth_c;
  float* weights_f;
  float* feature_f;
  float* depth_f;
  float* z_fine;
  // in-kernel random draws (include/crnerf.h, philox.h: the same counters as the fp32 16x16x4 kernels, so one seed gives one set of draws
  // whichever kernel renders); rng_flags == 0: none
  unsigned long long rng_seed; long rng_ray_offset; int rng_flags; float perturb;
  float* z_coarse_out; float* noise_c_out; float* noise_f_out;
  int repair;   // crnerf_render_rays_f32x3_repair: one workgroup per ray quad; it leaves at once unless one of its rays came out of a previous render as NaN
};

// What the training twin adds (crnerf_render_rays_train_f32x3): every tile's layer activations + relu bits in the layout of the fp32 training
// twins (ActSaveX, mlp_core_x3.h) and its raw MLP output row, per pass -- the fp32 backward twins read them unchanged.
// LDS behind the four waves' ray scratch: one word per lane and pass, the running maximum |operand| of the h2 training twin (TrainHookX::publish_range)
constexpr int LDS_RANGE_X = LDS_SCRATCH_X + 4 * SCRATCH_BYTES;
constexpr int LDS_RANGE_BYTES = 2 * 256 * 4;
static_assert(LDS_RANGE_X + LDS_RANGE_BYTES <= 160 * 1024, "LDS budget of the training twin");
struct NoHookX {
  static constexpr bool on = false;
  __device__ __forceinline__ void publish_range(lds_char*, int, int, int) const {}
  __device__ __forceinline__ NoSaveX saver(int, long, int, int, bool, int) const { return NoSaveX(); }
  __device__ __forceinline__ void raw(int, long, int, int, bool, int, const f32x16 (&)[2], float) const {}
};
struct TrainHookX {
  static constexpr bool on = true;
  float* acts[2];   // [10][R*N][256] + masks, pass 0 = coarse (N = Nc), pass 1 = fine (N = Nc+Ni)
  float* rawo[2];   // [R*N][65]
  long R;
  // (pass ? [1] : [0], not [pass]: a dynamic index into a by-value kernel argument goes through private memory, the pointer comes back in a VGPR
  // and hipcc wraps EVERY row store in a readfirstlane waterfall loop -- twelve instructions and a branch per store; round 5, found in the ISA)
  __device__ __forceinline__ ActSaveX saver(int pass, long r, int N, int n, bool ok, int h) const {
    return ActSaveX{pass ? acts[1] : acts[0], R * N, r * N + n, ok, h, (uint32_t)(LDS_RANGE_X + 4 * ((pass ? 256 : 0) + (int)threadIdx.x))};
  }
  // the workgroup's largest |operand| per pass (the lanes' LDS words, raised tile by tile by mlp_tile_h2t) into the saved state's range word
  __device__ __forceinline__ void publish_range(lds_char* lds, int Nc, int Nf, int npass) const {
#if CRNERF_X_NP == 2
    __syncthreads();
    if ((int)threadIdx.x < npass) {
      const __attribute__((address_space(3))) uint32_t* w = (const __attribute__((address_space(3))) uint32_t*)(lds + LDS_RANGE_X) + 256 * threadIdx.x;
      uint32_t b = 0u;
      for (int t = 0; t < 256; ++t) b = w[t] > b ? w[t] : b;
      if (b != 0u) atomicMax(acts_range_word(threadIdx.x ? acts[1] : acts[0], R * (threadIdx.x ? Nf : Nc)), b);
    }
#endif
  }
  __device__ __forceinline__ void raw(int pass, long r, int N, int n, bool ok, int h, const f32x16 (&feat)[2], float sigma) const {
    if (!ok) return;
    float* o = (pass ? rawo[1] : rawo[0]) + (r * N + n) * OUT_DIM;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[32 * t + 8 * q + 4 * h + j] = feat[t][4 * q + j];
    if (h == 0) o[FEAT_DIM] = sigma;
  }
};

// RNG: the instantiation with the in-kernel draws -- a template parameter like render_fused16.hip's, so the kernels without draws carry no
// Philox registers through the MLP loop
template <bool RNG, class HOOK>
__device__ __forceinline__ void render_rays_x3_body(const RenderParamsX a, const HOOK hook) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int Nc = a.Nc, Ni = a.Ni, Nf = Nc + Ni;

  if (a.repair) {   // (bit test: this unit is built with -fno-honor-nans)
    const long rr = (long)blockIdx.x * 4 + wave;
    auto is_nan = [](float v) { return (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u; };
    const bool bad = rr < a.R && (is_nan(a.feature_c[rr * FEAT_DIM]) || (Ni > 0 && is_nan(a.feature_f[rr * FEAT_DIM])));
    if (!__syncthreads_or(bad)) return;
  }
  load_consts(lds, a.packed0, a.packed1);
#if CRNERF_X_NP == 2
  // a pack that carries the range flag (crnerf_pack_mlp_weights_h2_async of a weight >= 255): nothing is rendered, every ray of this workgroup is
  // marked NaN -- what crnerf_render_rays[_train]_f32x3_repair looks for -- so precision "auto" needs no host round trip to fall back
  if ((__float_as_uint(((const lds_float*)(lds + LDS_CONST0))[H2_FLAG_WORD]) | __float_as_uint(((const lds_float*)(lds + LDS_CONST1))[H2_FLAG_WORD])) != 0u) {
    for (int it = 0; it < a.iters; ++it) {
      const long rr = ((long)it * gridDim.x + blockIdx.x) * 4 + wave;
      if (rr < a.R && lane == 0) {
        a.feature_c[rr * FEAT_DIM] = __uint_as_float(0x7fc00000u);
        if (Ni > 0) a.feature_f[rr * FEAT_DIM] = __uint_as_float(0x7fc00000u);
      }
    }
    return;
  }
#endif
  RayScratch scr;
  scr.bind(lds + LDS_SCRATCH_X + wave * SCRATCH_BYTES);
#if CRNERF_X_NP == 2
  if constexpr (HOOK::on) {   // this lane's two range words (its own: no barrier needed before it raises them)
    __attribute__((address_space(3))) uint32_t* w = (__attribute__((address_space(3))) uint32_t*)(lds + LDS_RANGE_X) + threadIdx.x;
    w[0] = 0u; w[256] = 0u;
  }
#endif

  const int tiles_c = (Nc + 31) >> 5, tiles_f = Ni > 0 ? (Nf + 31) >> 5 : 0;
  WeightPipeX pipe;
  pipe.start(lds, a.packed0 + CONST_BYTES, a.packed1 + CONST_BYTES, tiles_c, tiles_c + tiles_f, lane, wave);
  xu32x4 cur[X_AHEAD];
  pipe.prime(cur);
  PhaseTimer tm;
  tm.start(blockIdx.x == 0 && threadIdx.x == 0);

#pragma unroll 1
  for (int it = 0; it < a.iters; ++it) {
    const long rr = ((long)it * gridDim.x + blockIdx.x) * 4 + wave;
    const bool ray_ok = rr < a.R;
    const long r = ray_ok ? rr : a.R - 1;
    const float* ray = a.rays + r * 8;
    const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
    const float near = ray[6], far = ray[7];
    f32x16 dv[1];
    {
      // rendering.py:155  dir_embedded = embedding_dir(kwargs.get('view_dir', rays_d))
      const float* vd = a.view_dir ? a.view_dir + r * 3 : ray + 3;
      float tmp[16];
      posenc_regs<DIR_FREQS, 8>(vd[0], vd[1], vd[2], h, tmp);
#pragma unroll
      for (int i = 0; i < 16; ++i) dv[0][i] = tmp[i];
    }
    const RayRng rng{(uint32_t)(a.rng_seed & 0xffffffffu), (uint32_t)(a.rng_seed >> 32)};
    const long rng_ray = a.rng_ray_offset + r;
    for (int n = lane; n < Nc; n += 64) {
      auto zlin = [&](int k) { return coarse_depth(near, far, a.z_steps ? a.z_steps[k] : linspace01(k, Nc), a.use_disp); };
      float zv;
      if (a.z_coarse) zv = a.z_coarse[r * Nc + n];
      else if (RNG && (a.rng_flags & 1)) {
        // rendering.py:169-176: mid-points, lower / upper interval ends, z = lower + (upper - lower) * (perturb * U[0,1))
        const float z0 = zlin(n);
        const float lower = n == 0 ? z0 : 0.5f * (zlin(n - 1) + z0);
        const float upper = n == Nc - 1 ? z0 : 0.5f * (z0 + zlin(n + 1));
        zv = lower + (upper - lower) * (a.perturb * rng.uniform(RNG_STREAM_JITTER, rng_ray, n));
      } else zv = zlin(n);
      scr.zc[n] = zv;
      if (RNG && a.z_coarse_out && ray_ok) a.z_coarse_out[r * Nc + n] = zv;
    }
    wave_lds_fence();

    // pass 0 = coarse model on zc, pass 1 = fine model on the merged zs; ONE copy of the MLP code
    const int npass = Ni > 0 ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      const int N = pass ? Nf : Nc;
      const lds_float* zsrc = pass ? scr.zs : scr.zc;
      const float* noise_row = pass ? (a.noise_f ? a.noise_f + r * Nf : nullptr) : (a.noise_c ? a.noise_c + r * Nc : nullptr);
      float* weights_row = pass ? a.weights_f + r * Nf : a.weights_c + r * Nc;
      CompositeState st;
      st.reset();
      const int tiles = (N + 31) >> 5;
#pragma unroll 1
      for (int tile = 0; tile < tiles; ++tile) {
        const int n = tile * 32 + p;
        const bool valid = n < N;
        const int nc = valid ? n : N - 1;
        const float zn = zsrc[nc];
        const float znext = zsrc[nc + 1 < N ? nc + 1 : N - 1];
        // rendering.py:178 / :188  xyz = rays_o + rays_d * z  (separate mul and add)
        const float x = ox + dx * zn, y = oy + dy * zn, z = oz + dz * zn;
        f32x16 pe[3], feat[2];
        {
          float tmp[48];
          posenc_regs<XYZ_FREQS, 24>(x, y, z, h, tmp);
#pragma unroll
          for (int i = 0; i < 48; ++i) pe[i / 16][i % 16] = tmp[i];
        }
        float sigma;
        const auto sv = hook.saver(pass, r, N, n, valid && ray_ok, h);
        mlp_tile_x3(pipe, pass, pe, dv, feat, sigma, h, cur, tm, sv);
        hook.raw(pass, r, N, n, valid && ray_ok, h, feat, sigma);
        float noise = (noise_row && valid) ? noise_row[n] * a.noise_std : 0.0f;
        if (RNG && (a.rng_flags & 4) && valid) {          // rendering.py:125  noise = randn_like(sigma) * noise_std
          const float draw = rng.normal(pass ? RNG_STREAM_NOISE_FINE : RNG_STREAM_NOISE_COARSE, rng_ray, n);
          noise = draw * a.noise_std;
          float* no = pass ? a.noise_f_out : a.noise_c_out;
          if (no && ray_ok && h == 0) no[r * N + n] = draw;
        }
        const float w = composite_tile(st, feat, sigma, noise, zn, znext, n == N - 1, valid, p);
        if (valid && h == 0) {
          if (ray_ok) weights_row[n] = w;
          if (pass == 0) scr.wc[n] = w;
        }
        tm.tick(T_COMPOSITE);
      }
      composite_finish(st);
      if (ray_ok)
        store_ray_feature(st, (pass ? a.feature_f : a.feature_c) + r * FEAT_DIM, (pass ? a.depth_f : a.depth_c) + r, p, h);
      if (pass == 0 && Ni > 0) {
        wave_lds_fence();
        sample_pdf_wave(scr, Nc, Ni, a.u ? a.u + r * a.u_stride : nullptr, lane, (RNG && (a.rng_flags & 2)) ? &rng : nullptr, rng_ray);
        merge_sort_wave(scr, Nc, Ni, lane);
        if (a.z_fine && ray_ok)
          for (int n = lane; n < Nf; n += 64) a.z_fine[r * Nf + n] = scr.zs[n];
      }
      tm.tick(T_RAYLEVEL);
    }
  }
  tm.flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  hook.publish_range(lds, Nc, Nf, Ni > 0 ? 2 : 1);
}

__global__ __launch_bounds__(256, 1) void render_rays_x3_kernel(RenderParamsX a) { render_rays_x3_body<false>(a, NoHookX()); }
__global__ __launch_bounds__(256, 1) void render_rays_x3_rng_kernel(RenderParamsX a) { render_rays_x3_body<true>(a, NoHookX()); }
__global__ __launch_bounds__(256, 1) void render_rays_train_x3_kernel(RenderParamsX a, TrainHookX hook) { render_rays_x3_body<false>(a, hook); }
__global__ __launch_bounds__(256, 1) void render_rays_train_x3_rng_kernel(RenderParamsX a, TrainHookX hook) { render_rays_x3_body<true>(a, hook); }

int launch_render_rays_x3(const RenderArgs& a, hipStream_t stream) {
  if (a.R <= 0) return 0;
  if (a.Nc < 2 || a.Nc > MAX_NC) return set_error(-2, "render_rays_f32x3: N_samples must be in [2, 256] for the fused kernel");
  if (a.Ni < 0 || a.Ni > MAX_NI) return set_error(-2, "render_rays_f32x3: N_importance must be in [0, 256] for the fused kernel");
  if (a.Ni > 0 && a.Nc < 3) return set_error(-2, "render_rays_f32x3: hierarchical sampling needs N_samples >= 3");
  if (a.Ni > 0 && !a.packed_fine) return set_error(-3, "render_rays_f32x3: N_importance > 0 but no fine model");
  RenderParamsX k;
  k.packed0 = (const char*)a.packed_coarse;
  k.packed1 = (const char*)(a.packed_fine ? a.packed_fine : a.packed_coarse);
  k.rays = a.rays; k.view_dir = a.view_dir; k.z_coarse = a.z_coarse; k.z_steps = a.z_steps; k.u = a.u; k.u_stride = a.u_stride;
  k.noise_c = a.noise_coarse; k.noise_f = a.noise_fine; k.noise_std = a.noise_std; k.use_disp = a.use_disp;
  k.R = a.R; k.Nc = a.Nc; k.Ni = a.Ni;
  k.rng_seed = a.rng_seed; k.rng_ray_offset = a.rng_ray_offset; k.rng_flags = a.rng_flags; k.perturb = a.perturb;
  k.z_coarse_out = a.z_coarse_out; k.noise_c_out = a.noise_coarse_out; k.noise_f_out = a.noise_fine_out;
  const bool rngk = a.rng_flags != 0 || a.z_coarse_out;
  k.weights_c = a.weights_coarse; k.feature_c = a.feature_coarse; k.depth_c = a.depth_coarse;
  k.weights_f = a.weights_fine; k.feature_f = a.feature_fine; k.depth_f = a.depth_fine; k.z_fine = a.z_fine;
  const long quads = (a.R + 3) / 4;
  const int cus = num_cus();
  if (a.repair && quads > 0x7fffffffL) return set_error(-3, "render_rays_f32x3_repair: renders of < 2^33 rays only");
  k.repair = a.repair;
  const int grid = a.repair ? (int)quads : (int)(quads < cus ? quads : cus);   // one workgroup per CU, persistent over ray quads (repair: one per quad)
  k.iters = (int)((quads + grid - 1) / grid);
  const size_t shmem = LDS_SCRATCH_X + 4 * SCRATCH_BYTES;
  if (a.train_acts_coarse) {   // training twin (crnerf_render_rays_train_f32x3 / _f32h2; repair: the rays the h2 twin poisoned, saved rows included)
    if (a.Ni > 0 && (!a.train_acts_fine || !a.train_raw_fine)) return set_error(-1, "render_rays_train_f32x3: fine buffers are NULL");
    if (!a.train_raw_coarse) return set_error(-1, "render_rays_train_f32x3: raw_coarse is NULL");
    if ((unsigned long long)a.R * (unsigned)(a.Nc + a.Ni) * 1024ull >= (unsigned long long)SAVEX_OOB)
      return set_error(-2, "render_rays_train_f32x3: more than 3.9 M sample points per pass and call (the saved rows are addressed with 32-bit offsets)");
    TrainHookX h{{(float*)a.train_acts_coarse, (float*)a.train_acts_fine}, {a.train_raw_coarse, a.train_raw_fine}, a.R};
    if (!a.repair) {   // the range words of both passes start at zero (the repair launch leaves what the h2 twin raised: kernels.h acts_range_word)
      if (int rc = zero_acts_range(a.train_acts_coarse, a.R * a.Nc, stream)) return rc;
      if (a.Ni > 0)
        if (int rc = zero_acts_range(a.train_acts_fine, a.R * (a.Nc + a.Ni), stream)) return rc;
    }
    const void* fn = rngk ? (const void*)render_rays_train_x3_rng_kernel : (const void*)render_rays_train_x3_kernel;
    const size_t shmem_t = shmem + LDS_RANGE_BYTES;      // + the h2 twin's per-lane range words (x3: unused)
    if (int rc = ensure_dynamic_lds(fn, shmem_t, "render_rays_train_x3_kernel")) return rc;
    if (rngk) hipLaunchKernelGGL(render_rays_train_x3_rng_kernel, dim3(grid), dim3(256), shmem_t, stream, k, h);
    else hipLaunchKernelGGL(render_rays_train_x3_kernel, dim3(grid), dim3(256), shmem_t, stream, k, h);
    return check_launch("render_rays_train_x3_kernel");
  }
  const void* fn = rngk ? (const void*)render_rays_x3_rng_kernel : (const void*)render_rays_x3_kernel;
  if (int rc = ensure_dynamic_lds(fn, shmem, "render_rays_x3_kernel")) return rc;
  if (rngk) hipLaunchKernelGGL(render_rays_x3_rng_kernel, dim3(grid), dim3(256), shmem, stream, k);
  else hipLaunchKernelGGL(render_rays_x3_kernel, dim3(grid), dim3(256), shmem, stream, k);
  return check_launch("render_rays_x3_kernel");
}

#ifdef CRNERF_TIMING
#if CRNERF_X_NP == 2
extern "C" int crnerf_debug_read_timing_h2(unsigned long long* host_out) {
#else
extern "C" int crnerf_debug_read_timing_x3(unsigned long long* host_out) {
#endif
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crnerf_timing), sizeof(unsigned long long) * T_COUNT);
}
#endif

}  // namespace crnerf
