// Fused volumetric renderer on the v16 core (mlp_core16.h): two wavefronts per SIMD.
// Contract: render_rays_cross_ray, models/rendering.py:50-196 (coarse pass, sample_pdf + merge, fine pass; outputs per A0 of SURVEY 8a).
//
// Work decomposition: workgroup = 8 waves = 4 rays; a ray belongs to a PAIR of waves (A = even wave,
// B = odd wave).  The ray's samples are walked in 32-sample steps; in step k wave A owns samples
// [32k, 32k+16), wave B [32k+16, 32k+32), each as one 16-point MFMA tile.  The pair exchanges only
// scalars through LDS: the product of (1-alpha) over each wave's 16 samples per step (running
// transmittance), and the per-ray feature/depth partial sums at the end of a pass.  sample_pdf and the
// merge run on the pair's 128 lanes.  All 8 waves execute identical control flow, so plain workgroup
// barriers order the exchanges.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core16.h"
#include "mlp_train16.h"
#include "pair_ops.h"
#include "ray_ops.h"

namespace crnerf {

struct RenderParams16 {
  const char* packed0; const char* packed1;
  const float* rays; const float* view_dir; const float* z_coarse; const float* z_steps; const float* u; long u_stride;
  const float* noise_c; const float* noise_f; float noise_std; int use_disp;
  long R; int Nc, Ni, iters;
  unsigned int* sched;   // null: static quad -> workgroup map (iters passes); else {next-quad counter, finished-workgroup counter}
  float* weights_c; float* feature_c; float* depth_c; float* weights_f; float* feature_f; float* depth_f; float* z_fine;
  // in-kernel random draws (include/crnerf.h, philox.h); rng_flags == 0: none
  unsigned long long rng_seed; long rng_ray_offset; int rng_flags; float perturb;
  float* z_coarse_out; float* noise_c_out; float* noise_f_out;
};

static __device__ unsigned int crnerf_sched16[SCHED_SLOTS][2];   // kernels.h "Dynamic work distribution"

// What the training twin adds to the renderer (crnerf_render_rays_train_f32): every tile's layer activations + relu bits
// (ActSaver, mlp_train16.h) and its raw MLP output row, per pass.  The inference kernel instantiates the no-op hook.
struct NoHook {
  __device__ __forceinline__ NoSave saver(int, long, int, int, bool, int) const { return NoSave(); }
  __device__ __forceinline__ void raw(int, long, int, int, bool, int, const f32x4 (&)[4], float) const {}
};
struct TrainHook {
  float* acts[2];   // [10][R*N][256] + masks, pass 0 = coarse (N = Nc), pass 1 = fine (N = Nc+Ni)
  float* rawo[2];   // [R*N][65]
  long R;
  __device__ __forceinline__ ActSaver saver(int pass, long r, int N, int n, bool ok, int g) const {
    return ActSaver{pass ? acts[1] : acts[0], R * N, r * N + n, ok, g};
  }
  __device__ __forceinline__ void raw(int pass, long r, int N, int n, bool ok, int g, const f32x4 (&feat)[4], float sigma) const {
    if (!ok) return;
    float* o = (pass ? rawo[1] : rawo[0]) + (r * N + n) * OUT_DIM;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) o[16 * T + 4 * g + q] = feat[T][q];
    if (g == 0) o[FEAT_DIM] = sigma;
  }
};

// RNG: the instantiation with the in-kernel random draws (philox.h).  It is a template parameter, not a runtime flag: the keys, the
// global ray index and three more output pointers would otherwise be live across the MLP of EVERY launch (measured: 10 -> 20
// spilled registers in the inference kernel, 77 -> 90 in the training twin).
template <bool RNG, class HOOK>
__device__ __forceinline__ void render_rays16_impl(const RenderParams16& a, const HOOK& hook) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int pair = wave >> 1, half = wave & 1;
  const int lane128 = half * 64 + lane;
  const int Nc = a.Nc, Ni = a.Ni, Nf = Nc + Ni;

  load_consts(lds, a.packed0, a.packed1);
  PairScratch scr;
  scr.bind(lds + LDS_SCRATCH + pair * PAIR_BYTES);
  lds_float* dirbuf = (lds_float*)(lds + LDS_SCRATCH + 4 * PAIR_BYTES) + wave * 32;   // 32 floats per wave

  const int steps_c = (Nc + 31) >> 5, steps_f = Ni > 0 ? (Nf + 31) >> 5 : 0;
  WeightPipe16 pipe;
  pipe.start(lds, a.packed0 + CONST_BYTES, a.packed1 + CONST_BYTES, steps_c, steps_c + steps_f, lane, wave);
  f32x4 q[V16_AHEAD];
  pipe.prime(q);
  PhaseTimer tm;
  tm.start(blockIdx.x == 0 && threadIdx.x == 0);

  // ray quads: statically strided over the grid, or (a.sched) pulled from a device counter -- the index for the NEXT pass is
  // requested at the top of a pass and picked up at its end, so the atomic's round trip is never waited for
  __attribute__((address_space(3))) unsigned int* qslot =
      (__attribute__((address_space(3))) unsigned int*)(lds + LDS_SCRATCH + 4 * PAIR_BYTES + V16_WAVES * 32 * sizeof(float));
  const long quads = (a.R + 3) / 4;
  long quad = blockIdx.x;
#pragma unroll 1
  for (int it = 0; a.sched ? quad < quads : it < a.iters; ++it) {
    unsigned int nxt = 0;
    if (a.sched && threadIdx.x == 0) nxt = gridDim.x + atomicAdd(a.sched, 1u);
    const long rr = quad * 4 + pair;
    const bool ray_ok = rr < a.R;
    const long r = ray_ok ? rr : a.R - 1;
    const float* ray = a.rays + r * 8;
    const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
    const float near = ray[6], far = ray[7];
    {
      const float* vd = a.view_dir ? a.view_dir + r * 3 : ray + 3;       // rendering.py:155
      f32x4 dv[2];
      posenc_regs16<DIR_FREQS, 2>(vd[0], vd[1], vd[2], g, dv);
      if (p == 0) {                                                      // per-ray constant: park it in LDS (DirLds)
        *(__attribute__((address_space(3))) f32x4*)(dirbuf + 8 * g) = dv[0];
        *(__attribute__((address_space(3))) f32x4*)(dirbuf + 8 * g + 4) = dv[1];
      }
    }
    const DirLds dirsrc{dirbuf + 8 * g};
    const RayRng rng{(uint32_t)(a.rng_seed & 0xffffffffu), (uint32_t)(a.rng_seed >> 32)};
    const long rng_ray = a.rng_ray_offset + r;
    for (int n = lane128; n < Nc; n += 128) {
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
    wg_barrier();

    const int npass = Ni > 0 ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      const int N = pass ? Nf : Nc;
      const lds_float* zsrc = pass ? scr.zs : scr.zc;
      const float* noise_row = pass ? (a.noise_f ? a.noise_f + r * Nf : nullptr) : (a.noise_c ? a.noise_c + r * Nc : nullptr);
      float* weights_row = pass ? a.weights_f + r * Nf : a.weights_c + r * Nc;
      double carry = 1.0;
      f32x4 facc[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) facc[T] = f32x4{0, 0, 0, 0};
      float dacc = 0.0f;
      const int steps = (N + 31) >> 5;
#pragma unroll 1
      for (int k = 0; k < steps; ++k) {
        const int n = 32 * k + 16 * half + p;
        const bool valid = n < N;
        const int nc = valid ? n : N - 1;
        const float zn = zsrc[nc];
        const float znext = zsrc[nc + 1 < N ? nc + 1 : N - 1];
        const float x = ox + dx * zn, y = oy + dy * zn, z = oz + dz * zn;   // rendering.py:178 / :188
        f32x4 pe[6], feat[4];
        posenc_regs16<XYZ_FREQS, 6>(x, y, z, g, pe);
        float sigma;
        mlp_tile16(pipe, pass, pe, dirsrc, feat, sigma, g, q, tm, hook.saver(pass, r, N, n, valid && ray_ok, g));
        hook.raw(pass, r, N, n, valid && ray_ok, g, feat, sigma);
        // ---- compositing, rendering.py:121-143
        float noise = (noise_row && valid) ? noise_row[n] * a.noise_std : 0.0f;
        if (RNG && (a.rng_flags & 4) && valid) {          // rendering.py:125  noise = randn_like(sigma) * noise_std
          const float draw = rng.normal(pass ? RNG_STREAM_NOISE_FINE : RNG_STREAM_NOISE_COARSE, rng_ray, n);
          noise = draw * a.noise_std;
          float* no = pass ? a.noise_f_out : a.noise_c_out;
          if (no && ray_ok && g == 0) no[r * N + n] = draw;
        }
        const float delta = (n == N - 1) ? 1e2f : znext - zn;
        const float alpha = valid ? 1.0f - expf(-delta * fmaxf(sigma + noise, 0.0f)) : 0.0f;
        double incl = (double)(1.0f - alpha);
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          const double o = shfl_up_f64(incl, d, 16);
          if (p >= d) incl *= o;
        }
        double excl = shfl_up_f64(incl, 1, 16);
        if (p == 0) excl = 1.0;
        const double tot = shfl_f64(incl, 15, 16);
        if (lane == 0) scr.xprod[half] = tot;
        wg_barrier();
        const double other = scr.xprod[half ^ 1];
        const double prod_a = half ? other : tot, prod_b = half ? tot : other;
        const float Tr = (float)(half ? carry * prod_a * excl : carry * excl);
        carry = carry * prod_a * prod_b;
        const float w = alpha * Tr;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
          for (int rr2 = 0; rr2 < 4; ++rr2) facc[T][rr2] += w * feat[T][rr2];
        dacc += w * zn;
        if (valid && g == 0) {
          if (ray_ok) weights_row[n] = w;
          if (pass == 0) scr.wc[n] = w;
        }
        tm.tick(T_COMPOSITE);
      }
      // per-wave reduction over the 16 point lanes, then pair combine through LDS
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
          for (int rr2 = 0; rr2 < 4; ++rr2) facc[T][rr2] += __shfl_xor(facc[T][rr2], d);
        dacc += __shfl_xor(dacc, d);
      }
      if (half == 1 && p == 0) {
#pragma unroll
        for (int T = 0; T < 4; ++T) *(__attribute__((address_space(3))) f32x4*)(scr.xfeat + 16 * T + 4 * g) = facc[T];
        if (g == 0) scr.xfeat[64] = dacc;
      }
      wg_barrier();
      if (half == 0 && p == 0 && ray_ok) {
        float* frow = (pass ? a.feature_f : a.feature_c) + r * FEAT_DIM;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          const f32x4 o = *(const __attribute__((address_space(3))) f32x4*)(scr.xfeat + 16 * T + 4 * g);
          *(f32x4*)(frow + 16 * T + 4 * g) = facc[T] + o;
        }
        if (g == 0) (pass ? a.depth_f : a.depth_c)[r] = dacc + scr.xfeat[64];
      }
      if (pass == 0 && Ni > 0) {
        sample_pdf_pair(scr, Nc, Ni, a.u ? a.u + r * a.u_stride : nullptr, lane, lane128, (RNG && (a.rng_flags & 2)) ? &rng : nullptr, rng_ray);
        wg_barrier();
        merge_sort_pair(scr, Nc, Ni, lane128);
        wg_barrier();
        if (a.z_fine && ray_ok)
          for (int n = lane128; n < Nf; n += 128) a.z_fine[r * Nf + n] = scr.zs[n];
      }
      tm.tick(T_RAYLEVEL);
    }
    if (a.sched && threadIdx.x == 0) *qslot = nxt;
    wg_barrier();   // scratch is rewritten by the next ray
    quad = a.sched ? (long)__builtin_amdgcn_readfirstlane((int)*qslot) : quad + gridDim.x;   // slot rewritten a whole pass later
  }
  if (a.sched && threadIdx.x == 0 && atomicAdd(a.sched + 1, 1u) == gridDim.x - 1) {   // last workgroup out: leave the slot zeroed
    atomicExch(a.sched, 0u);       // device-scope, like the increments: the per-XCD L2s are not coherent for plain stores
    atomicExch(a.sched + 1, 0u);
  }
  tm.flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512, 2) void render_rays16_kernel(RenderParams16 a) { render_rays16_impl<false>(a, NoHook{}); }
__global__ __launch_bounds__(512, 2) void render_rays16_rng_kernel(RenderParams16 a) { render_rays16_impl<true>(a, NoHook{}); }

// Fused TRAINING forward: the same launch (posenc -> MLP -> compositing, coarse -> sample_pdf -> fine) that also keeps what
// the backward twins need -- no [P,93] / [P,120] embeddings and no separate compositing pass ever exist in HBM.
__global__ __launch_bounds__(512, 2) void render_rays_train16_kernel(RenderParams16 a, TrainHook h) { render_rays16_impl<false>(a, h); }
__global__ __launch_bounds__(512, 2) void render_rays_train16_rng_kernel(RenderParams16 a, TrainHook h) { render_rays16_impl<true>(a, h); }

int launch_render_rays16(const RenderArgs& a, hipStream_t stream) {
  if (a.R <= 0) return 0;
  if (a.Nc < 2 || a.Nc > MAX_NC) return set_error(-2, "render_rays: N_samples must be in [2, 256] for the fused kernel");
  if (a.Ni < 0 || a.Ni > MAX_NI) return set_error(-2, "render_rays: N_importance must be in [0, 256] for the fused kernel");
  if (a.Ni > 0 && a.Nc < 3) return set_error(-2, "render_rays: hierarchical sampling needs N_samples >= 3");
  if (a.Ni > 0 && !a.packed_fine) return set_error(-3, "render_rays: N_importance > 0 but no fine model");
  RenderParams16 k;
  k.packed0 = (const char*)a.packed_coarse;
  k.packed1 = (const char*)(a.packed_fine ? a.packed_fine : a.packed_coarse);
  k.rays = a.rays; k.view_dir = a.view_dir; k.z_coarse = a.z_coarse; k.z_steps = a.z_steps; k.u = a.u; k.u_stride = a.u_stride;
  k.noise_c = a.noise_coarse; k.noise_f = a.noise_fine; k.noise_std = a.noise_std; k.use_disp = a.use_disp;
  k.R = a.R; k.Nc = a.Nc; k.Ni = a.Ni;
  k.weights_c = a.weights_coarse; k.feature_c = a.feature_coarse; k.depth_c = a.depth_coarse;
  k.weights_f = a.weights_fine; k.feature_f = a.feature_fine; k.depth_f = a.depth_fine; k.z_fine = a.z_fine;
  k.rng_seed = a.rng_seed; k.rng_ray_offset = a.rng_ray_offset; k.rng_flags = a.rng_flags; k.perturb = a.perturb;
  k.z_coarse_out = a.z_coarse_out; k.noise_c_out = a.noise_coarse_out; k.noise_f_out = a.noise_fine_out;
  const long quads = (a.R + 3) / 4;
  const int cus = num_cus();
  const int grid = (int)(quads < cus ? quads : cus);
  k.iters = (int)((quads + grid - 1) / grid);
  k.sched = k.iters > 1 ? sched_slot((const void*)crnerf_sched16) : nullptr;
  const size_t shmem = LDS_SCRATCH + 4 * PAIR_BYTES + V16_WAVES * 32 * sizeof(float) + 16;
  if (a.train_acts_coarse) {
    if (a.Ni > 0 && (!a.train_acts_fine || !a.train_raw_fine)) return set_error(-1, "render_rays_train: fine buffers are NULL");
    if (!a.train_raw_coarse) return set_error(-1, "render_rays_train: raw_coarse is NULL");
    TrainHook h{{(float*)a.train_acts_coarse, (float*)a.train_acts_fine}, {a.train_raw_coarse, a.train_raw_fine}, a.R};
    if (int rc = zero_acts_range(a.train_acts_coarse, a.R * a.Nc, stream)) return rc;     // the fp32 twin tracks no operand range (kernels.h)
    if (a.Ni > 0)
      if (int rc = zero_acts_range(a.train_acts_fine, a.R * (a.Nc + a.Ni), stream)) return rc;
    const bool rngk = a.rng_flags != 0 || a.z_coarse_out;
    const void* fn = rngk ? (const void*)render_rays_train16_rng_kernel : (const void*)render_rays_train16_kernel;
    if (int rc = ensure_dynamic_lds(fn, shmem, "render_rays_train16_kernel")) return rc;
    if (rngk) hipLaunchKernelGGL(render_rays_train16_rng_kernel, dim3(grid), dim3(512), shmem, stream, k, h);
    else hipLaunchKernelGGL(render_rays_train16_kernel, dim3(grid), dim3(512), shmem, stream, k, h);
    return check_launch("render_rays_train16_kernel");
  }
  const bool rngk = a.rng_flags != 0 || a.z_coarse_out;
  const void* fn = rngk ? (const void*)render_rays16_rng_kernel : (const void*)render_rays16_kernel;
  if (int rc = ensure_dynamic_lds(fn, shmem, "render_rays16_kernel")) return rc;
  if (rngk) hipLaunchKernelGGL(render_rays16_rng_kernel, dim3(grid), dim3(512), shmem, stream, k);
  else hipLaunchKernelGGL(render_rays16_kernel, dim3(grid), dim3(512), shmem, stream, k);
  return check_launch("render_rays16_kernel");
}

#ifdef CRNERF_TIMING
extern "C" int crnerf_debug_read_timing16(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crnerf_timing), sizeof(unsigned long long) * T_COUNT);
}
#endif

}  // namespace crnerf
