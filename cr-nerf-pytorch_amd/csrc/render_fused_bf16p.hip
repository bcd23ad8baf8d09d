// Fused volumetric renderer on the bf16 matrix cores, pair core (mlp_core_bf16p.h): the whole of render_rays_cross_ray
// (models/rendering.py:50-196) in ONE launch, NeRF_sigma in the mixed-precision semantics of include/crnerf.h (bf16 MFMA
// operands, fp32 accumulate), everything around the MLP in fp32.
//
// Work decomposition: workgroup = 8 waves (two per SIMD) = 4 rays; a ray belongs to a PAIR of waves (A = even wave, B = odd
// wave).  The ray's samples are walked in 64-sample steps; in step k wave A owns samples [64k, 64k+32), wave B
// [64k+32, 64k+64), each as one 32-point MFMA tile.  The pair exchanges only scalars through LDS: the product of (1 - alpha)
// over each wave's 32 samples per step (running transmittance), and the per-ray feature / depth partial sums at the end of a
// pass.  sample_pdf and the merge run on the pair's 128 lanes (pair_ops.h).  All 8 waves execute identical control flow, so
// plain workgroup barriers order the exchanges.  Per-ray feature sums: the 32 products w_n * feat of a tile are reduced over
// the wave's 32 point lanes by a transpose-fold (31 cross-lane moves) into ONE register per lane, so that no 32-register
// accumulator is live across the MLP (the kernel must fit 256 registers for two waves per SIMD).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core_bf16p.h"
#include "pair_ops.h"
#include "ray_ops.h"

namespace crnerf {

struct RenderParamsP {
  const char* packed0; const char* packed1;
  const float* rays; const float* view_dir; const float* z_coarse; const float* z_steps; const float* u; long u_stride;
  const float* noise_c; const float* noise_f; float noise_std; int use_disp;
  long R; int Nc, Ni, iters;
  unsigned int* sched;   // null: static quad -> workgroup map (iters passes); else {next-quad counter, finished-workgroup counter}
  float* weights_c; float* feature_c; float* depth_c; float* weights_f; float* feature_f; float* depth_f; float* z_fine;
  const float* wc_in;    // render_rays_bf16p_fine_kernel: the coarse pass's weights [R, Nc], computed elsewhere (crnerf_render_rays_bf16_fine)
};

constexpr int LDS_DIR_P = LDS_SCRATCH_P + 4 * PAIR_BYTES;        // 8 waves x 64 B: the ray's direction embedding as B operands
constexpr int LDS_QSLOT_P = LDS_DIR_P + P_WAVES * 64;
constexpr int LDS_TOTAL_P = LDS_QSLOT_P + 16;
constexpr int LDS_STAGE_P = LDS_TOTAL_P;                         // training twin: 8 waves x SAVE_LDS_WAVE, the tile pairs' rows on their way out
constexpr int LDS_TOTAL_TRAIN_P = LDS_STAGE_P + P_WAVES * SAVE_LDS_WAVE;
static_assert(LDS_TOTAL_TRAIN_P <= 160 * 1024 && LDS_STAGE_P % 16 == 0, "LDS budget of the training twin");

// What the training twin adds (crnerf_render_rays_train_bf16): per pass the saved state of mlp_core_bf16p.h "training twin" -- activation rows,
// activity bits, the embedded input, all written from the registers they are born in -- and the raw MLP output rows [R*N,65] fp32.  The inference
// kernel instantiates the no-op hook.
struct NoHookP {
  static constexpr bool on = false;
  __device__ __forceinline__ NoSaveP saver(int, long, int, int, int, bool, int, int, lds_char*) const { return NoSaveP(); }
};
struct TrainHookP {
  static constexpr bool on = true;
  char* acts[2];    // crnerf_mlp_train_mixed_acts_bytes(R*N) each, pass 0 = coarse (N = Nc), pass 1 = fine (N = Nc+Ni)
  float* rawo[2];   // [R*N][65]
  long R;
  __device__ __forceinline__ bool stores_on() const {
    return true;
  }
  // n0: the tile's first sample, n = n0 + p this lane's; ray_ok: r < R
  __device__ __forceinline__ ActSaveP saver(int pass, long r, int N, int n0, int n, bool ray_ok, int lane, int wave, lds_char* lds) const {
    const long P = R * N, pt = r * N + n;
    const int h = lane >> 5;
    bool ok = ray_ok && n < N;
    int pts = ray_ok ? (N - n0 < 32 ? N - n0 : 32) : 0;   // points of this tile (<= 0: none)
    ActSaveP sv;
    sv.acts = pass ? acts[1] : acts[0]; sv.slot_bytes = P * 512; sv.P = P;
    sv.boff = ok ? (uint32_t)pt * 32u + 8u * (uint32_t)h : SAVE_OOB;
    sv.toff = (uint32_t)(r * N + n0 + (lane >> 3)) * 512u + 16u * (uint32_t)(lane & 7);
    sv.rowlim = pts - (lane >> 3);
    sv.lds = lds;
    const uint32_t blk = LDS_STAGE_P + (uint32_t)wave * SAVE_LDS_WAVE;
    sv.lds_w = blk + (uint32_t)(lane & 31) * SAVE_LDS_ROW + 16u * (uint32_t)h;
    sv.lds_r = blk + (uint32_t)(lane >> 3) * SAVE_LDS_ROW + 16u * (uint32_t)(lane & 7);
    return sv;
  }
};

static __device__ unsigned int crnerf_sched_bf16p[SCHED_SLOTS][2];   // kernels.h "Dynamic work distribution"
#ifdef CRNERF_TIMING
static __device__ unsigned long long crnerf_wg_times_p[2 * 1024];
#endif


// sum over the 32 lanes of this lane's half of v[idx], idx = p: after the five halving steps lane p holds the total of value p
__device__ __forceinline__ float fold32(float (&v)[32], int p) {
#define CRNERF_FOLD(W, BIT)                                         \
  {                                                                 \
    const bool up = (p & (BIT)) != 0;                               \
    _Pragma("unroll") for (int u = 0; u < (W) / 2; ++u) {          \
      const float lo = v[u], hi = v[u + (W) / 2];                   \
      v[u] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, (BIT));      \
    }                                                               \
  }
  CRNERF_FOLD(32, 16) CRNERF_FOLD(16, 8) CRNERF_FOLD(8, 4) CRNERF_FOLD(4, 2) CRNERF_FOLD(2, 1)
#undef CRNERF_FOLD
  return v[0];
}

// FINE_ONLY (crnerf_render_rays_bf16_fine, precision "bf16_hc"): the coarse pass has been rendered by another core (fp32-accurate, so that the fine
// depths are the fp32 reference's); its weights come in through a.wc_in, the kernel runs sample_pdf + merge on them and the FINE pass only --
// packed0 == packed1 == the fine model, the ring starts on model 1 and never leaves it.
template <class HOOK, bool FINE_ONLY = false>
__device__ __forceinline__ void render_rays_bf16p_body(const RenderParamsP& a, const HOOK& hook) {
#ifdef CRNERF_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 1024) crnerf_wg_times_p[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int pair = wave >> 1, half = wave & 1;
  const int lane128 = half * 64 + lane;
  const int Nc = a.Nc, Ni = a.Ni, Nf = Nc + Ni;

  PhaseTimer tm;
  tm.start(blockIdx.x == 0 && threadIdx.x == 0);
  load_consts(lds, a.packed0, a.packed1);
  PairScratch scr;
  scr.bind(lds + LDS_SCRATCH_P + pair * PAIR_BYTES);
  lds_char* dirbuf = lds + LDS_DIR_P + wave * 64;

  WeightPipeP pipe;
  pipe.start(lds, a.packed0 + CONST_BYTES, a.packed1 + CONST_BYTES, FINE_ONLY ? 1 : 0, lane, wave, HOOK::on);
  u32x4 q[B_AHEAD];
  pipe.prime(q);
  tm.tick(T_RAYLEVEL);

  __attribute__((address_space(3))) unsigned int* qslot = (__attribute__((address_space(3))) unsigned int*)(lds + LDS_QSLOT_P);
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
      // rendering.py:155  dir_embedded = embedding_dir(kwargs.get('view_dir', rays_d)): a per-ray constant, parked in LDS
      const float* vd = a.view_dir ? a.view_dir + r * 3 : ray + 3;
      u32x4 tmp[KS_DIR];
      posenc_b<DIR_FREQS, KS_DIR>(vd[0], vd[1], vd[2], h, tmp);
      if (p == 0) {
#pragma unroll
        for (int s = 0; s < KS_DIR; ++s) *(__attribute__((address_space(3))) u32x4*)(dirbuf + 32 * s + 16 * h) = tmp[s];
      }
    }
    for (int n = lane128; n < Nc; n += 128)
      scr.zc[n] = a.z_coarse ? a.z_coarse[r * Nc + n] : coarse_depth(near, far, a.z_steps ? a.z_steps[n] : linspace01(n, Nc), a.use_disp);
    if constexpr (FINE_ONLY) {
      for (int n = lane128; n < Nc; n += 128) scr.wc[n] = a.wc_in[r * Nc + n];
    }
    wg_barrier();

    const int npass = Ni > 0 ? 2 : 1;
#pragma unroll 1
    for (int pass = FINE_ONLY ? 1 : 0; pass < npass; ++pass) {
      if constexpr (FINE_ONLY) {   // what follows the coarse pass in the full kernel (below), on the weights that came in
        sample_pdf_pair(scr, Nc, Ni, a.u ? a.u + r * a.u_stride : nullptr, lane, lane128);
        wg_barrier();
        merge_sort_pair(scr, Nc, Ni, lane128);
        wg_barrier();
        if (a.z_fine && ray_ok)
          for (int n = lane128; n < Nf; n += 128) a.z_fine[r * Nf + n] = scr.zs[n];
      }
      const int N = pass ? Nf : Nc;
      const lds_float* zsrc = pass ? scr.zs : scr.zc;
      const float* noise_row = pass ? (a.noise_f ? a.noise_f + r * Nf : nullptr) : (a.noise_c ? a.noise_c + r * Nc : nullptr);
      float* weights_row = pass ? a.weights_f + r * Nf : a.weights_c + r * Nc;
      double carry = 1.0;
      float fsum = 0.0f, dacc = 0.0f;
      const int steps = (N + 63) >> 6;
#pragma unroll 1
      for (int k = 0; k < steps; ++k) {
        const int n = 64 * k + 32 * half + p;
        const bool valid = n < N;
        const int nc = valid ? n : N - 1;
        const float zn = zsrc[nc];
        const float znext = zsrc[nc + 1 < N ? nc + 1 : N - 1];
        const float x = ox + dx * zn, y = oy + dy * zn, z = oz + dz * zn;   // rendering.py:178 / :188 (separate mul and add)
        u32x4 pe[KS_XYZ];
        posenc_b<XYZ_FREQS, KS_XYZ>(x, y, z, h, pe);
        tm.tick(T_X0);
        f32x16 feat[2];
        float sigma;
        // the model of the tile after this one: same pass, the fine pass, or the next ray's coarse pass
        const int next_model = FINE_ONLY ? 1 : (k + 1 < steps ? pass : (pass + 1 < npass ? 1 : 0));
        auto sv = hook.saver(pass, r, N, 64 * k + 32 * half, n, ray_ok, lane, wave, lds);
        if constexpr (HOOK::on) {   // the embedded input as the MLP multiplies it: xyz k-steps 0..5, dir k-steps 6, 7 (32 B per point and k-step);
          const __amdgpu_buffer_rsrc_t xr = sv.xb();                        // unconditional stores, counted in SAVE_TILE_BURST (mlp_core_bf16p.h)
          const uint32_t xo = sv.boff == SAVE_OOB ? SAVE_OOB : (uint32_t)(r * N + n) * 256u + 16u * (uint32_t)h;
#pragma unroll
          for (int s = 0; s < KS_XYZ; ++s) __builtin_amdgcn_raw_buffer_store_b128(pe[s], xr, (int)(xo + 32u * s), 0, 0);
#pragma unroll
          for (int s = 0; s < KS_DIR; ++s)
            __builtin_amdgcn_raw_buffer_store_b128(*(const __attribute__((address_space(3))) u32x4*)(dirbuf + 16 * h + 32 * s), xr, (int)(xo + 32u * (KS_XYZ + s)), 0, 0);
        }
        mlp_tile_p(pipe, pass, next_model, pe, dirbuf + 16 * h, feat, sigma, h, q, tm, sv);
        if constexpr (HOOK::on) {   // raw MLP output row (rgb features after the sigmoid, sigma after the softplus): what compositing consumes.
          // 8 x 16 B + sigma, unconditional (SAVE_TILE_BURST); rows are 260 B apart: dword-aligned 16-byte stores
          const long P = hook.R * N;
          const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(pass ? hook.rawo[1] : hook.rawo[0], 0, (int)(uint32_t)(P * (OUT_DIM * 4)), SAVE_FLAGS);
          const uint32_t ro = (valid && ray_ok && hook.stores_on()) ? (uint32_t)(r * N + n) * (uint32_t)(OUT_DIM * 4) + 16u * (uint32_t)h : SAVE_OOB;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const u32x4 v = {__float_as_uint(feat[t][4 * c]), __float_as_uint(feat[t][4 * c + 1]), __float_as_uint(feat[t][4 * c + 2]), __float_as_uint(feat[t][4 * c + 3])};
              __builtin_amdgcn_raw_buffer_store_b128(v, rr, (int)(ro + 4u * (32 * t + 8 * c)), 0, 0);
            }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sigma), rr, (int)(h == 0 && ro != SAVE_OOB ? ro + 4u * FEAT_DIM : SAVE_OOB), 0, 0);
        }
        // ---- compositing, rendering.py:121-143 (both lane halves evaluate the same 32 samples)
        const float noise = (noise_row && valid) ? noise_row[n] * a.noise_std : 0.0f;
        const float delta = (n == N - 1) ? 1e2f : znext - zn;
        const float alpha = valid ? 1.0f - expf(-delta * fmaxf(sigma + noise, 0.0f)) : 0.0f;
        // inclusive prefix product over the 32 lanes of each half on the DPP network (ray_ops.h composite_tile64)
        double incl = (double)(1.0f - alpha);
        incl *= dpp_f64<0x111, 0xf>(incl);   // row_shr:1
        incl *= dpp_f64<0x112, 0xf>(incl);   // row_shr:2
        incl *= dpp_f64<0x114, 0xf>(incl);   // row_shr:4
        incl *= dpp_f64<0x118, 0xf>(incl);   // row_shr:8
        incl *= dpp_f64<0x142, 0xa>(incl);   // row_bcast:15 -> rows 1, 3
        double excl = dpp_f64<0x138, 0xf>(incl);   // wave_shr:1
        if (p == 0) excl = 1.0;
        const double tot = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(incl), 31), __builtin_amdgcn_readlane(__double2loint(incl), 31));
        if (lane == 0) scr.xprod[half] = tot;
        wg_barrier();
        const double other = scr.xprod[half ^ 1];
        const double prod_a = half ? other : tot, prod_b = half ? tot : other;
        const float Tr = (float)(half ? carry * prod_a * excl : carry * excl);
        carry = carry * prod_a * prod_b;
        const float w = alpha * Tr;
        float v[32];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[16 * t + e] = w * feat[t][e];
        fsum += fold32(v, p);
        dacc += w * zn;
        if (valid && h == 0) {
          if (ray_ok) weights_row[n] = w;
          if (pass == 0) scr.wc[n] = w;
        }
        tm.tick(T_COMPOSITE);
      }
      // lane (p, h) holds the wave's sum of feature 32(p>>4) + 8((p&15)>>2) + 4h + (p&3); pair combine through LDS
      const int fidx = 32 * (p >> 4) + 8 * ((p & 15) >> 2) + 4 * h + (p & 3);
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) dacc += __shfl_xor(dacc, d);
      if (half == 1) {
        scr.xfeat[fidx] = fsum;
        if (lane == 0) scr.xfeat[64] = dacc;
      }
      wg_barrier();
      if (half == 0 && ray_ok) {
        ((pass ? a.feature_f : a.feature_c) + r * FEAT_DIM)[fidx] = fsum + scr.xfeat[fidx];
        if (lane == 0) (pass ? a.depth_f : a.depth_c)[r] = dacc + scr.xfeat[64];
      }
      tm.tick(T_X5);
      if (pass == 0 && Ni > 0) {
        sample_pdf_pair(scr, Nc, Ni, a.u ? a.u + r * a.u_stride : nullptr, lane, lane128);
        wg_barrier();
        tm.tick(T_X6);
        merge_sort_pair(scr, Nc, Ni, lane128);
        wg_barrier();
        tm.tick(T_X7);
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
    atomicExch(a.sched, 0u);
    atomicExch(a.sched + 1, 0u);
  }
  tm.flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CRNERF_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 1024) crnerf_wg_times_p[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

__global__ __launch_bounds__(512, 2) void render_rays_bf16p_kernel(RenderParamsP a) { render_rays_bf16p_body(a, NoHookP()); }
__global__ __launch_bounds__(512, 2) void render_rays_train_bf16p_kernel(RenderParamsP a, TrainHookP hook) { render_rays_bf16p_body(a, hook); }
__global__ __launch_bounds__(512, 2) void render_rays_bf16p_fine_kernel(RenderParamsP a) { render_rays_bf16p_body<NoHookP, true>(a, NoHookP()); }

int launch_render_rays_bf16p(const RenderArgs& a, hipStream_t stream) {
  if (a.R <= 0) return 0;
  if (a.Nc < 2 || a.Nc > MAX_NC) return set_error(-2, "render_rays_bf16: N_samples must be in [2, 256] for the fused kernel");
  if (a.Ni < 0 || a.Ni > MAX_NI) return set_error(-2, "render_rays_bf16: N_importance must be in [0, 256] for the fused kernel");
  if (a.Ni > 0 && a.Nc < 3) return set_error(-2, "render_rays_bf16: hierarchical sampling needs N_samples >= 3");
  if (a.Ni > 0 && !a.packed_fine) return set_error(-3, "render_rays_bf16: N_importance > 0 but no fine model");
  RenderParamsP k;
  k.packed0 = (const char*)a.packed_coarse;
  k.packed1 = (const char*)(a.packed_fine ? a.packed_fine : a.packed_coarse);
  k.rays = a.rays; k.view_dir = a.view_dir; k.z_coarse = a.z_coarse; k.z_steps = a.z_steps; k.u = a.u; k.u_stride = a.u_stride;
  k.noise_c = a.noise_coarse; k.noise_f = a.noise_fine; k.noise_std = a.noise_std; k.use_disp = a.use_disp;
  k.R = a.R; k.Nc = a.Nc; k.Ni = a.Ni;
  k.weights_c = a.weights_coarse; k.feature_c = a.feature_coarse; k.depth_c = a.depth_coarse;
  k.weights_f = a.weights_fine; k.feature_f = a.feature_fine; k.depth_f = a.depth_fine; k.z_fine = a.z_fine;
  k.wc_in = nullptr;
  if (a.fine_only) {   // crnerf_render_rays_bf16_fine: weights_coarse is an INPUT, the fine model runs alone
    if (a.Ni <= 0 || !a.packed_fine || !a.weights_coarse) return set_error(-3, "render_rays_bf16_fine: needs N_importance > 0, the fine model and weights_coarse");
    if (a.train_acts_coarse) return set_error(-3, "render_rays_bf16_fine: no training twin");
    k.packed0 = k.packed1 = (const char*)a.packed_fine;
    k.wc_in = a.weights_coarse;
    k.weights_c = k.feature_c = k.depth_c = nullptr;
  }
  const long quads = (a.R + 3) / 4;
  const int cus = num_cus();
  const int grid = (int)(quads < cus ? quads : cus);   // one workgroup per CU, persistent over ray quads
  k.iters = (int)((quads + grid - 1) / grid);
  k.sched = k.iters > 1 ? sched_slot((const void*)crnerf_sched_bf16p) : nullptr;
  if (a.train_acts_coarse) {   // training twin (crnerf_render_rays_train_bf16)
    if (a.Ni > 0 && (!a.train_acts_fine || !a.train_raw_fine)) return set_error(-1, "render_rays_train_bf16: fine buffers are NULL");
    if (!a.train_raw_coarse) return set_error(-1, "render_rays_train_bf16: raw_coarse is NULL");
    if ((unsigned long long)a.R * (unsigned)(a.Nc + a.Ni) * 512ull >= (unsigned long long)SAVE_OOB)
      return set_error(-2, "render_rays_train_bf16: more than 7.8 M sample points per pass and call (the saved rows are addressed with 32-bit offsets)");
    TrainHookP h{{(char*)a.train_acts_coarse, (char*)a.train_acts_fine}, {a.train_raw_coarse, a.train_raw_fine}, a.R};
    if (int rc = ensure_dynamic_lds((const void*)render_rays_train_bf16p_kernel, LDS_TOTAL_TRAIN_P, "render_rays_train_bf16p_kernel")) return rc;
    hipLaunchKernelGGL(render_rays_train_bf16p_kernel, dim3(grid), dim3(512), LDS_TOTAL_TRAIN_P, stream, k, h);
    return check_launch("render_rays_train_bf16p_kernel");
  }
  if (a.fine_only) {
    if (int rc = ensure_dynamic_lds((const void*)render_rays_bf16p_fine_kernel, LDS_TOTAL_P, "render_rays_bf16p_fine_kernel")) return rc;
    hipLaunchKernelGGL(render_rays_bf16p_fine_kernel, dim3(grid), dim3(512), LDS_TOTAL_P, stream, k);
    return check_launch("render_rays_bf16p_fine_kernel");
  }
  if (int rc = ensure_dynamic_lds((const void*)render_rays_bf16p_kernel, LDS_TOTAL_P, "render_rays_bf16p_kernel")) return rc;
  hipLaunchKernelGGL(render_rays_bf16p_kernel, dim3(grid), dim3(512), LDS_TOTAL_P, stream, k);
  return check_launch("render_rays_bf16p_kernel");
}

#ifdef CRNERF_TIMING
extern "C" int crnerf_debug_read_wgtimes_bf16p(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crnerf_wg_times_p), sizeof(unsigned long long) * 2 * 1024);
}
extern "C" int crnerf_debug_read_timing_bf16p(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crnerf_timing), sizeof(unsigned long long) * T_COUNT);
}
#endif

}  // namespace crnerf
