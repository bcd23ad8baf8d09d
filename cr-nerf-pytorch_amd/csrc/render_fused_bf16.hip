// Fused volumetric renderer on the bf16 matrix cores: the whole of render_rays_cross_ray
// (models/rendering.py:50-196) for one ray per wavefront in ONE launch, NeRF_sigma evaluated in the
// mixed-precision semantics of mlp_core_bf16.h (bf16 operands, fp32 accumulate); everything around the MLP --
// depths, xyz = o + d z, compositing, sample_pdf, the z merge -- is the fp32 code of the fp32 renderer (ray_ops.h).
//
// Work decomposition: workgroup = 4 waves = 4 rays in lockstep on the shared LDS weight ring; a ray's N samples
// are walked in 64-sample tiles (two 32-point groups) by its wave, carrying the running transmittance, so
// compositing needs no cross-wave traffic.  Grid-stride over ray quads.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core_bf16.h"
#include "ray_ops.h"

namespace crnerf {

struct RenderParamsB {
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
  unsigned int* sched;   // null: static quad -> workgroup map (iters passes); else {next-quad counter, finished-workgroup counter}
  float* weights_c;
  float* feature_c;
  float* depth_c;
  float* weights_f;
  float* feature_f;
  float* depth_f;
  float* z_fine;
};

static __device__ unsigned int crnerf_sched_bf16[SCHED_SLOTS][2];   // kernels.h "Dynamic work distribution"
#ifdef CRNERF_TIMING
static __device__ unsigned long long crnerf_wg_times[2 * 1024];   // per workgroup: s_memrealtime at kernel entry / exit (100 MHz), tools/wg_times.py
#endif
__global__ __launch_bounds__(256, 1) void render_rays_bf16_kernel(RenderParamsB a) {
#ifdef CRNERF_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 1024) crnerf_wg_times[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int Nc = a.Nc, Ni = a.Ni, Nf = Nc + Ni;

  PhaseTimer tm;
  tm.start(blockIdx.x == 0 && threadIdx.x == 0);
  load_consts(lds, a.packed0, a.packed1);
  RayScratch scr;
  scr.bind(lds + LDS_SCRATCH + wave * SCRATCH_BYTES);

  WeightPipeB pipe;
  pipe.start(lds, a.packed0 + CONST_BYTES, a.packed1 + CONST_BYTES, 0, lane, wave);
  u32x4 q[B_AHEAD];
  pipe.prime(q);
  tm.tick(T_RAYLEVEL);   // kernel prologue: constants into LDS, first three weight stages in flight, first fragments read

  // ray quads: statically strided over the grid, or (a.sched) pulled from a device counter -- the index for the NEXT pass is
  // requested at the top of a pass and picked up at its end, so the atomic's round trip is never waited for
  __attribute__((address_space(3))) unsigned int* qslot = (__attribute__((address_space(3))) unsigned int*)(lds + LDS_SCRATCH + 4 * SCRATCH_BYTES);
  const long quads = (a.R + 3) / 4;
  long quad = blockIdx.x;
#pragma unroll 1
  for (int it = 0; a.sched ? quad < quads : it < a.iters; ++it) {
    unsigned int nxt = 0;
    if (a.sched && threadIdx.x == 0) nxt = gridDim.x + atomicAdd(a.sched, 1u);
    const long rr = quad * 4 + wave;
    const bool ray_ok = rr < a.R;
    const long r = ray_ok ? rr : a.R - 1;
    const float* ray = a.rays + r * 8;
    const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
    const float near = ray[6], far = ray[7];
    u32x4 dv[KS_DIR][2];
    {
      // rendering.py:155  dir_embedded = embedding_dir(kwargs.get('view_dir', rays_d)); one ray per wave, so both
      // point groups share the operand
      const float* vd = a.view_dir ? a.view_dir + r * 3 : ray + 3;
      u32x4 tmp[KS_DIR];
      posenc_b<DIR_FREQS, KS_DIR>(vd[0], vd[1], vd[2], h, tmp);
#pragma unroll
      for (int s = 0; s < KS_DIR; ++s) dv[s][0] = dv[s][1] = tmp[s];
    }
    for (int n = lane; n < Nc; n += 64)
      scr.zc[n] = a.z_coarse ? a.z_coarse[r * Nc + n] : coarse_depth(near, far, a.z_steps ? a.z_steps[n] : linspace01(n, Nc), a.use_disp);
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
      const int tiles = (N + 63) >> 6;
#pragma unroll 1
      for (int tile = 0; tile < tiles; ++tile) {
        u32x4 pe[KS_XYZ][2];
        float zn[2], znext[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int n = tile * 64 + 32 * g + p;
          const int nc = n < N ? n : N - 1;
          zn[g] = zsrc[nc];
          znext[g] = zsrc[nc + 1 < N ? nc + 1 : N - 1];
          // rendering.py:178 / :188  xyz = rays_o + rays_d * z  (separate mul and add)
          const float x = ox + dx * zn[g], y = oy + dy * zn[g], z = oz + dz * zn[g];
          u32x4 tmp[KS_XYZ];
          posenc_b<XYZ_FREQS, KS_XYZ>(x, y, z, h, tmp);
#pragma unroll
          for (int s = 0; s < KS_XYZ; ++s) pe[s][g] = tmp[s];
        }
        f32x16 feat[2][2];
        float sigma[2];
        // the model of the tile after this one: same pass, the fine pass, or the next ray's coarse pass
        const int next_model = tile + 1 < tiles ? pass : (pass + 1 < npass ? 1 : 0);
        mlp_tile_b(pipe, pass, next_model, pe, dv, feat, sigma, h, q, tm);
        float noise[2];
        bool is_last[2], valid[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int n = tile * 64 + 32 * g + p;
          valid[g] = n < N;
          is_last[g] = n == N - 1;
          noise[g] = (noise_row && valid[g]) ? noise_row[n] * a.noise_std : 0.0f;
        }
        const Weights2 w = composite_tile64(st, feat, sigma, noise, zn, znext, is_last, valid, lane);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int n = tile * 64 + 32 * g + p;
          if (valid[g] && h == 0) {
            if (ray_ok) weights_row[n] = w.w[g];
            if (pass == 0) scr.wc[n] = w.w[g];
          }
        }
        tm.tick(T_COMPOSITE);
      }
      composite_finish(st);
      if (ray_ok)
        store_ray_feature(st, (pass ? a.feature_f : a.feature_c) + r * FEAT_DIM, (pass ? a.depth_f : a.depth_c) + r, p, h);
      tm.tick(T_X5);
      if (pass == 0 && Ni > 0) {
        wave_lds_fence();
        sample_pdf_wave(scr, Nc, Ni, a.u ? a.u + r * a.u_stride : nullptr, lane);
        tm.tick(T_X6);
        merge_sort_wave(scr, Nc, Ni, lane);
        tm.tick(T_X7);
        if (a.z_fine && ray_ok)
          for (int n = lane; n < Nf; n += 64) a.z_fine[r * Nf + n] = scr.zs[n];
      }
      tm.tick(T_RAYLEVEL);
    }
    if (a.sched) {
      if (threadIdx.x == 0) *qslot = nxt;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      quad = (long)__builtin_amdgcn_readfirstlane((int)*qslot);   // rewritten only at the end of the next pass, many barriers later
    } else {
      quad += gridDim.x;
    }
  }
  if (a.sched && threadIdx.x == 0 && atomicAdd(a.sched + 1, 1u) == gridDim.x - 1) {   // last workgroup out: leave the slot zeroed
    atomicExch(a.sched, 0u);       // device-scope, like the increments: the per-XCD L2s are not coherent for plain stores
    atomicExch(a.sched + 1, 0u);
  }
  tm.flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CRNERF_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 1024) crnerf_wg_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

int launch_render_rays_bf16(const RenderArgs& a, hipStream_t stream) {
  if (a.R <= 0) return 0;
  if (a.Nc < 2 || a.Nc > MAX_NC) return set_error(-2, "render_rays_bf16: N_samples must be in [2, 256] for the fused kernel");
  if (a.Ni < 0 || a.Ni > MAX_NI) return set_error(-2, "render_rays_bf16: N_importance must be in [0, 256] for the fused kernel");
  if (a.Ni > 0 && a.Nc < 3) return set_error(-2, "render_rays_bf16: hierarchical sampling needs N_samples >= 3");
  if (a.Ni > 0 && !a.packed_fine) return set_error(-3, "render_rays_bf16: N_importance > 0 but no fine model");
  RenderParamsB k;
  k.packed0 = (const char*)a.packed_coarse;
  k.packed1 = (const char*)(a.packed_fine ? a.packed_fine : a.packed_coarse);
  k.rays = a.rays; k.view_dir = a.view_dir; k.z_coarse = a.z_coarse; k.z_steps = a.z_steps; k.u = a.u; k.u_stride = a.u_stride;
  k.noise_c = a.noise_coarse; k.noise_f = a.noise_fine; k.noise_std = a.noise_std; k.use_disp = a.use_disp;
  k.R = a.R; k.Nc = a.Nc; k.Ni = a.Ni;
  k.weights_c = a.weights_coarse; k.feature_c = a.feature_coarse; k.depth_c = a.depth_coarse;
  k.weights_f = a.weights_fine; k.feature_f = a.feature_fine; k.depth_f = a.depth_fine; k.z_fine = a.z_fine;
  const long quads = (a.R + 3) / 4;
  const int cus = num_cus();
  const int grid = (int)(quads < cus ? quads : cus);   // one workgroup per CU, persistent over ray quads
  k.iters = (int)((quads + grid - 1) / grid);
  k.sched = k.iters > 1 ? sched_slot((const void*)crnerf_sched_bf16) : nullptr;
  const size_t shmem = LDS_SCRATCH + 4 * SCRATCH_BYTES + 16;
  if (int rc = ensure_dynamic_lds((const void*)render_rays_bf16_kernel, shmem, "render_rays_bf16_kernel")) return rc;
  hipLaunchKernelGGL(render_rays_bf16_kernel, dim3(grid), dim3(256), shmem, stream, k);
  return check_launch("render_rays_bf16_kernel");
}

#ifdef CRNERF_TIMING
extern "C" int crnerf_debug_read_wgtimes_bf16(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crnerf_wg_times), sizeof(unsigned long long) * 2 * 1024);
}
extern "C" int crnerf_debug_read_timing_bf16(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crnerf_timing), sizeof(unsigned long long) * T_COUNT);
}
#endif

}  // namespace crnerf
