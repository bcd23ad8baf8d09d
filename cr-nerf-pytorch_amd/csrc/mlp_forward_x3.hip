// Stand-alone NeRF_sigma forward on the x3 core (mlp_core_x3.h): x[P,120] fp32 (already embedded) -> out[P,65] fp32, fp32-accurate products
// on the bf16 matrix cores.  Module-level entry (NeRF_sigma.__call__, models/nerf.py:157-182) and the unit under test for that core.
#include <hip/hip_runtime.h>
#include "kernels.h"
#if defined(CRNERF_X_NP) && CRNERF_X_NP == 2
#include "mlp_core_h2t.h"
#else
#include "mlp_core_x3.h"
#endif

namespace crnerf {

// B-operand register rho of half h holds padded slot 8(rho/4) + 4h + rho%4 (layout.h posenc_slot_to_col)
template <int F, int NREG>
__device__ __forceinline__ void gather_embedded_x3(const float* __restrict__ row, int h, bool valid, float* dst) {
#pragma unroll
  for (int rho = 0; rho < NREG; ++rho) {
    const int k0 = 8 * (rho / 4) + (rho % 4);
    const int c0 = posenc_slot_to_col(k0, F), c1 = posenc_slot_to_col(k0 + 4, F);
    const int c = h ? c1 : c0;
    dst[rho] = (valid && c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
  }
}

__global__ __launch_bounds__(256, 1) void mlp_forward_x3_kernel(const char* __restrict__ packed, const float* __restrict__ x, float* __restrict__ out,
                                                                int sigma_only, long P, int iters, int repair) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;

  if (repair) {   // crnerf_mlp_forward_f32x3_repair: one workgroup per 128 points; it leaves at once unless a previous call left a NaN sigma among them
    const long n = ((long)blockIdx.x * 4 + wave) * 32 + p;
    const bool bad = n < P && (__float_as_uint(sigma_only ? out[n] : out[n * OUT_DIM + FEAT_DIM]) & 0x7fffffffu) > 0x7f800000u;
    if (!__syncthreads_or(bad)) return;
  }
  load_consts(lds, packed, packed);
  WeightPipeX pipe;
  pipe.start(lds, packed + CONST_BYTES, packed + CONST_BYTES, 1, 1, lane, wave);
  xu32x4 q[X_AHEAD];
  pipe.prime(q);
  PhaseTimer tm;
  tm.start(false);

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * 4 + wave;
    const long n = tile * 32 + p;
    const bool valid = n < P;
    const int xdim = sigma_only ? XYZ_DIM : IN_DIM;
    const float* row = x + (valid ? n : 0) * xdim;
    f32x16 pe[3], dv[1], feat[2];
    float sigma;
    {
      float tmp[48];
      gather_embedded_x3<XYZ_FREQS, 48>(row, h, valid, tmp);
#pragma unroll
      for (int i = 0; i < 48; ++i) pe[i / 16][i % 16] = tmp[i];
      float tmpd[16];
      gather_embedded_x3<DIR_FREQS, 16>(row + XYZ_DIM, h, valid && !sigma_only, tmpd);
#pragma unroll
      for (int i = 0; i < 16; ++i) dv[0][i] = tmpd[i];
    }
    mlp_tile_x3(pipe, 0, pe, dv, feat, sigma, h, q, tm);
    if (valid) {
      if (sigma_only) {
        if (h == 0) out[n] = sigma;
      } else {
        float* o = out + n * OUT_DIM;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[32 * t + 8 * (r >> 2) + 4 * h + (r & 3)] = feat[t][r];
        if (h == 0) o[FEAT_DIM] = sigma;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the prefetches still in flight before the workgroup (and its LDS) goes away
}

int launch_mlp_forward_x3(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream, int repair) {
  if (P <= 0) return 0;
  const long groups = (P + 127) / 128;   // 128 points per workgroup-iteration
  const int cus = num_cus();
  if (repair && groups > 0x7fffffffL) return set_error(-2, "mlp_forward_f32x3_repair: too many points");
  const int grid = repair ? (int)groups : (int)(groups < cus ? groups : cus);
  const int iters = (int)((groups + grid - 1) / grid);
  const size_t shmem = LDS_SCRATCH_X;
  if (int rc = ensure_dynamic_lds((const void*)mlp_forward_x3_kernel, shmem, "mlp_forward_x3_kernel")) return rc;
  hipLaunchKernelGGL(mlp_forward_x3_kernel, dim3(grid), dim3(256), shmem, stream, (const char*)packed, x, out, sigma_only, P, iters, repair);
  return check_launch("mlp_forward_x3_kernel");
}

}  // namespace crnerf
