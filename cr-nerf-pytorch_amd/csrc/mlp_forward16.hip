// Stand-alone NeRF_sigma forward on the v16 core: x[P,120] -> out[P,65] (module entry / unit under test).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core16.h"

namespace crnerf {

__global__ __launch_bounds__(512, 2) void mlp_forward16_kernel(const char* __restrict__ packed, const float* __restrict__ x,
                                                               float* __restrict__ out, int sigma_only, long P, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;

  load_consts(lds, packed, packed);
  WeightPipe16 pipe;
  pipe.start(lds, packed + CONST_BYTES, packed + CONST_BYTES, 1, 1, lane, wave);
  f32x4 q[V16_AHEAD];
  pipe.prime(q);
  PhaseTimer tm;
  tm.start(false);

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * V16_WAVES + wave;
    const long n = tile * 16 + p;
    const bool valid = n < P;
    const int xdim = sigma_only ? XYZ_DIM : IN_DIM;
    const float* row = x + (valid ? n : 0) * xdim;
    f32x4 pe[6], feat[4];
    DirRegs dreg;
    f32x4 (&dv)[2] = dreg.v;
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = posenc_slot_to_col16(16 * v + 4 * g + r, XYZ_FREQS);
        pe[v][r] = (valid && c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
      }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = posenc_slot_to_col16(16 * v + 4 * g + r, DIR_FREQS);
        dv[v][r] = (valid && !sigma_only && c >= 0) ? row[XYZ_DIM + (c < 0 ? 0 : c)] : 0.0f;
      }
    float sigma;
    mlp_tile16(pipe, 0, pe, dreg, feat, sigma, g, q, tm);
    if (valid) {
      if (sigma_only) {
        if (g == 0) out[n] = sigma;
      } else {
        float* o = out + n * OUT_DIM;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[16 * T + 4 * g + r] = feat[T][r];
        if (g == 0) o[FEAT_DIM] = sigma;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_mlp_forward16(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream) {
  if (P <= 0) return 0;
  const long groups = (P + 127) / 128;  // 128 points per workgroup-iteration (8 waves x 16)
  const int cus = num_cus();
  const int grid = (int)(groups < cus ? groups : cus);
  const int iters = (int)((groups + grid - 1) / grid);
  const size_t shmem = LDS_SCRATCH;
  if (int rc = ensure_dynamic_lds((const void*)mlp_forward16_kernel, shmem, "mlp_forward16_kernel")) return rc;
  hipLaunchKernelGGL(mlp_forward16_kernel, dim3(grid), dim3(512), shmem, stream, (const char*)packed, x, out, sigma_only, P, iters);
  return check_launch("mlp_forward16_kernel");
}

}  // namespace crnerf
