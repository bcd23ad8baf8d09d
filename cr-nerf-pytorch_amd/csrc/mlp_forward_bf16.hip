// Stand-alone NeRF_sigma forward on the bf16 matrix cores: x[P,120] fp32 (already embedded) -> out[P,65] fp32.
// Module-level entry (NeRF_sigma.__call__, models/nerf.py:157-182) in the mixed-precision semantics of
// mlp_core_bf16.h, and the unit under test for that core; the production path is render_fused_bf16.hip.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core_bf16.h"

namespace crnerf {

// dword pp of k-step s, lane half h = bf16 pair of padded slots 16s + 8h + 2pp (+1)
template <int F, int NS>
__device__ __forceinline__ void gather_embedded_b(const float* __restrict__ row, int h, bool valid, u32x4 (&dst)[NS][2], int g) {
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      float v[2];
#pragma unroll
      for (int sc = 0; sc < 2; ++sc) {
        const int c0 = posenc_slot_to_col_b(16 * s + 2 * pp + sc, F), c1 = posenc_slot_to_col_b(16 * s + 8 + 2 * pp + sc, F);
        const int c = h ? c1 : c0;
        const float t = row[c < 0 ? 0 : c];          // unconditional load (row is clamped), then select: no branches
        v[sc] = (valid && c >= 0) ? t : 0.0f;
      }
      dst[s][g][pp] = pk_bf16(v[0], v[1]);
    }
}

__global__ __launch_bounds__(256, 1) void mlp_forward_bf16_kernel(const char* __restrict__ packed, const float* __restrict__ x,
                                                                  float* __restrict__ out, int sigma_only, long P, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;

  load_consts(lds, packed, packed);
  WeightPipeB pipe;
  pipe.start(lds, packed + CONST_BYTES, packed + CONST_BYTES, 0, lane, wave);
  u32x4 q[B_AHEAD];
  pipe.prime(q);
  PhaseTimer tm;
  tm.start(false);

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * 4 + wave;
    const int xdim = sigma_only ? XYZ_DIM : IN_DIM;
    u32x4 pe[KS_XYZ][2], dv[KS_DIR][2];
    int hg = h;
    asm volatile("" : "+v"(hg));   // keep the 240 half-dependent column selects inside the loop (else hoisted and spilled)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const long n = tile * 64 + 32 * g + p;
      const bool valid = n < P;
      const float* row = x + (valid ? n : 0) * xdim;
      gather_embedded_b<XYZ_FREQS, KS_XYZ>(row, hg, valid, pe, g);
      gather_embedded_b<DIR_FREQS, KS_DIR>(row + XYZ_DIM, hg, valid && !sigma_only, dv, g);
    }
    f32x16 feat[2][2];
    float sigma[2];
    mlp_tile_b(pipe, 0, 0, pe, dv, feat, sigma, h, q, tm);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const long n = tile * 64 + 32 * g + p;
      if (n < P) {
        if (sigma_only) {
          if (h == 0) out[n] = sigma[g];
        } else {
          float* o = out + n * OUT_DIM;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[32 * t + 8 * (r >> 2) + 4 * h + (r & 3)] = feat[g][t][r];
          if (h == 0) o[FEAT_DIM] = sigma[g];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_mlp_forward_bf16(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream) {
  if (P <= 0) return 0;
  const long groups = (P + 255) / 256;  // 256 points per workgroup-iteration
  const int cus = num_cus();
  const int grid = (int)(groups < cus ? groups : cus);
  const int iters = (int)((groups + grid - 1) / grid);
  const size_t shmem = LDS_SCRATCH;
  if (int rc = ensure_dynamic_lds((const void*)mlp_forward_bf16_kernel, shmem, "mlp_forward_bf16_kernel")) return rc;
  hipLaunchKernelGGL(mlp_forward_bf16_kernel, dim3(grid), dim3(256), shmem, stream, (const char*)packed, x, out, sigma_only, P, iters);
  return check_launch("mlp_forward_bf16_kernel");
}

}  // namespace crnerf
