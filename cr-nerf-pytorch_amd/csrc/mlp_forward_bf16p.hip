// Stand-alone NeRF_sigma forward on the bf16 PAIR core (mlp_core_bf16p.h: 32-point tiles, two waves per SIMD): x[P,120] fp32 (already embedded)
// -> out[P,65] fp32, in the mixed-precision semantics of include/crnerf.h "bf16".  Module-level entry (NeRF_sigma.__call__, models/nerf.py:157-182);
// round 4: crnerf_mlp_forward_bf16 moved here from the round-1/2 core (mlp_forward_bf16.hip; removed in round 5) --
// the two cores' outputs are bit-identical (tests/test_gpu_bf16.py).  The production path is the fused renderer, render_fused_bf16p.hip.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core_bf16p.h"

namespace crnerf {

constexpr int LDS_DIR_M = LDS_SCRATCH_P;                    // 8 waves x 64 lanes x 64 B: every POINT's direction embedding as B operands (dword 32 s)
constexpr int LDS_TOTAL_M = LDS_DIR_M + P_WAVES * 64 * 64;
static_assert(LDS_TOTAL_M <= 160 * 1024, "LDS budget");

// dword pp of k-step s, lane half h = bf16 pair of padded slots 16s + 8h + 2pp (+1)   (gather_embedded_b of mlp_forward_bf16.hip, one 32-point group)
template <int F, int NS>
__device__ __forceinline__ void gather_embedded_p(const float* __restrict__ row, int h, bool valid, u32x4 (&dst)[NS]) {
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
      dst[s][pp] = pk_bf16(v[0], v[1]);
    }
}

__global__ __launch_bounds__(512, 2) void mlp_forward_bf16p_kernel(const char* __restrict__ packed, const float* __restrict__ x, float* __restrict__ out,
                                                                   int sigma_only, long P, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;

  load_consts(lds, packed, packed);
  lds_char* dirbuf = lds + LDS_DIR_M + wave * (64 * 64) + lane * 64;   // this lane's own 2 x 16 bytes (the fused renderer parks ONE embedding per ray)
  WeightPipeP pipe;
  pipe.start(lds, packed + CONST_BYTES, packed + CONST_BYTES, 0, lane, wave);
  u32x4 q[B_AHEAD];
  pipe.prime(q);
  PhaseTimer tm;
  tm.start(false);
  NoSaveP sv;

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * P_WAVES + wave;
    const long n = tile * 32 + p;
    const bool valid = n < P;
    const int xdim = sigma_only ? XYZ_DIM : IN_DIM;
    const float* row = x + (valid ? n : 0) * xdim;
    int hg = h;
    asm volatile("" : "+v"(hg));   // keep the half-dependent column selects inside the loop (else hoisted and spilled)
    u32x4 pe[KS_XYZ], dv[KS_DIR];
    gather_embedded_p<XYZ_FREQS, KS_XYZ>(row, hg, valid, pe);
    gather_embedded_p<DIR_FREQS, KS_DIR>(row + XYZ_DIM, hg, valid && !sigma_only, dv);
#pragma unroll
    for (int s = 0; s < KS_DIR; ++s) *(__attribute__((address_space(3))) u32x4*)(dirbuf + 32 * s) = dv[s];
    f32x16 feat[2];
    float sigma;
    mlp_tile_p(pipe, 0, 0, pe, dirbuf, feat, sigma, h, q, tm, sv);   // (reads its lane's dirbuf after its own writes: same wave, in order)
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_mlp_forward_bf16p(const void* packed, const float* x, float* out, long P, int sigma_only, hipStream_t stream) {
  if (P <= 0) return 0;
  const long groups = (P + 255) / 256;  // 256 points per workgroup-iteration (eight 32-point tiles)
  const int cus = num_cus();
  const int grid = (int)(groups < cus ? groups : cus);
  const int iters = (int)((groups + grid - 1) / grid);
  if (int rc = ensure_dynamic_lds((const void*)mlp_forward_bf16p_kernel, LDS_TOTAL_M, "mlp_forward_bf16p_kernel")) return rc;
  hipLaunchKernelGGL(mlp_forward_bf16p_kernel, dim3(grid), dim3(512), LDS_TOTAL_M, stream, (const char*)packed, x, out, sigma_only, P, iters);
  return check_launch("mlp_forward_bf16p_kernel");
}

}  // namespace crnerf
