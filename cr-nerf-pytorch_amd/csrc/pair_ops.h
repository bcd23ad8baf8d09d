// Per-ray operations for kernels that give one ray to a PAIR of wavefronts (render_fused16.hip, render_fused_bf16p.hip):
// the pair's LDS scratch, the LDS-only workgroup barrier, sample_pdf and the z merge on the pair's 128 lanes.
// Reference: models/rendering.py:7-46 (sample_pdf), :187 (sort(cat)).
#pragma once
#include <hip/hip_runtime.h>
#include "philox.h"
#include "ray_ops.h"

namespace crnerf {

constexpr int PAIR_FLOATS = 4 * MAX_NC + (MAX_NC + MAX_NI) + 128;   // zc, wc, cdf, zf(<=MAX_NI==MAX_NC), zs, exchange
constexpr int PAIR_BYTES = PAIR_FLOATS * 4;
static_assert(MAX_NI <= MAX_NC, "zf shares the MAX_NC sizing");

struct PairScratch {
  lds_float *zc, *wc, *cdf, *zf, *zs, *xfeat;   // xfeat[64] + depth at [64]
  __attribute__((address_space(3))) double* xprod;  // [2]
  __device__ __forceinline__ void bind(lds_char* b) {
    zc = (lds_float*)b; wc = zc + MAX_NC; cdf = wc + MAX_NC; zf = cdf + MAX_NC; zs = zf + MAX_NC;
    xfeat = zs + (MAX_NC + MAX_NI);
    xprod = (__attribute__((address_space(3))) double*)(xfeat + 96);
  }
};

// LDS-only exchange between the waves of the workgroup: drain this wave's LDS queue, barrier.  (A
// workgroup-scope fence would also drain vmcnt, i.e. the LDS-DMA prefetches in flight.)
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// sample_pdf on the pair's 128 lanes (rendering.py:7-46).  Both waves build the cdf redundantly
// (identical values) so only the sample / merge phases need the partner.
// rng != null: the uniforms are drawn in-kernel (philox.h, stream RNG_STREAM_U, ray index rng_ray) instead of read from u_row.
__device__ __forceinline__ void sample_pdf_pair(PairScratch& s, int Nc, int Ni, const float* u_row, int lane, int lane128, const RayRng* rng = nullptr,
                                                long rng_ray = 0) {
  const int n_ = Nc - 2;
  const float eps = 1e-5f;
  float part = 0.0f;
  for (int i = lane; i < n_; i += 64) part += s.wc[1 + i] + eps;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  const float wsum = part;
  double carry = 0.0;
  for (int base = 0; base < n_; base += 64) {
    const int i = base + lane;
    const float pdf = (i < n_) ? (s.wc[1 + i] + eps) / wsum : 0.0f;
    double incl = (double)pdf;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double o = shfl_up_f64(incl, d, 64);
      if (lane >= d) incl += o;
    }
    incl += carry;
    carry = shfl_f64(incl, 63, 64);
    if (i < n_) s.cdf[i + 1] = (float)incl;
  }
  if (lane == 0) s.cdf[0] = 0.0f;
  wave_lds_fence();
  for (int k = lane128; k < Ni; k += 128) {
    const float u = rng ? rng->uniform(RNG_STREAM_U, rng_ray, k) : (u_row ? u_row[k] : linspace01(k, Ni));
    int lo = 0, hi = n_ + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s.cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 < 0 ? 0 : lo - 1;
    const int above = lo > n_ ? n_ : lo;
    const float c0 = s.cdf[below], c1 = s.cdf[above];
    const float b0 = 0.5f * (s.zc[below] + s.zc[below + 1]);
    const float b1 = 0.5f * (s.zc[above] + s.zc[above + 1]);
    float denom = c1 - c0;
    if (denom < eps) denom = 1.0f;
    s.zf[k] = b0 + (u - c0) / denom * (b1 - b0);
  }
}

__device__ __forceinline__ void merge_sort_pair(PairScratch& s, int Nc, int Ni, int lane128) {
  const int N = Nc + Ni;
  // both inputs ascending (always, unless the caller supplies unsorted depths / uniforms): two-way merge by binary
  // search (ray_ops.h merge_sort_wave); each wave of the pair checks the whole arrays, so both take the same branch
  const bool sorted = wave_ascending(s.zc, Nc, lane128 & 63) && wave_ascending(s.zf, Ni, lane128 & 63);
  for (int e = lane128; e < N; e += 128) {
    const bool is_c = e < Nc;
    const float v = is_c ? s.zc[e] : s.zf[e - Nc];
    int rank;
    if (sorted)
      rank = is_c ? e + bound_lds<true>(s.zf, Ni, v) : (e - Nc) + bound_lds<false>(s.zc, Nc, v);
    else
      rank = count_before(s.zc, Nc, v, is_c ? e : Nc) + count_before(s.zf, Ni, v, is_c ? 0 : e - Nc);
    s.zs[rank] = v;
  }
}

}  // namespace crnerf
