// Per-ray (= per-wavefront) operations of the volumetric renderer: sample depths, alpha
// compositing with a wave-prefix transmittance, inverse-CDF hierarchical sampling and the z merge.
// One ray per wavefront; all per-ray arrays live in a wave-private LDS scratch.
//
// Reference: models/rendering.py
//   :161-178  coarse depths (linear / disparity)           -> coarse_depth()
//   :116-143  deltas, alpha, exclusive cumprod, weights    -> composite_tile() / wave reductions
//   :7-46     sample_pdf                                   -> sample_pdf_wave()
//   :187      sort(cat(z_coarse, z_fine))                  -> merge_sort_wave()
//
// Numerical notes (parity with the reference's CPU path): products that the reference evaluates
// as separate mul/add are kept un-fused (this TU is built with -ffp-contract=off); the two scans
// (cumsum of the pdf, cumprod of 1-alpha) run in fp64 like ATen's CPU cumsum/cumprod accumulators,
// then round to fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mlp_core.h"
#include "philox.h"

namespace crnerf {

constexpr int MAX_NC = 256;
constexpr int MAX_NI = 256;
constexpr int SCRATCH_FLOATS = MAX_NC + MAX_NC + MAX_NI + (MAX_NC + MAX_NI);  // zc, wc, zf, zs
constexpr int SCRATCH_BYTES = SCRATCH_FLOATS * 4;                             // 5 KiB per wave

struct RayScratch {
  lds_float* zc;   // [Nc] coarse depths
  lds_float* wc;   // [Nc] coarse weights, then reused as the cdf [Nc-1]
  lds_float* zf;   // [Ni] importance samples
  lds_float* zs;   // [Nc+Ni] merged, ascending
  // base 16-byte aligned, caps multiples of 4 (merge_sort_wave reads zc / zf as float4)
  __device__ __forceinline__ void bind(lds_char* base, int nc_cap = MAX_NC, int ni_cap = MAX_NI) {
    zc = (lds_float*)base;
    wc = zc + nc_cap;
    zf = wc + nc_cap;
    zs = zf + ni_cap;
  }
};

// wave-private LDS traffic only needs the LDS queue drained, not a workgroup barrier
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// torch.linspace(0, 1, n)[i] as ATen computes it (symmetric around the midpoint)
__device__ __forceinline__ float linspace01(int i, int n) {
  if (n <= 1) return 0.0f;
  const float step = 1.0f / (float)(n - 1);
  return (i < n / 2) ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// rendering.py:161-165
__device__ __forceinline__ float coarse_depth(float near, float far, float s, int use_disp) {
  const float t = 1.0f - s;
  if (!use_disp) return near * t + far * s;
  return 1.0f / ((1.0f / near) * t + (1.0f / far) * s);
}

__device__ __forceinline__ double shfl_up_f64(double v, int delta, int width) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_up(lo, delta, width);
  hi = __shfl_up(hi, delta, width);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src, int width) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl(lo, src, width);
  hi = __shfl(hi, src, width);
  return __hiloint2double(hi, lo);
}

// State of one compositing pass over a ray, distributed as lane (p = lane&31, h = lane>>5).
struct CompositeState {
  double t_carry;     // transmittance entering the current tile
  f32x16 facc[2];     // sum_n w_n * feat_n, this lane's point column, features 32t+8q+4h+j
  float dacc;         // sum_n w_n * z_n (lanes h == 0 only)
  __device__ __forceinline__ void reset() {
    t_carry = 1.0;
    dacc = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) facc[t][r] = 0.0f;
  }
};

// One 32-sample tile.  zn / znext: depth of this lane's sample and the following one; is_last:
// n == N-1 (delta = 1e2, rendering.py:122); valid: n < N.  Returns the weight w_n.
__device__ __forceinline__ float composite_tile(CompositeState& st, const f32x16 (&feat)[2], float sigma, float noise,
                                                float zn, float znext, bool is_last, bool valid, int p) {
  const float delta = is_last ? 1e2f : znext - zn;
  const float alpha = valid ? 1.0f - expf(-delta * fmaxf(sigma + noise, 0.0f)) : 0.0f;
  double incl = (double)(1.0f - alpha);  // inclusive prefix product over the 32 lanes of this half
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const double o = shfl_up_f64(incl, d, 32);
    if (p >= d) incl *= o;
  }
  double excl = shfl_up_f64(incl, 1, 32);
  if (p == 0) excl = 1.0;
  const float T = (float)(st.t_carry * excl);
  st.t_carry *= shfl_f64(incl, 31, 32);
  const float w = alpha * T;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) st.facc[t][r] += w * feat[t][r];
  st.dacc += w * zn;
  return w;
}

// One 64-sample tile held as two 32-point groups (lane (p, h) owns points p and 32 + p, bf16 core): ONE 64-lane scan
// instead of two 32-lane ones.  Lane 32h + p evaluates alpha of sample 32h + p (= its group-h point), the scan runs
// over lanes in sample order, and the two halves swap weights at the end.  sigma / noise / zn / znext / is_last /
// valid are indexed by group.  Returns this lane's two weights.
// source lane's value through the DPP network, 1.0 where the control selects no source lane / the row is masked out
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0x3ff00000, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

struct Weights2 { float w[2]; };
__device__ __forceinline__ Weights2 composite_tile64(CompositeState& st, const f32x16 (&feat)[2][2], const float (&sigma)[2],
                                                     const float (&noise)[2], const float (&zn)[2], const float (&znext)[2],
                                                     const bool (&is_last)[2], const bool (&valid)[2], int lane) {
  const int h = lane >> 5;
  const float sg = h ? sigma[1] : sigma[0], ns = h ? noise[1] : noise[0];
  const float z0 = h ? zn[1] : zn[0], z1 = h ? znext[1] : znext[0];
  const bool last = h ? is_last[1] : is_last[0], ok = h ? valid[1] : valid[0];
  const float delta = last ? 1e2f : z1 - z0;
  const float alpha = ok ? 1.0f - expf(-delta * fmaxf(sg + ns, 0.0f)) : 0.0f;
  // inclusive prefix product over the 64 lanes on the DPP network (no LDS crossbar round trips): four shifts inside the
  // 16-lane rows, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3.  A lane without a source
  // keeps the identity 1.0 it passes as `old`.
  double incl = (double)(1.0f - alpha);
  incl *= dpp_f64<0x111, 0xf>(incl);   // row_shr:1
  incl *= dpp_f64<0x112, 0xf>(incl);   // row_shr:2
  incl *= dpp_f64<0x114, 0xf>(incl);   // row_shr:4
  incl *= dpp_f64<0x118, 0xf>(incl);   // row_shr:8
  incl *= dpp_f64<0x142, 0xa>(incl);   // row_bcast:15 -> rows 1, 3
  incl *= dpp_f64<0x143, 0xc>(incl);   // row_bcast:31 -> rows 2, 3
  const double excl = dpp_f64<0x138, 0xf>(incl);   // wave_shr:1; lane 0 keeps 1.0
  const float T = (float)(st.t_carry * excl);
  st.t_carry *= __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(incl), 63), __builtin_amdgcn_readlane(__double2loint(incl), 63));
  const float mine = alpha * T;
  const float other = __shfl_xor(mine, 32);
  Weights2 r;
  r.w[0] = h ? other : mine;
  r.w[1] = h ? mine : other;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      st.facc[t][k] += r.w[0] * feat[0][t][k];
      st.facc[t][k] += r.w[1] * feat[1][t][k];
    }
  st.dacc += r.w[0] * zn[0];
  st.dacc += r.w[1] * zn[1];
  return r;
}

// Cross-lane reduction of the accumulators over the 32 point-lanes of each half; afterwards every
// lane holds the totals (features 32t+8q+4h+j of the ray in facc[t][4q+j]).
// sum over the 32 lanes of each wave half, result in every lane: four DPP steps inside a 16-lane row (quad xor 1,
// quad xor 2, half-row mirror, row mirror) and one ds_swizzle across the two rows -- no address registers and a
// fraction of the latency of five ds_bpermute round trips per value
__device__ __forceinline__ float half_wave_sum(float v) {
  auto dpp_add = [](float x, auto ctrl) __attribute__((always_inline)) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
}

__device__ __forceinline__ void composite_finish(CompositeState& st) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) st.facc[t][r] = half_wave_sum(st.facc[t][r]);
  st.dacc = half_wave_sum(st.dacc);
}

__device__ __forceinline__ void store_ray_feature(const CompositeState& st, float* feature_row, float* depth, int p, int h) {
  if (p == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {st.facc[t][4 * q + 0], st.facc[t][4 * q + 1], st.facc[t][4 * q + 2], st.facc[t][4 * q + 3]};
        *(f32x4*)(feature_row + 32 * t + 8 * q + 4 * h) = v;
      }
    if (h == 0) *depth = st.dacc;
  }
}

// sample_pdf(bins = z_mid, weights = w[1:-1], Ni, det) -- rendering.py:7-46, called at :183-184.
// In: s.zc[Nc], s.wc[Nc] (coarse weights).  Out: s.zf[Ni].  u_row: per-ray uniforms or null (det).
// rng != null: the uniforms are drawn in-kernel (philox.h, stream RNG_STREAM_U, ray index rng_ray) instead of read from u_row.
__device__ __forceinline__ void sample_pdf_wave(RayScratch& s, int Nc, int Ni, const float* u_row, int lane, const RayRng* rng = nullptr,
                                                long rng_ray = 0) {
  const int n_ = Nc - 2;       // number of pdf bins
  const float eps = 1e-5f;
  // sum of (w + eps) over the interior weights
  float part = 0.0f;
  for (int i = lane; i < n_; i += 64) part += s.wc[1 + i] + eps;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  const float wsum = part;
  // cdf[0] = 0, cdf[k] = float(sum_{i<k} double(pdf_i)); computed chunk-wise with an fp64 wave scan
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
    wave_lds_fence();  // all lanes have read wc[1+i] of this chunk before cdf overwrites wc[i+1]
    if (i < n_) s.wc[i + 1] = (float)incl;
  }
  if (lane == 0) s.wc[0] = 0.0f;
  wave_lds_fence();
  const lds_float* cdf = s.wc;  // [n_ + 1]
  for (int k = lane; k < Ni; k += 64) {
    const float u = rng ? rng->uniform(RNG_STREAM_U, rng_ray, k) : (u_row ? u_row[k] : linspace01(k, Ni));
    // searchsorted(cdf, u, right=True): number of entries <= u
    int lo = 0, hi = n_ + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 < 0 ? 0 : lo - 1;
    const int above = lo > n_ ? n_ : lo;
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = 0.5f * (s.zc[below] + s.zc[below + 1]);
    const float b1 = 0.5f * (s.zc[above] + s.zc[above + 1]);
    float denom = c1 - c0;
    if (denom < eps) denom = 1.0f;
    s.zf[k] = b0 + (u - c0) / denom * (b1 - b0);
  }
  wave_lds_fence();
}

// number of entries of arr[0..n) that sort before a value v: (o < v), plus ties (o == v) at positions j < tie_below.
// All lanes walk the same addresses (LDS broadcast), four entries per ds_read_b128 and several reads in flight --
// the scalar one-entry-per-iteration form is bound by LDS latency (58k cycles per ray at 64+128 samples).
__device__ __forceinline__ int count_before(const lds_float* arr, int n, float v, int tie_below) {
  int rank = 0, j = 0;
#pragma unroll 4
  for (; j + 4 <= n; j += 4) {
    const f32x4 o = *(const __attribute__((address_space(3))) f32x4*)(arr + j);
#pragma unroll
    for (int t = 0; t < 4; ++t) rank += (int)(o[t] < v) | ((int)(o[t] == v) & (int)(j + t < tie_below));   // branch-free
  }
  for (; j < n; ++j) {
    const float o = arr[j];
    rank += (int)(o < v) | ((int)(o == v) & (int)(j < tie_below));
  }
  return rank;
}

__device__ __forceinline__ bool wave_ascending(const lds_float* a, int n, int lane) {
  bool bad = false;
  for (int i = lane; i + 1 < n; i += 64) bad |= a[i] > a[i + 1];
  return __ballot(bad) == 0;
}
// first index with a[idx] >= v (STRICT = true) / a[idx] > v (STRICT = false) in an ascending array
template <bool STRICT>
__device__ __forceinline__ int bound_lds(const lds_float* a, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float o = a[mid];
    if (STRICT ? (o < v) : (o <= v)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// zs = sort(cat(zc, zf)), stable with coarse entries before equal fine ones (what torch.sort returns for the values).
// Both inputs ascending (always, unless the caller supplies unsorted depths / uniforms): a two-way merge by binary
// search, rank = own index + number of entries of the OTHER array that precede -- ~8 dependent LDS reads per entry.
// Otherwise: rank counting over all entries, O(N^2/64) compares per lane.
__device__ __forceinline__ void merge_sort_wave(RayScratch& s, int Nc, int Ni, int lane) {
  const int N = Nc + Ni;
  const bool sorted = wave_ascending(s.zc, Nc, lane) && wave_ascending(s.zf, Ni, lane);
  for (int e = lane; e < N; e += 64) {
    const bool is_c = e < Nc;
    const float v = is_c ? s.zc[e] : s.zf[e - Nc];
    int rank;
    if (sorted)
      rank = is_c ? e + bound_lds<true>(s.zf, Ni, v) : (e - Nc) + bound_lds<false>(s.zc, Nc, v);
    else   // ties: a coarse entry precedes every equal fine entry; equal entries of one array keep their order
      rank = count_before(s.zc, Nc, v, is_c ? e : Nc) + count_before(s.zf, Ni, v, is_c ? 0 : e - Nc);
    s.zs[rank] = v;
  }
  wave_lds_fence();
}

}  // namespace crnerf
