// Stand-alone (un-fused) per-ray kernels: alpha compositing over a materialised [R,N,65] tensor and
// sample_pdf + merge.  They back the C-ABI test entry points and the general-N fallback path; the
// production path is render_fused16.hip.  Both are HBM-bandwidth kernels: one ray per wavefront,
// coalesced 260-B sample rows, wave-prefix transmittance.
// Reference: models/rendering.py:116-143 (compositing), :7-46 + :183-187 (sample_pdf, merge).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"
#include "philox.h"
#include "ray_ops.h"

namespace crnerf {

__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                        const float* __restrict__ noise, float noise_std,
                                                        float* __restrict__ weights, float* __restrict__ feature,
                                                        float* __restrict__ depth, long R, int N) {
  __shared__ float wbuf[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    const float* rr = raw + r * (long)N * OUT_DIM;
    const float* zr = z + r * (long)N;
    double carry = 1.0;
    float facc = 0.0f, dacc = 0.0f;
    for (int base = 0; base < N; base += 64) {
      const int n = base + lane;
      const bool valid = n < N;
      const int nc = valid ? n : N - 1;
      const float zn = zr[nc];
      const float znext = zr[nc + 1 < N ? nc + 1 : N - 1];
      const float sigma = rr[(long)nc * OUT_DIM + FEAT_DIM];
      const float nz = (noise && valid) ? noise[r * (long)N + n] * noise_std : 0.0f;
      const float delta = (n == N - 1) ? 1e2f : znext - zn;
      const float alpha = valid ? 1.0f - expf(-delta * fmaxf(sigma + nz, 0.0f)) : 0.0f;
      double incl = (double)(1.0f - alpha);
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_f64(incl, d, 64);
        if (lane >= d) incl *= o;
      }
      double excl = shfl_up_f64(incl, 1, 64);
      if (lane == 0) excl = 1.0;
      const float T = (float)(carry * excl);
      carry *= shfl_f64(incl, 63, 64);
      const float w = alpha * T;
      if (valid) weights[r * (long)N + n] = w;
      dacc += w * zn;
      wbuf[wave][lane] = w;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int cnt = N - base < 64 ? N - base : 64;
      int i = 0;
      for (; i + 8 <= cnt; i += 8) {      // eight 260-B sample rows in flight per wave (a one-row loop reached 45 % of HBM)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rr[(long)(base + i + u) * OUT_DIM + lane];   // lane = channel
#pragma unroll
        for (int u = 0; u < 8; ++u) facc += wbuf[wave][i + u] * v[u];
      }
      for (; i < cnt; ++i) facc += wbuf[wave][i] * rr[(long)(base + i) * OUT_DIM + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) dacc += __shfl_xor(dacc, d);
    feature[r * FEAT_DIM + lane] = facc;
    if (lane == 0) depth[r] = dacc;
  }
}

int launch_composite(const float* raw, const float* z, const float* noise, float noise_std, float* weights, float* feature,
                     float* depth, long R, int N, hipStream_t stream) {
  if (R <= 0) return 0;
  if (N < 1) return set_error(-2, "composite: N must be >= 1");
  const long blocks = (R + 3) / 4;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  hipLaunchKernelGGL(composite_kernel, dim3(grid), dim3(256), 0, stream, raw, z, noise, noise_std, weights, feature, depth, R, N);
  return check_launch("composite_kernel");
}

__global__ __launch_bounds__(64) void sample_pdf_merge_kernel(const float* __restrict__ z_coarse, const float* __restrict__ w_coarse,
                                                              const float* __restrict__ u, long u_stride, float* __restrict__ z_sorted,
                                                              float* __restrict__ z_samples, long R, int Nc, int Ni) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  RayScratch s;
  s.bind((lds_char*)smem, (Nc + 3) & ~3, (Ni + 3) & ~3);   // 16-byte aligned arrays: merge_sort_wave reads them as float4
  for (long r = blockIdx.x; r < R; r += gridDim.x) {
    for (int n = lane; n < Nc; n += 64) {
      s.zc[n] = z_coarse[r * Nc + n];
      s.wc[n] = w_coarse[r * Nc + n];
    }
    wave_lds_fence();
    sample_pdf_wave(s, Nc, Ni, u ? u + r * u_stride : nullptr, lane);
    merge_sort_wave(s, Nc, Ni, lane);
    for (int n = lane; n < Nc + Ni; n += 64) z_sorted[r * (long)(Nc + Ni) + n] = s.zs[n];
    if (z_samples)
      for (int n = lane; n < Ni; n += 64) z_samples[r * (long)Ni + n] = s.zf[n];
    wave_lds_fence();
  }
}

int launch_sample_pdf_merge(const float* z_coarse, const float* weights_coarse, const float* u, long u_stride, float* z_fine_sorted,
                            float* z_samples, long R, int Nc, int Ni, hipStream_t stream) {
  if (R <= 0) return 0;
  if (Nc < 3 || Ni < 1) return set_error(-2, "sample_pdf_merge: need N_samples >= 3 and N_importance >= 1");
  const size_t shmem = (size_t)(3 * ((Nc + 3) & ~3) + 2 * ((Ni + 3) & ~3)) * 4;
  if (shmem > 160 * 1024) return set_error(-2, "sample_pdf_merge: 3*N_samples + 2*N_importance exceeds LDS");
  if (int rc = ensure_dynamic_lds((const void*)sample_pdf_merge_kernel, shmem, "sample_pdf_merge_kernel")) return rc;
  const int grid = (int)(R < 8192 ? R : 8192);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3(grid), dim3(64), shmem, stream, z_coarse, weights_coarse, u, u_stride, z_fine_sorted,
                     z_samples, R, Nc, Ni);
  return check_launch("sample_pdf_merge_kernel");
}

}  // namespace crnerf

// ---------------------------------------------------------------- compositing backward (training)
// Backward twin of composite_kernel: given dL/dfeature[R,64] (and optionally dL/ddepth[R], dL/dweights[R,N])
// produces dL/draw[R,N,65] for the raw MLP outputs.  With g_n = dL/dw_n = Gf.f_n + Gd z_n + Gw_n,
//   dL/df_n     = w_n Gf
//   dL/dalpha_n = T_n (g_n - U_n),   U_n = sum_{m>n} g_m alpha_m prod_{n<j<m} (1 - alpha_j)
//                 (division-free form of -sum_{m>n} g_m w_m / (1 - alpha_n); U_{N-1} = 0,
//                  U_n = g_{n+1} alpha_{n+1} + (1 - alpha_{n+1}) U_{n+1}: an affine suffix scan)
//   dL/dsigma_n = dL/dalpha_n * delta_n * (1 - alpha_n) * [sigma_n + noise_n > 0]
// Reference forward being differentiated: models/rendering.py:121-143 (autograd does this in the reference).
namespace crnerf {

__global__ __launch_bounds__(256) void composite_backward_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                 const float* __restrict__ noise, float noise_std,
                                                                 const float* __restrict__ d_feature, const float* __restrict__ d_depth,
                                                                 const float* __restrict__ d_weights, float* __restrict__ d_raw,
                                                                 long R, int N
#ifdef CRNERF_TIMING
                                                                 , int dbg   // timing experiments: bit k keeps phase A/B/C/D; results are garbage unless 15
#endif
) {
#ifndef CRNERF_TIMING
  constexpr int dbg = 15;
#endif
  // One ray per wavefront.  raw is read ONCE (phase A; a separate strided pass over the sigma column is evicted from L2 before
  // the row pass re-reads the same lines: measured 2x the fetch bytes) and d_raw is written ONCE as a flat stream (phase D).
  extern __shared__ __attribute__((aligned(16))) float smb[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* al = smb + (size_t)wave * (5 * N + 64);   // alpha
  float* Tt = al + N;                       // transmittance
  float* gg = Tt + N;                       // g_n
  float* dsg = gg + N;                      // dL/dsigma_n
  float* sig = dsg + N;                     // raw sigma_n (+ noise)
  float* gfs = sig + N;                     // dL/dfeature of the ray, by channel
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    const float* rr = raw + r * (long)N * OUT_DIM;
    const float* zr = z + r * (long)N;
    float* dr = d_raw + r * (long)N * OUT_DIM;
    // phase A: lane = channel.  32 sample rows in flight per wave; their 32 dot products g_f . f_n are reduced TOGETHER by a
    // transpose-reduce: at the step for lane bit B each lane keeps half of its values and receives the partner's other half, so
    // a batch costs 16+8+4+2+1+1 = 32 cross-lane moves (one per sample) instead of a 6-step butterfly per sample (192).  After
    // the five halving steps lane l holds the sum for sample u = (l >> 1) & 31.  The row's 65th float (sigma) rides along.
    const float gf = d_feature[r * FEAT_DIM + lane];
    const float gd = d_depth ? d_depth[r] : 0.0f;
    gfs[lane] = gf;
    const bool even = !(lane & 1);
    if (dbg & 1)
    for (int n = 0; n < N; n += 32) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {     // unconditional loads (clamped row): a branch around a load costs a vmcnt(0) at the join
        const int nu = n + u < N ? n + u : N - 1;
        const float x = rr[(long)nu * OUT_DIM + lane];
        v[u] = (n + u < N) ? gf * x : 0.0f;
      }
      const int ms = n + (lane & 31);
      if (lane < 32 && ms < N) sig[ms] = rr[(long)ms * OUT_DIM + FEAT_DIM] + (noise ? noise[r * (long)N + ms] * noise_std : 0.0f);
#define CRNERF_FOLD(W, BIT)                                         \
  {                                                                 \
    const bool up = (lane & (BIT)) != 0;                            \
    _Pragma("unroll") for (int u = 0; u < (W) / 2; ++u) {          \
      const float lo = v[u], hi = v[u + (W) / 2];                   \
      v[u] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, (BIT));      \
    }                                                               \
  }
      CRNERF_FOLD(32, 32) CRNERF_FOLD(16, 16) CRNERF_FOLD(8, 8) CRNERF_FOLD(4, 4) CRNERF_FOLD(2, 2)
#undef CRNERF_FOLD
      const float dot = v[0] + __shfl_xor(v[0], 1);
      const int m = n + ((lane >> 1) & 31);
      if (even && m < N) gg[m] = dot + gd * zr[m] + (d_weights ? d_weights[r * (long)N + m] : 0.0f);
    }
    wave_lds_fence();
    // phase B: lane = sample.  alpha, T (same arithmetic as the forward kernel), sigma from LDS
    double carry = 1.0;
    if (dbg & 2)
    for (int base = 0; base < N; base += 64) {
      const int n = base + lane;
      const bool valid = n < N;
      const int nc = valid ? n : N - 1;
      const float zn = zr[nc], znext = zr[nc + 1 < N ? nc + 1 : N - 1];
      const float delta = (n == N - 1) ? 1e2f : znext - zn;
      const float alpha = valid ? 1.0f - expf(-delta * fmaxf(sig[nc], 0.0f)) : 0.0f;
      double incl = (double)(1.0f - alpha);
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_f64(incl, d, 64);
        if (lane >= d) incl *= o;
      }
      double excl = shfl_up_f64(incl, 1, 64);
      if (lane == 0) excl = 1.0;
      if (valid) { al[n] = alpha; Tt[n] = (float)(carry * excl); }
      carry *= shfl_f64(incl, 63, 64);
    }
    wave_lds_fence();
    // phase C: reverse affine scan, lane = sample (lane 0 = LAST sample of the chunk)
    float ucarry = 0.0f;                 // U of the lowest sample of the chunk processed before (the later samples)
    float a_next = 1.0f, b_next = 0.0f;  // (1-alpha, g*alpha) of the sample right after the current chunk
    if (dbg & 4)
    for (int top = N; top > 0; top -= 64) {
      const int n = top - 1 - lane;                 // descending
      const bool valid = n >= 0;
      const int nc = valid ? n : 0;
      // element for sample n: maps U_n = b_{n+1} + a_{n+1} U_{n+1}; shift by one so lane holds (a_{n+1}, b_{n+1})
      const float a_self = valid ? 1.0f - al[nc] : 1.0f, b_self = valid ? gg[nc] * al[nc] : 0.0f;
      float a = __shfl_up(a_self, 1, 64), b = __shfl_up(b_self, 1, 64);
      if (lane == 0) { a = a_next; b = b_next; }
      float A = a, B = b;                            // inclusive scan of the affine maps: (A,B) takes the chunk's incoming U to U_n
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const float Ap = __shfl_up(A, d, 64), Bp = __shfl_up(B, d, 64);
        if (lane >= d) { B = fmaf(A, Bp, B); A = A * Ap; }
      }
      const float U = fmaf(A, ucarry, B);
      if (valid) {
        const float zn = zr[nc], znext = zr[nc + 1 < N ? nc + 1 : N - 1];
        const float delta = (nc == N - 1) ? 1e2f : znext - zn;
        const float dalpha = Tt[nc] * (gg[nc] - U);
        dsg[nc] = (sig[nc] > 0.0f) ? dalpha * delta * (1.0f - al[nc]) : 0.0f;
      }
      ucarry = __shfl(U, 63, 64);
      a_next = __shfl(a_self, 63, 64);
      b_next = __shfl(b_self, 63, 64);
    }
    wave_lds_fence();
    // phase D: the ray's d_raw block [N][65] as ONE flat contiguous stream (element e = 65 n + c: w_n gf_c for c < 64,
    // dL/dsigma_n for c = 64), 16 bytes per lane when the block is 16-byte aligned (N % 4 == 0).  Writing channels 0..63 and the
    // sigma column in separate passes leaves every 260-B row as partial lines that leave L2 before they are completed.
    const int total = N * OUT_DIM;
    auto elem = [&](int e) {
      const int nn = e / OUT_DIM, c = e - nn * OUT_DIM;
      return c < FEAT_DIM ? (al[nn] * Tt[nn]) * gfs[c] : dsg[nn];
    };
    if (!(dbg & 8)) {
    } else if ((N & 3) == 0) {
      for (int e = 4 * lane; e < total; e += 256) {
        const float4 o = make_float4(elem(e), elem(e + 1), elem(e + 2), elem(e + 3));
        *(float4*)(dr + e) = o;
      }
    } else {
      for (int e = lane; e < total; e += 64) dr[e] = elem(e);
    }
    wave_lds_fence();
  }
}

int launch_composite_backward(const float* raw, const float* z, const float* noise, float noise_std, const float* d_feature,
                              const float* d_depth, const float* d_weights, float* d_raw, long R, int N, hipStream_t stream) {
  if (R <= 0) return 0;
  if (N < 1) return set_error(-2, "composite_backward: N must be >= 1");
  const size_t shmem = (size_t)4 * (5 * N + 64) * sizeof(float);
  if (shmem > 160 * 1024) return set_error(-2, "composite_backward: N too large for LDS (4 rays x (5 N + 64) floats: N <= 2035)");
  if (int rc = ensure_dynamic_lds((const void*)composite_backward_kernel, shmem, "composite_backward_kernel")) return rc;
  const long blocks = (R + 3) / 4;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
#ifdef CRNERF_TIMING
  static const int dbg = getenv("CRNERF_CB_PHASES") ? atoi(getenv("CRNERF_CB_PHASES")) : 15;   // -DCRNERF_TIMING builds only (tools/cb_phase_bench.py)
  hipLaunchKernelGGL(composite_backward_kernel, dim3(grid), dim3(256), shmem, stream, raw, z, noise, noise_std, d_feature, d_depth,
                     d_weights, d_raw, R, N, dbg);
#else
  hipLaunchKernelGGL(composite_backward_kernel, dim3(grid), dim3(256), shmem, stream, raw, z, noise, noise_std, d_feature, d_depth,
                     d_weights, d_raw, R, N);
#endif
  return check_launch("composite_backward_kernel");
}

// The renderer's in-kernel draws as a tensor (crnerf_rng_fill_f32): out[r * n + s] = draw(seed, stream, ray_offset + r, s).
__global__ __launch_bounds__(256) void rng_fill_kernel(float* __restrict__ out, long total, int n, RayRng rng, int stream_id, long ray_offset) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / n;
    const int s = (int)(idx - r * n);
    out[idx] = stream_id >= RNG_STREAM_NOISE_COARSE ? rng.normal(stream_id, ray_offset + r, s) : rng.uniform(stream_id, ray_offset + r, s);
  }
}

int launch_rng_fill(float* out, long R, int n, unsigned long long seed, int stream_id, long ray_offset, hipStream_t stream) {
  const long total = R * (long)n;
  if (total <= 0) return 0;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, out, total, n,
                     RayRng{(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)}, stream_id, ray_offset);
  return check_launch("rng_fill_kernel");
}

}  // namespace crnerf
