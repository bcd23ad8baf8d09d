// Cross-ray feature transformation + decoder (the only step of the path with a dependence BETWEEN rays).
//
// Reference: style_net.forward models/linearStyleTransfer.py:284-291 -> MulLayer.forward :58-94
//            -> CNN.forward :28-37; NeuralRenderer.forward models/nerf_decoder_stylenerf.py:279-291
//            (n_blocks == 0: rgb = sigmoid(Conv1x1_{64->3}(x))).
//
// Data layout: the feature grid is consumed pixel-major, x[HW,64] -- exactly the renderer's
// feature_fine[R,64]; the reference's NCHW [1,64,H,W] is a transposed view of the same memory.
// RGB is written planar [3,HW] (= NCHW [1,3,H,W]).
//
// The math is split at its two global reductions so that a collective can sit between the pieces
// when rays are sharded over GPUs (option B):
//   chansum   : per-channel sums over pixels                     (-> all-reduce -> mean)
//   gram      : G_sum = sum_px f(x - mean) f(x - mean)^T, f = the 64->128->64->32 1x1-conv chain
//                                                                  (-> all-reduce)
//   matrix    : M = fc(G_sum / count)                              (replicated, tiny)
//   fold      : everything after the Gram is affine per pixel: rgb_pre = A x + v with
//               A = Wrgb Wunzip T Wcomp (3x64), T = sMatrix cMatrix; folded once per image
//   apply     : rgb = sigmoid(A x + v)                             (HBM-bound stream: 256 B in, 12 B out)
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "crossray.h"

namespace crnerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- channel sums
__global__ __launch_bounds__(256) void chansum_partial_kernel(const float* __restrict__ x, long HW, float* __restrict__ partial) {
  // thread = (pixel slot, channel quad); 16 channel quads x 16 pixel slots per block
  __shared__ f32x4 red[256];
  const int cq = threadIdx.x & 15, ps = threadIdx.x >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (long px = (long)blockIdx.x * 16 + ps; px < HW; px += (long)gridDim.x * 16) acc += *(const f32x4*)(x + px * 64 + cq * 4);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 8; s >= 1; s >>= 1) {
    if (ps < s) red[threadIdx.x] += red[threadIdx.x + s * 16];
    __syncthreads();
  }
  if (ps == 0) *(f32x4*)(partial + (long)blockIdx.x * 64 + cq * 4) = red[cq];
}

__global__ void reduce_rows_kernel(const float* __restrict__ partial, int rows, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.0f;
  for (int r = 0; r < rows; ++r) s += partial[(long)r * cols + c];
  out[c] = s;
}

int launch_crossray_chansum(const float* x, long HW, float* sum_out, float* workspace, hipStream_t stream) {
  if (HW <= 0) return set_error(-2, "crossray_chansum: empty grid");
  const int grid = (int)((HW + 15) / 16 < CROSSRAY_MAX_BLOCKS ? (HW + 15) / 16 : CROSSRAY_MAX_BLOCKS);
  hipLaunchKernelGGL(chansum_partial_kernel, dim3(grid), dim3(256), 0, stream, x, HW, workspace);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, stream, workspace, grid, 64, sum_out);
  return check_launch("crossray_chansum");
}

// ---------------------------------------------------------------- Gram of the conv chain
constexpr int GR_W1 = 0;                      // [128][64]
constexpr int GR_W2 = GR_W1 + 128 * 64;       // [64][128]
constexpr int GR_W3 = GR_W2 + 64 * 128;       // [32][64]
constexpr int GR_B1 = GR_W3 + 32 * 64;        // 128
constexpr int GR_B2 = GR_B1 + 128;            // 64
constexpr int GR_B3 = GR_B2 + 64;             // 32
constexpr int GR_MEAN = GR_B3 + 32;           // 64
constexpr int GR_H3 = GR_MEAN + 64;           // [256][33]
constexpr int GR_FLOATS = GR_H3 + 256 * 33;

__device__ __forceinline__ float lrelu02(float v) { return v > 0.0f ? v : 0.2f * v; }

__global__ __launch_bounds__(256, 1) void gram_partial_kernel(const float* __restrict__ x, long HW, const float* __restrict__ mean,
                                                              CnnTensors w, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 128 * 64; i += 256) { sm[GR_W1 + i] = w.w1[i]; sm[GR_W2 + i] = w.w2[i]; }
  for (int i = tid; i < 32 * 64; i += 256) sm[GR_W3 + i] = w.w3[i];
  if (tid < 128) sm[GR_B1 + tid] = w.b1[tid];
  if (tid < 64) { sm[GR_B2 + tid] = w.b2[tid]; sm[GR_MEAN + tid] = mean[tid]; }
  if (tid < 32) sm[GR_B3 + tid] = w.b3[tid];
  __syncthreads();

  // this thread's 4 Gram entries: row a, columns b0..b0+3
  const int ga = tid >> 3, gb0 = (tid & 7) * 4;
  float g[4] = {0, 0, 0, 0};

  const long nbatch = (HW + 255) / 256;
  for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const long px = batch * 256 + tid;
    const bool valid = px < HW;
    float xin[64];
    {
      const float* row = x + (valid ? px : 0) * 64;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const f32x4 v = *(const f32x4*)(row + 4 * c);
        const f32x4 m = *(const f32x4*)(sm + GR_MEAN + 4 * c);
        xin[4 * c + 0] = v[0] - m[0]; xin[4 * c + 1] = v[1] - m[1]; xin[4 * c + 2] = v[2] - m[2]; xin[4 * c + 3] = v[3] - m[3];
      }
    }
    float h2[64];
#pragma unroll
    for (int o = 0; o < 64; ++o) h2[o] = sm[GR_B2 + o];
    // layer 1 (64->128, LeakyReLU 0.2) produced 8 outputs at a time and folded straight into layer 2
#pragma unroll 1
    for (int oc = 0; oc < 128; oc += 8) {
      float h1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float a = sm[GR_B1 + oc + u];
        const float* wr = sm + GR_W1 + (oc + u) * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const f32x4 wv = *(const f32x4*)(wr + 4 * c);
          a = fmaf(wv[0], xin[4 * c + 0], a); a = fmaf(wv[1], xin[4 * c + 1], a);
          a = fmaf(wv[2], xin[4 * c + 2], a); a = fmaf(wv[3], xin[4 * c + 3], a);
        }
        h1[u] = lrelu02(a);
      }
#pragma unroll
      for (int o = 0; o < 64; ++o) {
        const f32x4 wa = *(const f32x4*)(sm + GR_W2 + o * 128 + oc);
        const f32x4 wb = *(const f32x4*)(sm + GR_W2 + o * 128 + oc + 4);
        float a = h2[o];
        a = fmaf(wa[0], h1[0], a); a = fmaf(wa[1], h1[1], a); a = fmaf(wa[2], h1[2], a); a = fmaf(wa[3], h1[3], a);
        a = fmaf(wb[0], h1[4], a); a = fmaf(wb[1], h1[5], a); a = fmaf(wb[2], h1[6], a); a = fmaf(wb[3], h1[7], a);
        h2[o] = a;
      }
    }
#pragma unroll
    for (int o = 0; o < 64; ++o) h2[o] = lrelu02(h2[o]);
    __syncthreads();  // previous batch's Gram reads of the h3 buffer are done
#pragma unroll 4
    for (int o = 0; o < 32; ++o) {
      float a = sm[GR_B3 + o];
      const float* wr = sm + GR_W3 + o * 64;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const f32x4 wv = *(const f32x4*)(wr + 4 * c);
        a = fmaf(wv[0], h2[4 * c + 0], a); a = fmaf(wv[1], h2[4 * c + 1], a);
        a = fmaf(wv[2], h2[4 * c + 2], a); a = fmaf(wv[3], h2[4 * c + 3], a);
      }
      sm[GR_H3 + tid * 33 + o] = valid ? a : 0.0f;
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < 256; ++p) {
      const float ha = sm[GR_H3 + p * 33 + ga];
      g[0] = fmaf(ha, sm[GR_H3 + p * 33 + gb0 + 0], g[0]);
      g[1] = fmaf(ha, sm[GR_H3 + p * 33 + gb0 + 1], g[1]);
      g[2] = fmaf(ha, sm[GR_H3 + p * 33 + gb0 + 2], g[2]);
      g[3] = fmaf(ha, sm[GR_H3 + p * 33 + gb0 + 3], g[3]);
    }
  }
  float* out = partial + (long)blockIdx.x * 1024 + ga * 32 + gb0;
  out[0] = g[0]; out[1] = g[1]; out[2] = g[2]; out[3] = g[3];
}

int launch_crossray_gram(const float* x, long HW, const float* mean, const CnnTensors& w, float* gram_sum, float* workspace,
                         hipStream_t stream) {
  if (HW <= 0) return set_error(-2, "crossray_gram: empty grid");
  const long nbatch = (HW + 255) / 256;
  const int grid = (int)(nbatch < 256 ? nbatch : 256);
  const size_t shmem = (size_t)GR_FLOATS * 4;
  hipError_t e = hipFuncSetAttribute((const void*)gram_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return set_error(-10, "hipFuncSetAttribute(gram_partial_kernel) failed");
  hipLaunchKernelGGL(gram_partial_kernel, dim3(grid), dim3(256), shmem, stream, x, HW, mean, w, workspace);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(4), dim3(256), 0, stream, workspace, grid, 1024, gram_sum);
  return check_launch("crossray_gram");
}

// ---------------------------------------------------------------- M = fc(G_sum / count): one wave per output row
__global__ __launch_bounds__(256) void gram_fc_kernel(const float* __restrict__ gram_sum, float inv_count, const float* __restrict__ fc_w,
                                                      const float* __restrict__ fc_b, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float* wr = fc_w + (long)row * 1024;
  float a = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x4 wv = *(const f32x4*)(wr + k * 256 + lane * 4);
    const f32x4 gv = *(const f32x4*)(gram_sum + k * 256 + lane * 4);
    a = fmaf(wv[0], gv[0] * inv_count, a); a = fmaf(wv[1], gv[1] * inv_count, a);
    a = fmaf(wv[2], gv[2] * inv_count, a); a = fmaf(wv[3], gv[3] * inv_count, a);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
  if (lane == 0) out[row] = a + fc_b[row];
}

int launch_crossray_matrix(const float* gram_sum, double count, const float* fc_w, const float* fc_b, float* out, hipStream_t stream) {
  if (!(count > 0)) return set_error(-2, "crossray_matrix: count must be positive");
  hipLaunchKernelGGL(gram_fc_kernel, dim3(256), dim3(256), 0, stream, gram_sum, (float)(1.0 / count), fc_w, fc_b, out);
  return check_launch("crossray_matrix");
}

// ---------------------------------------------------------------- fold: A[3][64], v[3]  (single 64-thread block)
__global__ __launch_bounds__(64) void fold_kernel(const float* __restrict__ sM, const float* __restrict__ cM, const float* __restrict__ c_mean,
                                                  const float* __restrict__ s_mean, FoldTensors w, float* __restrict__ affine) {
  __shared__ float T[32][33], U[3][32], P[3][64], Q[3][32];
  const int t = threadIdx.x;
  if (sM) {
    // T = sMatrix @ cMatrix                                   linearStyleTransfer.py:86
    for (int e = t; e < 1024; e += 64) {
      const int i = e >> 5, j = e & 31;
      float a = 0.0f;
      for (int k = 0; k < 32; ++k) a = fmaf(sM[i * 32 + k], cM[k * 32 + j], a);
      T[i][j] = a;
    }
    __syncthreads();
    // P = Wrgb @ Wunzip (3x32), Q = P @ T (3x32), A = Q @ Wcomp (3x64)
    if (t < 32)
      for (int r = 0; r < 3; ++r) {
        float a = 0.0f;
        for (int k = 0; k < 64; ++k) a = fmaf(w.rgb_w[r * 64 + k], w.unzip_w[k * 32 + t], a);
        U[r][t] = a;
      }
    __syncthreads();
    if (t < 32)
      for (int r = 0; r < 3; ++r) {
        float a = 0.0f;
        for (int k = 0; k < 32; ++k) a = fmaf(U[r][k], T[k][t], a);
        Q[r][t] = a;
      }
    __syncthreads();
    for (int r = 0; r < 3; ++r) {
      float a = 0.0f;
      for (int k = 0; k < 32; ++k) a = fmaf(Q[r][k], w.comp_w[k * 64 + t], a);
      P[r][t] = a;
      affine[r * 64 + t] = a;
    }
    __syncthreads();
    // v = Q (bcomp) - A cMean + Wrgb (bunzip + sMean) + brgb
    if (t < 3) {
      float a = w.rgb_b[t];
      for (int k = 0; k < 32; ++k) a = fmaf(Q[t][k], w.comp_b[k], a);
      for (int k = 0; k < 64; ++k) a = fmaf(-P[t][k], c_mean[k], a);
      for (int k = 0; k < 64; ++k) a = fmaf(w.rgb_w[t * 64 + k], w.unzip_b[k] + s_mean[k], a);
      affine[192 + t] = a;
    }
  } else {
    // type == "content": decoder only                          linearStyleTransfer.py:285-287
    for (int r = 0; r < 3; ++r) affine[r * 64 + t] = w.rgb_w[r * 64 + t];
    if (t < 3) affine[192 + t] = w.rgb_b[t];
  }
}

int launch_crossray_fold(const float* sM, const float* cM, const float* c_mean, const float* s_mean, const FoldTensors& w,
                         float* affine, hipStream_t stream) {
  hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(64), 0, stream, sM, cM, c_mean, s_mean, w, affine);
  return check_launch("crossray_fold");
}

// ---------------------------------------------------------------- apply: rgb[c][px] = sigmoid(A[c] . x[px] + v[c])
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ x, long HW, const float* __restrict__ affine,
                                                    float* __restrict__ rgb, long plane_stride) {
  __shared__ float A[196];
  if (threadIdx.x < 195) A[threadIdx.x] = affine[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x & 3;  // 4 lanes per pixel, 16 channels each
  for (long px = ((long)blockIdx.x * 256 + threadIdx.x) >> 2;; px += ((long)gridDim.x * 256) >> 2) {
    const bool valid = px < HW;
    if (__all(!valid)) break;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (valid) {
      const float* row = x + px * 64;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + 16 * k;
        const f32x4 v = *(const f32x4*)(row + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r = fmaf(A[c + e], v[e], r);
          g = fmaf(A[64 + c + e], v[e], g);
          b = fmaf(A[128 + c + e], v[e], b);
        }
      }
    }
    r += __shfl_xor(r, 1); g += __shfl_xor(g, 1); b += __shfl_xor(b, 1);
    r += __shfl_xor(r, 2); g += __shfl_xor(g, 2); b += __shfl_xor(b, 2);
    if (valid && q < 3) {
      const float pre = (q == 0 ? r : (q == 1 ? g : b)) + A[192 + q];
      rgb[q * plane_stride + px] = 1.0f / (1.0f + expf(-pre));
    }
  }
}

int launch_crossray_apply(const float* x, long HW, const float* affine, float* rgb, long plane_stride, hipStream_t stream) {
  if (HW <= 0) return 0;
  const long blocks = (HW * 4 + 255) / 256;
  const int grid = (int)(blocks < 2048 ? blocks : 2048);
  hipLaunchKernelGGL(apply_kernel, dim3(grid), dim3(256), 0, stream, x, HW, affine, rgb, plane_stride);
  return check_launch("crossray_apply");
}

}  // namespace crnerf
