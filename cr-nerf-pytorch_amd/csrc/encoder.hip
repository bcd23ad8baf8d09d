// Appearance encoder: encoder_sameoutputsize.forward, models/linearStyleTransfer.py:208-276
//   conv1 1x1 3->3 | reflpad, conv2 3x3 3->64, lrelu | reflpad, conv3 3x3 64->64, lrelu | maxpool2 |
//   reflpad, conv4 3x3 64->128, lrelu | reflpad, conv5 3x3 128->128, lrelu | maxpool2 |
//   reflpad, conv6 3x3 128->128, lrelu | AdaptiveAvgPool2d(32) | conv7 1x1 128->64, lrelu
// It runs once per image on the 1/8-scale photo (a few thousand pixels, ~0.6 GFLOP) and produces the
// style operand of the cross-ray decoder, written pixel-major [1024,64] (= what crossray.hip consumes).
// Direct convolutions, activations pixel-major (HWC): one workgroup = 8 pixels of a row x 64 output channels, input
// channels split over its four waves; the input values are wave-uniform scalar loads, the weight row a coalesced 256-B
// read of the [cin][tap][cout] re-layout made on the fly into the workspace.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace crnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : 0.2f * v; }
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }   // ReflectionPad2d(1)

// w[cout][cin][taps] -> wt[cin][taps][cout]
__global__ void transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin, int taps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cout * cin * taps) return;
  const int o = idx / (cin * taps), r = idx % (cin * taps);
  wt[r * cout + o] = w[idx];
}

// all seven re-layouts of a training forward in ONE launch (a block belongs to the job whose block range holds it; jobs by value)
struct TransposeJobs { const float* w[7]; float* wt[7]; int cout[7], rows[7], first_block[8]; };
__global__ void transpose_weights_batch_kernel(TransposeJobs j) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < 7; ++k) l += (int)blockIdx.x >= j.first_block[k];
  const int idx = ((int)blockIdx.x - j.first_block[l]) * blockDim.x + threadIdx.x;
  if (idx >= j.cout[l] * j.rows[l]) return;
  const int o = idx / j.rows[l], r = idx % j.rows[l];
  j.wt[l][r * j.cout[l] + o] = j.w[l][idx];
}

// NCHW [C,H,W] -> HWC [H*W, C]
__global__ void chw_to_hwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * HW) return;
  const int px = idx / C, c = idx % C;
  out[idx] = in[c * HW + px];
}

// The first layer (conv1: 1x1, 3 -> 3, no activation) together with the NCHW -> pixel-major re-layout of the photo: a0[px][c] = img[c][px],
// y1[px][o] = b[o] + sum_c a0[px][c] wt[c][o] with the products chained in conv_kernel<1, false>'s order (c = 0, 1, 2 onto 0, then + bias):
// the same bits as chw_to_hwc_kernel + that kernel, one launch instead of two per encoder pass.
__global__ void chw_conv1_kernel(const float* __restrict__ img, const float* __restrict__ wt, const float* __restrict__ b, float* __restrict__ a0,
                                 float* __restrict__ y1, int HW) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= HW) return;
  const float x0 = img[px], x1 = img[HW + px], x2 = img[2 * HW + px];
  a0[3 * px + 0] = x0; a0[3 * px + 1] = x1; a0[3 * px + 2] = x2;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    float acc = fmaf(x0, wt[o], 0.0f);
    acc = fmaf(x1, wt[3 + o], acc);
    acc = fmaf(x2, wt[6 + o], acc);
    y1[3 * px + o] = b[o] + (((acc + 0.0f) + 0.0f) + 0.0f);   // (conv_kernel adds the three idle waves' zero partial sums: -0 becomes +0 there too)
  }
}

// out[px][o] = act(b[o] + sum_{c,tap} in[reflect(px+tap)][c] * wt[c][tap][o]);  TAPS = 9 (3x3, reflection pad 1) or 1.
// One workgroup = PX horizontally adjacent pixels x 64 output channels; its four waves split the input channels (a
// quarter each when cin % 16 == 0) and their partial sums are added in wave order through LDS.  Per wave the weight row is
// ONE coalesced 256-B read per (tap, c) feeding PX FMAs, the PX input values are wave-uniform -> scalar loads (s_load)
// and SGPR FMA operands.  The images here are small (32x32 ... 256x256), so what bounds this kernel is the length of one
// wave's dependent load -> FMA chain, not arithmetic: one pixel per wave (two vector loads per FMA) ran 3.6 ms on the
// training step's 128x128 images, eight pixels per wave 1.4 ms, the channel split cuts the chain four times again.
constexpr int CONV_PX = 8;
template <int TAPS, bool ACT>
__global__ __launch_bounds__(256) void conv_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ b,
                                                   float* __restrict__ out, int H, int W, int cin, int cout) {
  __shared__ float red[3][CONV_PX][64];
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.y * 64 + lane;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int groups_per_row = (W + CONV_PX - 1) / CONV_PX;
  const int y = blockIdx.x / groups_per_row, x0 = (blockIdx.x % groups_per_row) * CONV_PX;
  const bool split = (cin & 15) == 0;
  const int c0 = split ? wave * (cin >> 2) : 0, c1 = split ? c0 + (cin >> 2) : (wave == 0 ? cin : 0);
  const int oc = o < cout ? o : cout - 1;
  float acc[CONV_PX];
#pragma unroll
  for (int j = 0; j < CONV_PX; ++j) acc[j] = 0.0f;
  if (TAPS == 9) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const long row = (long)reflect(y + ky - 1, H) * W;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* ip[CONV_PX];
#pragma unroll
        for (int j = 0; j < CONV_PX; ++j) ip[j] = in + (row + reflect((x0 + j < W ? x0 + j : W - 1) + kx - 1, W)) * cin;
        const float* wp = wt + (ky * 3 + kx) * cout + oc;
#pragma unroll 4
        for (int c = c0; c < c1; ++c) {
          const float wv = wp[(long)c * 9 * cout];
#pragma unroll
          for (int j = 0; j < CONV_PX; ++j) acc[j] = fmaf(ip[j][c], wv, acc[j]);
        }
      }
    }
  } else {
    const float* ip[CONV_PX];
#pragma unroll
    for (int j = 0; j < CONV_PX; ++j) ip[j] = in + ((long)y * W + (x0 + j < W ? x0 + j : W - 1)) * cin;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
      const float wv = wt[(long)c * cout + oc];
#pragma unroll
      for (int j = 0; j < CONV_PX; ++j) acc[j] = fmaf(ip[j][c], wv, acc[j]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < CONV_PX; ++j) red[wave - 1][j][lane] = acc[j];
  }
  __syncthreads();
  if (wave == 0 && o < cout) {
    const float bias = b[o];
#pragma unroll
    for (int j = 0; j < CONV_PX; ++j) {
      const float v = bias + (((acc[j] + red[0][j][lane]) + red[1][j][lane]) + red[2][j][lane]);   // fixed order: deterministic
      if (x0 + j < W) out[((long)y * W + x0 + j) * cout + o] = ACT ? lrelu(v) : v;
    }
  }
}

// ---- the wide layers (cin >= 64) as GEMMs on the fp32 matrix cores -------------------------------------------------------
// At the sizes this encoder sees (32x32 ... 64x48 photos: 64 ... 3,072 pixels per layer) the direct kernel above is bound by
// the length of one wave's dependent load -> FMA chain (16 ... 36 us per layer whatever the map size; the data gradient twice
// that).  A 3x3 convolution over reflection-padded patches IS a GEMM over the patch matrix X[px][c*9+tap] that the backward
// builds for the weight gradient anyway, with the ORIGINAL weight tensor w[cout][cin][3][3] = W[cout][cin*9] as second operand.
//
// C[m][n] = act(bias[n] + sum_k A[m][k] * B[n][k]): both operands K-contiguous ("NT").  One workgroup = ONE 32 x 32 tile of C;
// its four waves split K (fixed ranges; partial tiles are added in wave order through LDS: deterministic).  v_mfma_f32_32x32x2_f32:
// lane (i, kk) loads 16 bytes of row i of A and of row i of B at k = 8s + 4kk .. + 3 and issues four MFMAs (k pairs {8s + t,
// 8s + 4 + t}; the order of k inside a dot product is free as long as both operands agree).  K % 8 == 0, lda / ldb % 4 == 0.
template <bool ACT>
__global__ __launch_bounds__(256) void enc_gemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                          const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N, int K) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int mr = m0 + i < M ? m0 + i : M - 1, nr = n0 + i < N ? n0 + i : N - 1;   // clamped rows: always readable, dropped at the store
  const int chunks = K >> 3, q = chunks >> 2, r = chunks & 3;
  const int s0 = wave * q + (wave < r ? wave : r), s1 = s0 + q + (wave < r ? 1 : 0);
  const float* ap = A + (long)mr * lda + 4 * kk;
  const float* bp = B + (long)nr * ldb + 4 * kk;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  constexpr int U = 4;                                    // chunks in flight: 8 loads ahead of their 16 MFMAs
  f32x4 a[U], b[U];
  int s = s0;
  for (; s + U <= s1; s += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) { a[u] = *(const f32x4*)(ap + 8 * (s + u)); b[u] = *(const f32x4*)(bp + 8 * (s + u)); }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][t], acc, 0, 0, 0);
  }
  for (; s < s1; ++s) {
    const f32x4 av = *(const f32x4*)(ap + 8 * s), bv = *(const f32x4*)(bp + 8 * s);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
  }
  __syncthreads();
  if (wave == 0) {
    const int n = n0 + i;
    const float bv = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * kk;            // 32x32 C/D layout: register e of lane (i, kk) = row, column i
      const float v = bv + (((acc[e] + red[0][e][lane]) + red[1][e][lane]) + red[2][e][lane]);
      if (m < M && n < N) C[(long)m * ldc + n] = ACT ? lrelu(v) : v;
    }
  }
}

// X[px][c * 9 + tap] = in[reflect(px + tap)][c]: the reflection-padded 3x3 patches as a [HW, 9 cin] matrix
__global__ void enc_im2col_kernel(const float* __restrict__ in, float* __restrict__ X, int H, int W, int cin) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)H * W * cin * 9;
  if (idx >= n) return;
  const int col = (int)(idx % (cin * 9)), c = col / 9, tap = col % 9;
  const long px = idx / (cin * 9);
  const int py = (int)(px / W), pxx = (int)(px % W);
  X[idx] = in[((long)reflect(py + tap / 3 - 1, H) * W + reflect(pxx + tap % 3 - 1, W)) * cin + c];
}

// The same patch matrix over MaxPool2d(2, 2)(in): the pooled map is never written -- every patch element takes the maximum of its 2 x 2 window
// of the un-pooled [Hs, Ws] map (maxpool2_kernel's expression), H = Hs / 2, W = Ws / 2.  One launch instead of pool + im2col.
__global__ void enc_im2col_pooled_kernel(const float* __restrict__ in, float* __restrict__ X, int Hs, int Ws, int cin) {
  const int H = Hs / 2, W = Ws / 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)H * W * cin * 9;
  if (idx >= n) return;
  const int col = (int)(idx % (cin * 9)), c = col / 9, tap = col % 9;
  const long px = idx / (cin * 9);
  const int py = (int)(px / W), pxx = (int)(px % W);
  const int y = reflect(py + tap / 3 - 1, H), x = reflect(pxx + tap % 3 - 1, W);
  const float* p = in + ((long)(2 * y) * Ws + 2 * x) * cin + c;
  X[idx] = fmaxf(fmaxf(p[0], p[cin]), fmaxf(p[(long)Ws * cin], p[(long)Ws * cin + cin]));
}

void enc_im2col(const float* in, float* X, int H, int W, int cin, hipStream_t st) {
  const long nx = (long)H * W * cin * 9;
  hipLaunchKernelGGL(enc_im2col_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, in, X, H, W, cin);
}
void enc_gemm_nt(bool act, const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M, int N, int K, hipStream_t st) {
  const dim3 grid((M + 31) / 32, (N + 31) / 32);
  if (act) hipLaunchKernelGGL(enc_gemm_nt_kernel<true>, grid, dim3(256), 0, st, A, lda, B, ldb, bias, C, ldc, M, N, K);
  else hipLaunchKernelGGL(enc_gemm_nt_kernel<false>, grid, dim3(256), 0, st, A, lda, B, ldb, bias, C, ldc, M, N, K);
}
// a 3x3 (xcol = patch matrix to fill; kept by the training forward for the weight gradient) or 1x1 (xcol = null) layer with cin >= 64:
// out[px][o] = lrelu(b[o] + sum X[px][k] w[o][k]),  w = the module's own [cout][cin][k][k] tensor
void enc_conv_gemm(int taps, const float* in, float* xcol, const float* w, const float* b, float* out, int H, int W, int cin, int cout, hipStream_t st) {
  const float* X = in;
  if (taps == 9) { enc_im2col(in, xcol, H, W, cin, st); X = xcol; }
  enc_gemm_nt(true, X, cin * taps, w, cin * taps, b, out, cout, H * W, cout, cin * taps, st);
}

// 3x3 layer over MaxPool2d(2, 2)(in), in = the un-pooled [Hs, Ws, cin] map (enc_im2col_pooled_kernel)
void enc_conv_gemm_pooled(const float* in, float* xcol, const float* w, const float* b, float* out, int Hs, int Ws, int cin, int cout, hipStream_t st) {
  const int H = Hs / 2, W = Ws / 2;
  const long nx = (long)H * W * cin * 9;
  hipLaunchKernelGGL(enc_im2col_pooled_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, in, xcol, Hs, Ws, cin);
  enc_gemm_nt(true, xcol, cin * 9, w, cin * 9, b, out, cout, H * W, cout, cin * 9, st);
}

// MaxPool2d(2,2), floor mode
__global__ void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Ho * Wo * C) return;
  const int c = idx % C, px = idx / C, y = px / Wo, x = px % Wo;
  const float* p = in + ((long)(2 * y) * W + 2 * x) * C + c;
  out[idx] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[(long)W * C], p[(long)W * C + C]));
}

// AdaptiveAvgPool2d(S): window [floor(o*in/S), ceil((o+1)*in/S))
// Row band (round 6, the encoder over a band of an image's rows in ray-parallel training): `in` holds rows [yoff, yoff + H) of a map of Hg rows and
// only the output rows [o0, o1) are produced (out[(oy - o0) * S + ox]); the windows are those of the WHOLE map.  Whole map: Hg = H, yoff = 0, [0, S).
__global__ void adaptive_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int S, int Hg, int yoff, int o0, int o1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (o1 - o0) * S * C) return;
  const int c = idx % C, px = idx / C, oy = o0 + px / S, ox = px % S;
  const int y0 = (oy * Hg) / S, y1 = ((oy + 1) * Hg + S - 1) / S, x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
  float s = 0.0f;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) s += in[((long)(y - yoff) * W + x) * C + c];
  out[idx] = s / (float)((y1 - y0) * (x1 - x0));
}

template <int TAPS, bool ACT>
static void conv(const float* in, const float* wt, const float* b, float* out, int H, int W, int cin, int cout, hipStream_t st);

// thin launch wrappers shared with encoder_train.hip
void enc_conv(int taps, bool act, const float* in, const float* wt, const float* b, float* out, int H, int W, int cin, int cout, hipStream_t st) {
  if (taps == 9) conv<9, true>(in, wt, b, out, H, W, cin, cout, st);
  else if (act) conv<1, true>(in, wt, b, out, H, W, cin, cout, st);
  else conv<1, false>(in, wt, b, out, H, W, cin, cout, st);
}
void enc_transpose_weights(const float* w, float* wt, int cout, int cin, int taps, hipStream_t st) {
  const int n = cout * cin * taps;
  hipLaunchKernelGGL(transpose_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, wt, cout, cin, taps);
}
// w: the encoder's 14 tensors (weights at even indices); wt[l] = where layer l's [cin][tap][cout] copy goes
void enc_transpose_weights_all(const float* const* w, float* const* wt, const int* cout, const int* cin, const int* taps, hipStream_t st) {
  TransposeJobs j;
  int blocks = 0;
  for (int l = 0; l < 7; ++l) {
    j.w[l] = w[2 * l]; j.wt[l] = wt[l]; j.cout[l] = cout[l]; j.rows[l] = cin[l] * taps[l];
    j.first_block[l] = blocks;
    blocks += (cout[l] * cin[l] * taps[l] + 255) / 256;
  }
  j.first_block[7] = blocks;
  hipLaunchKernelGGL(transpose_weights_batch_kernel, dim3(blocks), dim3(256), 0, st, j);
}
void enc_chw_conv1(const float* img, const float* wt, const float* b, float* a0, float* y1, int HW, hipStream_t st) {
  hipLaunchKernelGGL(chw_conv1_kernel, dim3((HW + 255) / 256), dim3(256), 0, st, img, wt, b, a0, y1, HW);
}
void enc_chw_to_hwc(const float* in, float* out, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(chw_to_hwc_kernel, dim3((C * HW + 255) / 256), dim3(256), 0, st, in, out, C, HW);
}
void enc_maxpool2(const float* in, float* out, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(maxpool2_kernel, dim3(((H / 2) * (W / 2) * C + 255) / 256), dim3(256), 0, st, in, out, H, W, C);
}
void enc_adaptive_avgpool(const float* in, float* out, int H, int W, int C, int S, hipStream_t st, int Hg, int yoff, int o0, int o1) {
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(((o1 - o0) * S * C + 255) / 256), dim3(256), 0, st, in, out, H, W, C, S, Hg, yoff, o0, o1);
}

static const int ENC_CIN[7] = {3, 3, 64, 64, 128, 128, 128}, ENC_COUT[7] = {3, 64, 64, 128, 128, 128, 64}, ENC_TAPS[7] = {1, 9, 9, 9, 9, 9, 1};

size_t encoder_workspace_bytes(int H, int W) {
  size_t wt = 0;
  for (int l = 0; l < 7; ++l) wt += (size_t)ENC_CIN[l] * ENC_COUT[l] * ENC_TAPS[l];
  const size_t act = (size_t)H * W * 128 > (size_t)32 * 32 * 128 ? (size_t)H * W * 128 : (size_t)32 * 32 * 128;
  size_t xcol = (size_t)H * W * 64 * 9;                                   // conv3's patch matrix; conv5's is (H/2)(W/2) x 1152
  if ((size_t)(H / 2) * (W / 2) * 128 * 9 > xcol) xcol = (size_t)(H / 2) * (W / 2) * 128 * 9;
  return (wt + 2 * act + xcol) * sizeof(float);
}

template <int TAPS, bool ACT>
static void conv(const float* in, const float* wt, const float* b, float* out, int H, int W, int cin, int cout, hipStream_t st) {
  const int groups = H * ((W + CONV_PX - 1) / CONV_PX);   // workgroups: one per group of CONV_PX pixels of a row
  hipLaunchKernelGGL((conv_kernel<TAPS, ACT>), dim3(groups, (cout + 63) / 64), dim3(256), 0, st, in, wt, b, out, H, W, cin, cout);
}

// img[3,H,W] (NCHW), weights = conv1.weight, conv1.bias, ..., conv7.weight, conv7.bias -> out[1024,64] pixel-major
int launch_encoder_forward(const float* img, int H, int W, const float* const* w, void* workspace, float* out, hipStream_t st) {
  if (H < 8 || W < 8) return set_error(-2, "encoder: image must be at least 8x8 (two 2x2 max-pools and reflection padding)");
  float* wt[7];
  float* p = (float*)workspace;
  for (int l = 0; l < 7; ++l) {
    wt[l] = p;
    const int n = ENC_CIN[l] * ENC_COUT[l] * ENC_TAPS[l];
    if (l < 2)   // only the two 3-channel layers still run on the direct kernel (which reads [cin][tap][cout])
      hipLaunchKernelGGL(transpose_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w[2 * l], wt[l], ENC_COUT[l], ENC_CIN[l], ENC_TAPS[l]);
    p += n;
  }
  const size_t act = (size_t)H * W * 128 > (size_t)32 * 32 * 128 ? (size_t)H * W * 128 : (size_t)32 * 32 * 128;
  float* a = p;
  float* b = p + act;
  float* xc = b + act;                                          // patch matrix of the layer in flight
  enc_chw_conv1(img, wt[0], w[1], a, b, H * W, st);             // NCHW -> pixel-major + conv1
  conv<9, true>(b, wt[1], w[3], a, H, W, 3, 64, st);            // conv2 + relu2
  enc_conv_gemm(9, a, xc, w[4], w[5], b, H, W, 64, 64, st);     // conv3 + relu3 (fp32 MFMA GEMM over the patch matrix)
  const int H2 = H / 2, W2 = W / 2;
  enc_conv_gemm_pooled(b, xc, w[6], w[7], a, H, W, 64, 128, st);        // max-pool + conv4 + relu4 (the pooled map lives in the patch matrix only)
  enc_conv_gemm(9, a, xc, w[8], w[9], b, H2, W2, 128, 128, st);         // conv5 + relu5
  const int H4 = H2 / 2, W4 = W2 / 2;
  enc_conv_gemm_pooled(b, xc, w[10], w[11], a, H2, W2, 128, 128, st);   // max-pool + conv6 + relu6
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3((32 * 32 * 128 + 255) / 256), dim3(256), 0, st, a, b, H4, W4, 128, 32, H4, 0, 0, 32);
  enc_conv_gemm(1, b, nullptr, w[12], w[13], out, 32, 32, 128, 64, st);   // conv7 + relu7
  return check_launch("encoder_forward");
}

}  // namespace crnerf
