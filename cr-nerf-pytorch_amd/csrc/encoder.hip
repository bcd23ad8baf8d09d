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

__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : 0.2f * v; }
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }   // ReflectionPad2d(1)

// w[cout][cin][taps] -> wt[cin][taps][cout]
__global__ void transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin, int taps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cout * cin * taps) return;
  const int o = idx / (cin * taps), r = idx % (cin * taps);
  wt[r * cout + o] = w[idx];
}

// NCHW [C,H,W] -> HWC [H*W, C]
__global__ void chw_to_hwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * HW) return;
  const int px = idx / C, c = idx % C;
  out[idx] = in[c * HW + px];
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
__global__ void adaptive_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int S) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * S * C) return;
  const int c = idx % C, px = idx / C, oy = px / S, ox = px % S;
  const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S, x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
  float s = 0.0f;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) s += in[((long)y * W + x) * C + c];
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
void enc_chw_to_hwc(const float* in, float* out, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(chw_to_hwc_kernel, dim3((C * HW + 255) / 256), dim3(256), 0, st, in, out, C, HW);
}
void enc_maxpool2(const float* in, float* out, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(maxpool2_kernel, dim3(((H / 2) * (W / 2) * C + 255) / 256), dim3(256), 0, st, in, out, H, W, C);
}
void enc_adaptive_avgpool(const float* in, float* out, int H, int W, int C, int S, hipStream_t st) {
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3((S * S * C + 255) / 256), dim3(256), 0, st, in, out, H, W, C, S);
}

static const int ENC_CIN[7] = {3, 3, 64, 64, 128, 128, 128}, ENC_COUT[7] = {3, 64, 64, 128, 128, 128, 64}, ENC_TAPS[7] = {1, 9, 9, 9, 9, 9, 1};

size_t encoder_workspace_bytes(int H, int W) {
  size_t wt = 0;
  for (int l = 0; l < 7; ++l) wt += (size_t)ENC_CIN[l] * ENC_COUT[l] * ENC_TAPS[l];
  const size_t act = (size_t)H * W * 128 > (size_t)32 * 32 * 128 ? (size_t)H * W * 128 : (size_t)32 * 32 * 128;
  return (wt + 2 * act) * sizeof(float);
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
    hipLaunchKernelGGL(transpose_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w[2 * l], wt[l], ENC_COUT[l], ENC_CIN[l], ENC_TAPS[l]);
    p += n;
  }
  const size_t act = (size_t)H * W * 128 > (size_t)32 * 32 * 128 ? (size_t)H * W * 128 : (size_t)32 * 32 * 128;
  float* a = p;
  float* b = p + act;
  hipLaunchKernelGGL(chw_to_hwc_kernel, dim3((3 * H * W + 255) / 256), dim3(256), 0, st, img, a, 3, H * W);
  conv<1, false>(a, wt[0], w[1], b, H, W, 3, 3, st);            // conv1
  conv<9, true>(b, wt[1], w[3], a, H, W, 3, 64, st);            // conv2 + relu2
  conv<9, true>(a, wt[2], w[5], b, H, W, 64, 64, st);           // conv3 + relu3
  hipLaunchKernelGGL(maxpool2_kernel, dim3(((H / 2) * (W / 2) * 64 + 255) / 256), dim3(256), 0, st, b, a, H, W, 64);
  const int H2 = H / 2, W2 = W / 2;
  conv<9, true>(a, wt[3], w[7], b, H2, W2, 64, 128, st);        // conv4 + relu4
  conv<9, true>(b, wt[4], w[9], a, H2, W2, 128, 128, st);       // conv5 + relu5
  hipLaunchKernelGGL(maxpool2_kernel, dim3(((H2 / 2) * (W2 / 2) * 128 + 255) / 256), dim3(256), 0, st, a, b, H2, W2, 128);
  const int H4 = H2 / 2, W4 = W2 / 2;
  conv<9, true>(b, wt[5], w[11], a, H4, W4, 128, 128, st);      // conv6 + relu6
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3((32 * 32 * 128 + 255) / 256), dim3(256), 0, st, a, b, H4, W4, 128, 32);
  conv<1, true>(b, wt[6], w[13], out, 32, 32, 128, 64, st);     // conv7 + relu7
  return check_launch("encoder_forward");
}

}  // namespace crnerf
