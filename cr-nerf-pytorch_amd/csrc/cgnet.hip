// Building blocks of the transient-mask network: Context_Guided_Network(classes=1, M=2, N=2, input_channel=3),
// models/lightweight_seg.py:274-368, applied to the 1/8-scale photo once per training step (train_mask_grid_sample.py:170-176).
// Five fused operators, forward and backward, on NCHW fp32 with batch 1:
//   conv2d           3x3 / 1x1, stride 1|2, zero padding, dilation, dense or depth-wise           (:12-140 Conv* classes)
//   bn_prelu         BatchNorm2d(eps=1e-3, batch statistics in training) + per-channel PReLU       (ConvBNPReLU, BNPReLU)
//   avgpool3s2       AvgPool2d(3, stride=2, padding=1), padding counted                            (InputInjection :258-270)
//   fglo             x * sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2)                                 (FGlo :143-162)
//   bilinear_gather  F.interpolate(..., mode='bilinear', align_corners=False) evaluated at a list of output pixels,
//                    optionally followed by sigmoid                                               (:366-367; train...py:172-175)
// The maps are tiny (<= 32x40x32 after the first stride-2 layer), so the kernels are written for obviousness, one thread
// (or one small block) per output element; nothing here is near a roofline and nothing needs to be.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace crnerf {

// ---------------------------------------------------------------- conv2d
__global__ void cg_conv_fwd_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.cout * g.Ho * g.Wo) return;
  const int ox = idx % g.Wo, oy = (idx / g.Wo) % g.Ho, co = idx / (g.Wo * g.Ho);
  const int cpg = g.depthwise ? 1 : g.cin;                    // input channels per output channel
  float acc = 0.0f;
  for (int c = 0; c < cpg; ++c) {
    const int ci = g.depthwise ? co : c;
    for (int ky = 0; ky < g.k; ++ky) {
      const int iy = oy * g.stride + ky * g.dil - g.pad;
      if (iy < 0 || iy >= g.H) continue;
      for (int kx = 0; kx < g.k; ++kx) {
        const int ix = ox * g.stride + kx * g.dil - g.pad;
        if (ix < 0 || ix >= g.W) continue;
        acc = fmaf(w[((co * cpg + c) * g.k + ky) * g.k + kx], x[(ci * g.H + iy) * g.W + ix], acc);
      }
    }
  }
  y[idx] = acc;
}

// Dense convolutions with many input channels (35 ... 256 x 9 taps): one WAVE per output element, lanes stride over the
// (ci, ky, kx) products (weights read coalesced) and a butterfly adds them up -- the maps are tiny, so what costs is the
// length of one thread's serial loop (1,179 dependent FMAs for level3_0), not arithmetic.
__global__ __launch_bounds__(256) void cg_conv_fwd_wave_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
  const int out = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (out >= g.cout * g.Ho * g.Wo) return;
  const int ox = out % g.Wo, oy = (out / g.Wo) % g.Ho, co = out / (g.Wo * g.Ho);
  const int kk = g.k * g.k, n = g.cin * kk;
  float acc = 0.0f;
  for (int idx = lane; idx < n; idx += 64) {
    const int ci = idx / kk, t = idx % kk;
    const int iy = oy * g.stride + (t / g.k) * g.dil - g.pad, ix = ox * g.stride + (t % g.k) * g.dil - g.pad;
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) acc = fmaf(w[(long)co * n + idx], x[(ci * g.H + iy) * g.W + ix], acc);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) y[out] = acc;
}

// dx[ci][iy][ix]: lanes stride over (co, ky, kx)
__global__ __launch_bounds__(256) void cg_conv_dgrad_wave_kernel(ConvGeom g, const float* __restrict__ w, const float* __restrict__ dy, float* __restrict__ dx) {
  const int in = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (in >= g.cin * g.H * g.W) return;
  const int ix = in % g.W, iy = (in / g.W) % g.H, ci = in / (g.W * g.H);
  const int kk = g.k * g.k, n = g.cout * kk;
  float acc = 0.0f;
  for (int idx = lane; idx < n; idx += 64) {
    const int co = idx / kk, t = idx % kk;
    const int ty = iy + g.pad - (t / g.k) * g.dil, tx = ix + g.pad - (t % g.k) * g.dil;
    if (ty < 0 || tx < 0 || ty % g.stride || tx % g.stride) continue;
    const int oy = ty / g.stride, ox = tx / g.stride;
    if (oy < g.Ho && ox < g.Wo) acc = fmaf(w[((long)co * g.cin + ci) * kk + t], dy[(co * g.Ho + oy) * g.Wo + ox], acc);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) dx[in] = g.accum ? dx[in] + acc : acc;
}

__global__ void cg_conv_dgrad_kernel(ConvGeom g, const float* __restrict__ w, const float* __restrict__ dy, float* __restrict__ dx) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.cin * g.H * g.W) return;
  const int ix = idx % g.W, iy = (idx / g.W) % g.H, ci = idx / (g.W * g.H);
  const int cpg = g.depthwise ? 1 : g.cin;
  float acc = 0.0f;
  for (int ky = 0; ky < g.k; ++ky) {
    const int ty = iy + g.pad - ky * g.dil;
    if (ty < 0 || ty % g.stride) continue;
    const int oy = ty / g.stride;
    if (oy >= g.Ho) continue;
    for (int kx = 0; kx < g.k; ++kx) {
      const int tx = ix + g.pad - kx * g.dil;
      if (tx < 0 || tx % g.stride) continue;
      const int ox = tx / g.stride;
      if (ox >= g.Wo) continue;
      if (g.depthwise) acc = fmaf(w[(ci * g.k + ky) * g.k + kx], dy[(ci * g.Ho + oy) * g.Wo + ox], acc);
      else
        for (int co = 0; co < g.cout; ++co) acc = fmaf(w[((co * cpg + ci) * g.k + ky) * g.k + kx], dy[(co * g.Ho + oy) * g.Wo + ox], acc);
    }
  }
  dx[idx] = g.accum ? dx[idx] + acc : acc;
}

// one 64-lane block per weight element: sum over output pixels
__global__ __launch_bounds__(64) void cg_conv_wgrad_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw) {
  const int cpg = g.depthwise ? 1 : g.cin;
  const int widx = blockIdx.x;
  const int kx = widx % g.k, ky = (widx / g.k) % g.k, c = (widx / (g.k * g.k)) % cpg, co = widx / (g.k * g.k * cpg);
  const int ci = g.depthwise ? co : c;
  float acc = 0.0f;
  for (int p = threadIdx.x; p < g.Ho * g.Wo; p += 64) {
    const int oy = p / g.Wo, ox = p % g.Wo;
    const int iy = oy * g.stride + ky * g.dil - g.pad, ix = ox * g.stride + kx * g.dil - g.pad;
    if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
    acc = fmaf(dy[(co * g.Ho + oy) * g.Wo + ox], x[(ci * g.H + iy) * g.W + ix], acc);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (threadIdx.x == 0) dw[widx] = acc;
}

int launch_cg_conv_forward(const ConvGeom& g, const float* x, const float* w, float* y, hipStream_t st) {
  const int n = g.cout * g.Ho * g.Wo;
  if (!g.depthwise && g.cin * g.k * g.k >= 128) hipLaunchKernelGGL(cg_conv_fwd_wave_kernel, dim3((n + 3) / 4), dim3(256), 0, st, g, x, w, y);
  else hipLaunchKernelGGL(cg_conv_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g, x, w, y);
  return check_launch("cg_conv_forward");
}
int launch_cg_conv_backward(const ConvGeom& g, const float* x, const float* w, const float* dy, float* dx, float* dw, hipStream_t st) {
  if (dx) {
    const int n = g.cin * g.H * g.W;
    if (!g.depthwise && g.cout * g.k * g.k >= 128) hipLaunchKernelGGL(cg_conv_dgrad_wave_kernel, dim3((n + 3) / 4), dim3(256), 0, st, g, w, dy, dx);
    else hipLaunchKernelGGL(cg_conv_dgrad_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g, w, dy, dx);
  }
  const int nw = g.cout * (g.depthwise ? 1 : g.cin) * g.k * g.k;
  hipLaunchKernelGGL(cg_conv_wgrad_kernel, dim3(nw), dim3(64), 0, st, g, x, dy, dw);
  return check_launch("cg_conv_backward");
}

// ---------------------------------------------------------------- BatchNorm2d + PReLU (one block per channel)
__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// training: batch statistics (biased variance for the normalisation), written to mean / invstd; eval: mean / invstd given
__device__ __forceinline__ void cg_bn_prelu_fwd_body(const int c, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ alpha, float* __restrict__ mean, float* __restrict__ invstd,
                                                              float* __restrict__ var_unbiased, float* __restrict__ y, int HW, float eps, int training,
                                                              float* __restrict__ run_mean, float* __restrict__ run_var, long long* __restrict__ n_tracked,
                                                              float momentum, float* red) {
  const float* xc = x + (long)c * HW;
  float m, is;
  if (training) {
    float s = 0.0f;
    for (int p = threadIdx.x; p < HW; p += 256) s += xc[p];
    m = block_sum256(s, red) / (float)HW;
    float q = 0.0f;
    for (int p = threadIdx.x; p < HW; p += 256) { const float d = xc[p] - m; q += d * d; }
    const float ss = block_sum256(q, red);
    is = rsqrtf(ss / (float)HW + eps);
    if (threadIdx.x == 0) {
      const float vu = HW > 1 ? ss / (float)(HW - 1) : ss;
      mean[c] = m; invstd[c] = is; var_unbiased[c] = vu;
      if (run_mean) {   // nn.BatchNorm2d's momentum update of the running buffers (torch: running.mul_(1 - momentum).add_(stat, alpha=momentum))
        run_mean[c] = fmaf(momentum, m, run_mean[c] * (1.0f - momentum));
        run_var[c] = fmaf(momentum, vu, run_var[c] * (1.0f - momentum));
        if (c == 0 && n_tracked) n_tracked[0] += 1;
      }
    }
  } else {
    m = mean[c];
    is = invstd[c];
  }
  const float ga = gamma[c], be = beta[c], al = alpha[c];
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float z = (xc[p] - m) * is * ga + be;
    y[(long)c * HW + p] = z > 0.0f ? z : al * z;
  }
}

__global__ __launch_bounds__(256) void cg_bn_prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ alpha, float* __restrict__ mean, float* __restrict__ invstd,
                                                              float* __restrict__ var_unbiased, float* __restrict__ y, int HW, float eps, int training,
                                                              float* __restrict__ run_mean, float* __restrict__ run_var, long long* __restrict__ n_tracked,
                                                              float momentum) {
  __shared__ float red[4];
  cg_bn_prelu_fwd_body(blockIdx.x, x, gamma, beta, alpha, mean, invstd, var_unbiased, y, HW, eps, training, run_mean, run_var, n_tracked, momentum, red);
}

// F_loc / F_sur -> concatenation -> BatchNorm + PReLU of a ContextGuidedBlock[_Down] in ONE launch (training chain): channel c of the
// concatenation is a depth-wise 3x3 convolution of input channel c % n (dilation 1 for c < n, `dil` above), so the block of channel c
// computes its own pre-BN map -- kept in `cat`, the backward reads it -- in cg_conv_fwd_kernel's order of products, and the statistics
// and the normalisation are cg_bn_prelu_fwd_body on it (every thread reads back the pixels it wrote itself).
__global__ __launch_bounds__(256) void cg_dwpair_bn_prelu_fwd_kernel(const float* __restrict__ yin, const float* __restrict__ w_loc, const float* __restrict__ w_sur,
                                                                     int n, int H, int W, int dil, float* cat, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ alpha, float* __restrict__ mean,
                                                                     float* __restrict__ invstd, float* __restrict__ var_unbiased, float* __restrict__ z, float eps,
                                                                     float* __restrict__ run_mean, float* __restrict__ run_var, long long* __restrict__ n_tracked,
                                                                     float momentum) {
  __shared__ float red[4];
  const int c = blockIdx.x, src = c % n, d = c >= n ? dil : 1, HW = H * W;
  const float* w = (c >= n ? w_sur : w_loc) + src * 9;
  const float* xs = yin + (long)src * HW;
  float* xc = cat + (long)c * HW;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int ox = p % W, oy = p / W;
    float acc = 0.0f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy + (ky - 1) * d;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox + (kx - 1) * d;
        if (ix < 0 || ix >= W) continue;
        acc = fmaf(w[ky * 3 + kx], xs[iy * W + ix], acc);
      }
    }
    xc[p] = acc;
  }
  __syncthreads();
  cg_bn_prelu_fwd_body(c, cat, gamma, beta, alpha, mean, invstd, var_unbiased, z, HW, eps, 1, run_mean, run_var, n_tracked, momentum, red);
}

// training == 0 (eval): mean / invstd are constants, so dx = dz * gamma * invstd
__device__ __forceinline__ void cg_bn_prelu_bwd_body(const int c, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ alpha, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dalpha, int HW, int training, float* red) {
  const float* xc = x + (long)c * HW;
  const float* dyc = dy + (long)c * HW;
  const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c], al = alpha[c];
  float s1 = 0.0f, s2 = 0.0f, sa = 0.0f;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float xh = (xc[p] - m) * is, z = xh * ga + be, g = dyc[p];
    const float dz = z > 0.0f ? g : al * g;
    s1 += dz;
    s2 += dz * xh;
    sa += z > 0.0f ? 0.0f : g * z;
  }
  s1 = block_sum256(s1, red);
  s2 = block_sum256(s2, red);
  sa = block_sum256(sa, red);
  if (threadIdx.x == 0) { dgamma[c] = s2; dbeta[c] = s1; dalpha[c] = sa; }
  const float inv_n = 1.0f / (float)HW;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float xh = (xc[p] - m) * is, z = xh * ga + be, g = dyc[p];
    const float dz = z > 0.0f ? g : al * g;
    dx[(long)c * HW + p] = training ? ga * is * (dz - s1 * inv_n - xh * s2 * inv_n) : ga * is * dz;
  }
}

__global__ __launch_bounds__(256) void cg_bn_prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ alpha, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dalpha, int HW, int training) {
  __shared__ float red[4];
  cg_bn_prelu_bwd_body(blockIdx.x, x, gamma, beta, alpha, mean, invstd, dy, dx, dgamma, dbeta, dalpha, HW, training, red);
}

// The backward of cg_dwpair_bn_prelu_fwd_kernel, one block per INPUT channel s: BatchNorm + PReLU backward of concatenation channels s
// (F_loc) and s + n (F_sur) into d_cat, then both depth-wise convolutions' weight gradients (nine taps each) and the sum of their data
// gradients -- d_y[s] is written once, F_loc's contribution first, as the two accumulating launches did.
__global__ __launch_bounds__(256) void cg_dwpair_bn_prelu_bwd_kernel(const float* __restrict__ yin, const float* __restrict__ w_loc, const float* __restrict__ w_sur,
                                                                     int n, int H, int W, int dil, const float* __restrict__ cat, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ alpha, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, const float* __restrict__ dz, float* d_cat,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dalpha,
                                                                     float* __restrict__ dw_loc, float* __restrict__ dw_sur, float* __restrict__ d_y) {
  __shared__ float red[4];
  const int s = blockIdx.x, HW = H * W;
  cg_bn_prelu_bwd_body(s, cat, gamma, beta, alpha, mean, invstd, dz, d_cat, dgamma, dbeta, dalpha, HW, 1, red);
  cg_bn_prelu_bwd_body(s + n, cat, gamma, beta, alpha, mean, invstd, dz, d_cat, dgamma, dbeta, dalpha, HW, 1, red);
  __syncthreads();                                           // d_cat[s], d_cat[s + n] are complete (and visible) for the whole block
  const float* xs = yin + (long)s * HW;
  for (int h = 0; h < 2; ++h) {
    const float* dc = d_cat + (long)(s + h * n) * HW;
    const int d = h ? dil : 1;
    float* dw = (h ? dw_sur : dw_loc) + s * 9;
    for (int t = 0; t < 9; ++t) {
      const int offy = (t / 3 - 1) * d, offx = (t % 3 - 1) * d;
      float acc = 0.0f;
      for (int p = threadIdx.x; p < HW; p += 256) {
        const int iy = p / W + offy, ix = p % W + offx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        acc = fmaf(dc[p], xs[iy * W + ix], acc);
      }
      acc = block_sum256(acc, red);
      if (threadIdx.x == 0) dw[t] = acc;
    }
  }
  for (int q = threadIdx.x; q < HW; q += 256) {
    const int ix = q % W, iy = q / W;
    float tot = 0.0f;
    for (int h = 0; h < 2; ++h) {
      const float* dc = d_cat + (long)(s + h * n) * HW;
      const float* w = (h ? w_sur : w_loc) + s * 9;
      const int d = h ? dil : 1;
      float acc = 0.0f;
      for (int ky = 0; ky < 3; ++ky) {                       // cg_conv_dgrad_kernel's order (stride 1, pad = d): output row oy = iy + d - ky d
        const int oy = iy + d - ky * d;
        if (oy < 0 || oy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int ox = ix + d - kx * d;
          if (ox < 0 || ox >= W) continue;
          acc = fmaf(w[ky * 3 + kx], dc[oy * W + ox], acc);
        }
      }
      tot = h ? tot + acc : acc;
    }
    d_y[(long)s * HW + q] = tot;
  }
}

int launch_cg_bn_prelu_forward(const float* x, const float* gamma, const float* beta, const float* alpha, float* mean, float* invstd, float* var_unbiased,
                               float* y, int C, int HW, float eps, int training, hipStream_t st, float* run_mean, float* run_var, long long* n_tracked,
                               float momentum) {
  hipLaunchKernelGGL(cg_bn_prelu_fwd_kernel, dim3(C), dim3(256), 0, st, x, gamma, beta, alpha, mean, invstd, var_unbiased, y, HW, eps, training, run_mean,
                     run_var, n_tracked, momentum);
  return check_launch("cg_bn_prelu_forward");
}
int launch_cg_bn_prelu_backward(const float* x, const float* gamma, const float* beta, const float* alpha, const float* mean, const float* invstd,
                                const float* dy, float* dx, float* dgamma, float* dbeta, float* dalpha, int C, int HW, int training, hipStream_t st) {
  hipLaunchKernelGGL(cg_bn_prelu_bwd_kernel, dim3(C), dim3(256), 0, st, x, gamma, beta, alpha, mean, invstd, dy, dx, dgamma, dbeta, dalpha, HW, training);
  return check_launch("cg_bn_prelu_backward");
}

int launch_cg_dwpair_bn_prelu_forward(const float* y, const float* w_loc, const float* w_sur, int n, int H, int W, int dil, float* cat, const float* gamma,
                                      const float* beta, const float* alpha, float* mean, float* invstd, float* var_unbiased, float* z, float eps,
                                      float* run_mean, float* run_var, long long* n_tracked, float momentum, hipStream_t st) {
  hipLaunchKernelGGL(cg_dwpair_bn_prelu_fwd_kernel, dim3(2 * n), dim3(256), 0, st, y, w_loc, w_sur, n, H, W, dil, cat, gamma, beta, alpha, mean, invstd,
                     var_unbiased, z, eps, run_mean, run_var, n_tracked, momentum);
  return check_launch("cg_dwpair_bn_prelu_forward");
}
int launch_cg_dwpair_bn_prelu_backward(const float* y, const float* w_loc, const float* w_sur, int n, int H, int W, int dil, const float* cat, const float* gamma,
                                       const float* beta, const float* alpha, const float* mean, const float* invstd, const float* dz, float* d_cat,
                                       float* dgamma, float* dbeta, float* dalpha, float* dw_loc, float* dw_sur, float* d_y, hipStream_t st) {
  hipLaunchKernelGGL(cg_dwpair_bn_prelu_bwd_kernel, dim3(n), dim3(256), 0, st, y, w_loc, w_sur, n, H, W, dil, cat, gamma, beta, alpha, mean, invstd, dz, d_cat,
                     dgamma, dbeta, dalpha, dw_loc, dw_sur, d_y);
  return check_launch("cg_dwpair_bn_prelu_backward");
}

// ---------------------------------------------------------------- AvgPool2d(3, stride 2, padding 1), count_include_pad
__global__ void cg_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, int Ho, int Wo) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * Ho * Wo) return;
  const int ox = idx % Wo, oy = (idx / Wo) % Ho, c = idx / (Wo * Ho);
  float s = 0.0f;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) s += x[(c * H + iy) * W + ix];
    }
  y[idx] = s / 9.0f;
}
__global__ void cg_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int C, int H, int W, int Ho, int Wo) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * H * W) return;
  const int ix = idx % W, iy = (idx / W) % H, c = idx / (W * H);
  float s = 0.0f;
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || (ty & 1) || ty / 2 >= Ho) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || (tx & 1) || tx / 2 >= Wo) continue;
      s += dy[(c * Ho + ty / 2) * Wo + tx / 2];
    }
  }
  dx[idx] = s / 9.0f;
}
int launch_cg_avgpool(const float* in, float* out, int C, int H, int W, int backward, hipStream_t st) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (!backward) hipLaunchKernelGGL(cg_avgpool_fwd_kernel, dim3((C * Ho * Wo + 255) / 256), dim3(256), 0, st, in, out, C, H, W, Ho, Wo);
  else hipLaunchKernelGGL(cg_avgpool_bwd_kernel, dim3((C * H * W + 255) / 256), dim3(256), 0, st, in, out, C, H, W, Ho, Wo);
  return check_launch("cg_avgpool");
}

// ---------------------------------------------------------------- FGlo: y = x * s[c], s = sigmoid(W2 relu(W1 mean(x) + b1) + b2)
// stats[0:C] = channel means, [C:C+R] = hidden (post-relu), [C+R:2C+R] = gates s
__global__ __launch_bounds__(256) void cg_chan_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int HW, float scale) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.0f;
  for (int p = threadIdx.x; p < HW; p += 256) s += b ? a[(long)c * HW + p] * b[(long)c * HW + p] : a[(long)c * HW + p];
  s = block_sum256(s, red);
  if (threadIdx.x == 0) out[c] = s * scale;
}
__global__ __launch_bounds__(256) void cg_fglo_fc_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ stats, int C, int R) {
  __shared__ float hid[64];
  const int t = threadIdx.x;
  if (t < R) {
    float a = b1[t];
    for (int c = 0; c < C; ++c) a = fmaf(w1[t * C + c], stats[c], a);
    hid[t] = a > 0.0f ? a : 0.0f;
    stats[C + t] = hid[t];
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    float a = b2[c];
    for (int r = 0; r < R; ++r) a = fmaf(w2[c * R + r], hid[r], a);
    stats[C + R + c] = 1.0f / (1.0f + expf(-a));
  }
}
// res (may be null): the block's residual input, added after the product is rounded -- what `x + F_glo(...)` gives (ContextGuidedBlock, add=True)
__global__ void cg_fglo_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ extra, float* __restrict__ y, int C,
                                     int HW, const float* __restrict__ res) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * HW) return;
  const int c = idx / HW;
  const float v = x[idx] * gate[c] + (extra ? extra[c] : 0.0f);
  y[idx] = res ? res[idx] + v : v;
}
__global__ void cg_add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) dst[idx] += src[idx];
}
int launch_cg_add_inplace(float* dst, const float* src, int n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cg_add_inplace_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, n);
  return check_launch("cg_add_inplace");
}
// ds[c] = sum dy*x  ->  through sigmoid, fc2, relu, fc1  ->  dm[c] / HW added to every pixel of channel c
__global__ __launch_bounds__(256) void cg_fglo_fc_bwd_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ stats,
                                                             const float* __restrict__ dsum, float* __restrict__ dw1, float* __restrict__ db1,
                                                             float* __restrict__ dw2, float* __restrict__ db2, float* __restrict__ dmean_over_hw, int C, int R,
                                                             float inv_hw) {
  __shared__ float dp2[256], dp1[64];
  const int t = threadIdx.x;
  const float* m = stats;
  const float* hid = stats + C;
  const float* s = stats + C + R;
  for (int c = t; c < C; c += 256) {
    const float g = dsum[c] * s[c] * (1.0f - s[c]);
    dp2[c] = g;
    db2[c] = g;
    for (int r = 0; r < R; ++r) dw2[c * R + r] = g * hid[r];
  }
  __syncthreads();
  if (t < R) {
    float dh = 0.0f;
    for (int c = 0; c < C; ++c) dh = fmaf(w2[c * R + t], dp2[c], dh);
    const float g = hid[t] > 0.0f ? dh : 0.0f;
    dp1[t] = g;
    db1[t] = g;
    for (int c = 0; c < C; ++c) dw1[t * C + c] = g * m[c];
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    float dm = 0.0f;
    for (int r = 0; r < R; ++r) dm = fmaf(w1[r * C + c], dp1[r], dm);
    dmean_over_hw[c] = dm * inv_hw;
  }
}
int launch_cg_fglo_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* stats, float* y, int C, int R, int HW,
                           hipStream_t st, const float* residual) {
  if (C > 256 || R > 64) return set_error(-2, "fglo: C <= 256 and C/reduction <= 64 expected");
  hipLaunchKernelGGL(cg_chan_reduce_kernel, dim3(C), dim3(256), 0, st, x, (const float*)nullptr, stats, HW, 1.0f / (float)HW);
  hipLaunchKernelGGL(cg_fglo_fc_kernel, dim3(1), dim3(256), 0, st, w1, b1, w2, b2, stats, C, R);
  hipLaunchKernelGGL(cg_fglo_scale_kernel, dim3((C * HW + 255) / 256), dim3(256), 0, st, x, stats + C + R, (const float*)nullptr, y, C, HW, residual);
  return check_launch("cg_fglo_forward");
}
// scratch: 2C floats
int launch_cg_fglo_backward(const float* x, const float* w1, const float* w2, const float* stats, const float* dy, float* scratch, float* dx, float* dw1,
                            float* db1, float* dw2, float* db2, int C, int R, int HW, hipStream_t st) {
  float* dsum = scratch;
  float* dmean = scratch + C;
  hipLaunchKernelGGL(cg_chan_reduce_kernel, dim3(C), dim3(256), 0, st, dy, x, dsum, HW, 1.0f);
  hipLaunchKernelGGL(cg_fglo_fc_bwd_kernel, dim3(1), dim3(256), 0, st, w1, w2, stats, dsum, dw1, db1, dw2, db2, dmean, C, R, 1.0f / (float)HW);
  hipLaunchKernelGGL(cg_fglo_scale_kernel, dim3((C * HW + 255) / 256), dim3(256), 0, st, dy, stats + C + R, dmean, dx, C, HW, (const float*)nullptr);
  return check_launch("cg_fglo_backward");
}

// ---------------------------------------------------------------- bilinear (align_corners = False) at listed output pixels
struct Taps { int i00, i01, i10, i11; float w00, w01, w10, w11; };
__device__ __forceinline__ Taps bilinear_taps(long pix, int h, int w, int Ho, int Wo) {
  const int oy = (int)(pix / Wo), ox = (int)(pix % Wo);
  float sy = ((float)oy + 0.5f) * ((float)h / (float)Ho) - 0.5f, sx = ((float)ox + 0.5f) * ((float)w / (float)Wo) - 0.5f;   // area_pixel_compute_source_index
  sy = sy < 0.0f ? 0.0f : sy;
  sx = sx < 0.0f ? 0.0f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  Taps t;
  t.i00 = y0 * w + x0; t.i01 = y0 * w + x1; t.i10 = y1 * w + x0; t.i11 = y1 * w + x1;
  t.w00 = (1.0f - ly) * (1.0f - lx); t.w01 = (1.0f - ly) * lx; t.w10 = ly * (1.0f - lx); t.w11 = ly * lx;
  return t;
}
__global__ void cg_bilinear_fwd_kernel(const float* __restrict__ in, const long* __restrict__ idx, float* __restrict__ out, long n, int h, int w, int Ho,
                                       int Wo, int sigmoid) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Taps t = bilinear_taps(idx ? idx[i] : i, h, w, Ho, Wo);
  const float v = t.w00 * in[t.i00] + t.w01 * in[t.i01] + t.w10 * in[t.i10] + t.w11 * in[t.i11];
  out[i] = sigmoid ? 1.0f / (1.0f + expf(-v)) : v;
}
// d_in must be zeroed; the scatter uses float atomics (a few thousand adds into <= a few thousand cells)
__global__ void cg_bilinear_bwd_kernel(const float* __restrict__ out, const float* __restrict__ d_out, const long* __restrict__ idx, float* __restrict__ d_in,
                                       long n, int h, int w, int Ho, int Wo, int sigmoid) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Taps t = bilinear_taps(idx ? idx[i] : i, h, w, Ho, Wo);
  float g = d_out[i];
  if (sigmoid) g *= out[i] * (1.0f - out[i]);
  atomicAdd(d_in + t.i00, g * t.w00);
  atomicAdd(d_in + t.i01, g * t.w01);
  atomicAdd(d_in + t.i10, g * t.w10);
  atomicAdd(d_in + t.i11, g * t.w11);
}
int launch_cg_bilinear(const float* in, const long* idx, float* out, long n, int h, int w, int Ho, int Wo, int sigmoid, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cg_bilinear_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, idx, out, n, h, w, Ho, Wo, sigmoid);
  return check_launch("cg_bilinear");
}
int launch_cg_bilinear_backward(const float* out, const float* d_out, const long* idx, float* d_in, long n, int h, int w, int Ho, int Wo, int sigmoid,
                                hipStream_t st) {
  if (hipMemsetAsync(d_in, 0, (size_t)h * w * sizeof(float), st) != hipSuccess) return set_error(-10, "hipMemsetAsync failed");
  if (n > 0) hipLaunchKernelGGL(cg_bilinear_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, d_out, idx, d_in, n, h, w, Ho, Wo, sigmoid);
  return check_launch("cg_bilinear_backward");
}

}  // namespace crnerf
