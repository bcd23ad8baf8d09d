// Training-side neighbours of the rendering path: the CRNeRF loss with its gradient, and the
// grid-sample batcher that cuts a training batch out of the HBM-resident ray / rgb buffers.
//
// Reference: CRNeRFLoss.forward, losses.py:49-78 (mask_regularize :80-91, _l2_regularize :93-96);
//            PhototourismDataset.__getitem__ (train), datasets/phototourism_mask_grid_sample.py:241-275.
// Both are HBM-bound streaming kernels: the loss reads 4*(3+3+3+1) B per ray once (forward) and once more (backward);
// the batcher gathers 9 + 3 floats per sample from random rows of buffers that stay resident in the 288 GB of HBM.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace crnerf {

constexpr int LOSS_TERMS = 7;       // kl_a, rec_a_random, c_l, content_constraint, r_ms, r_md, f_l (oracle LOSS_KEYS order)
constexpr int LOSS_BLOCKS = 256;

__device__ __forceinline__ float ld2(const float* p, long r, int c, long sr, long sc) { return p[r * sr + c * sc]; }

__global__ __launch_bounds__(256) void loss_partial_kernel(LossArgs a, float* __restrict__ partial) {
  float s[LOSS_TERMS] = {0, 0, 0, 0, 0, 0, 0};
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
  for (long r = tid; r < a.R; r += nthr) {
    const float m = a.mask ? a.mask[r] : 0.0f;
    float ec = 0.0f, ef = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = ld2(a.tgt, r, c, a.tg_sr, a.tg_sc);
      const float dc = ld2(a.rgb_c, r, c, a.rc_sr, a.rc_sc) - t;
      ec += dc * dc;
      if (a.rgb_f) {
        const float df = ld2(a.rgb_f, r, c, a.rf_sr, a.rf_sc) - t;
        ef += df * df;
      }
    }
    s[2] += (1.0f - m) * ec;
    s[6] += (1.0f - m) * ef;
    if (a.mask) {
      s[4] += m * m;
      const float q = (m - 0.5f) * (m - 0.5f) + 0.02f;
      s[5] += 1.0f / q;
    }
  }
  if (a.a)
    for (long i = tid; i < a.n_a; i += nthr) s[0] += a.a[i] * a.a[i];
  if (a.a_rec)
    for (long i = tid; i < a.n_rec; i += nthr) {
      const float d = a.a_rand[i] - a.a_rec[i];
      s[1] += a.mse_a ? d * d : fabsf(d);
    }
  if (a.c_wo)
    for (long i = tid; i < a.n_c; i += nthr) {
      const float d = a.c_wo[i] - a.c_with[i];
      s[3] += d * d;
    }
  __shared__ float red[4][LOSS_TERMS];
#pragma unroll
  for (int k = 0; k < LOSS_TERMS; ++k) {
    float v = s[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < LOSS_TERMS) partial[blockIdx.x * LOSS_TERMS + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void loss_final_kernel(const float* __restrict__ partial, int nblk, LossScales sc, float* __restrict__ losses) {
  const int k = threadIdx.x;
  if (k >= LOSS_TERMS) return;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += (double)partial[b * LOSS_TERMS + k];
  losses[k] = (float)(t * (double)sc.s[k]);
}

// upstream[k] = d(total)/d(losses[k]); every gradient is elementwise in its input
__global__ __launch_bounds__(256) void loss_backward_kernel(LossArgs a, LossScales sc, const float* __restrict__ upstream, LossGrads g) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
  float u[LOSS_TERMS];
#pragma unroll
  for (int k = 0; k < LOSS_TERMS; ++k) u[k] = upstream[k] * sc.s[k];
  for (long r = tid; r < a.R; r += nthr) {
    const float m = a.mask ? a.mask[r] : 0.0f;
    float ef = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = ld2(a.tgt, r, c, a.tg_sr, a.tg_sc);
      const float dc = ld2(a.rgb_c, r, c, a.rc_sr, a.rc_sc) - t;
      if (g.d_rgb_c) g.d_rgb_c[r * 3 + c] = u[2] * 2.0f * (1.0f - m) * dc;       // mask.detach() in c_l, losses.py:63
      if (a.rgb_f) {
        const float df = ld2(a.rgb_f, r, c, a.rf_sr, a.rf_sc) - t;
        ef += df * df;
        if (g.d_rgb_f) g.d_rgb_f[r * 3 + c] = u[6] * 2.0f * (1.0f - m) * df;
      }
    }
    if (a.mask && g.d_mask) {
      const float e = m - 0.5f, q = e * e + 0.02f;
      g.d_mask[r] = a.rgb_f ? u[4] * 2.0f * m - u[5] * 2.0f * e / (q * q) - u[6] * ef : 0.0f;   // r_ms, r_md, f_l exist with rgb_fine only (:68-72)
    }
  }
  if (a.a && g.d_a)
    for (long i = tid; i < a.n_a; i += nthr) g.d_a[i] = u[0] * 2.0f * a.a[i];
  if (a.a_rec && g.d_a_rec)
    for (long i = tid; i < a.n_rec; i += nthr) {
      const float d = a.a_rand[i] - a.a_rec[i];
      g.d_a_rec[i] = a.mse_a ? -u[1] * 2.0f * d : -u[1] * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    }
  if (a.c_wo && (g.d_c_wo || g.d_c_with))          // either side of the content pair may be the only one that needs a gradient
    for (long i = tid; i < a.n_c; i += nthr) {
      const float d = u[3] * 2.0f * (a.c_wo[i] - a.c_with[i]);
      if (g.d_c_wo) g.d_c_wo[i] = d;
      if (g.d_c_with) g.d_c_with[i] = -d;
    }
}

static int loss_blocks(const LossArgs& a) {
  long n = a.R;
  if (a.n_a > n) n = a.n_a;
  if (a.n_rec > n) n = a.n_rec;
  if (a.n_c > n) n = a.n_c;
  const long b = (n + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > LOSS_BLOCKS ? LOSS_BLOCKS : b));
}

size_t loss_workspace_bytes() { return (size_t)LOSS_BLOCKS * LOSS_TERMS * sizeof(float); }

int launch_loss_forward(const LossArgs& a, const LossScales& sc, float* losses, void* workspace, hipStream_t stream) {
  const int nblk = loss_blocks(a);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(nblk), dim3(256), 0, stream, a, (float*)workspace);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, nblk, sc, losses);
  return check_launch("loss_forward");
}

int launch_loss_backward(const LossArgs& a, const LossScales& sc, const float* upstream, const LossGrads& g, hipStream_t stream) {
  hipLaunchKernelGGL(loss_backward_kernel, dim3(loss_blocks(a) * 4), dim3(256), 0, stream, a, sc, upstream, g);
  return check_launch("loss_backward");
}

// ---------------------------------------------------------------- grid-sample batcher
// sample n = j * side + i  <->  lattice node (i, j): w from w_lin[i], h from h_lin[j]  (meshgrid 'ij' + permute(1, 0), :249-262)
__global__ __launch_bounds__(256) void grid_batch_kernel(BatchArgs a) {
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.side * a.side;
  if (n >= total) return;
  const int i = (int)(n % a.side), j = (int)(n / a.side);
  const float w_sb = a.w_lin[i] * a.scale + a.w_offset;       // separate mul / add, as the reference's tensor ops (:257-258)
  const float h_sb = a.h_lin[j] * a.scale + a.h_offset;
  const float w = floorf(w_sb * (float)a.img_w), h = floorf(h_sb * (float)a.img_h);
  const long pt = (long)(w + h * (float)a.img_w);             // fp32 like (w + h * img_w) before .long() (:262); exact below 2^24 pixels
  const long row = pt + a.row_offset;
  const float* src = a.all_rays + row * a.ray_stride;
#pragma unroll
  for (int c = 0; c < 8; ++c) a.rays[n * 8 + c] = src[c];
  a.ts[n] = (long)src[8];                                     // .long() of the stored float id (:268)
  const float* rgb = a.all_rgbs + row * 3;
  a.rgbs[n * 3 + 0] = rgb[0]; a.rgbs[n * 3 + 1] = rgb[1]; a.rgbs[n * 3 + 2] = rgb[2];
  a.rgb_idx[n] = pt;
  a.uv[n * 2 + 0] = h_sb;
  a.uv[n * 2 + 1] = w_sb;
}

int launch_grid_batch(const BatchArgs& a, hipStream_t stream) {
  const long total = (long)a.side * a.side;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(grid_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
  return check_launch("grid_batch_kernel");
}

// ---------------------------------------------------------------- Adam over one flat parameter buffer
// The reference's optimiser is torch.optim.Adam(parameters, lr, eps=1e-8, weight_decay) (utils/__init__.py:24-33) stepped once per
// batch by Lightning (train_mask_grid_sample.py:249-252).  Parameters, first and second moments live in three flat fp32 buffers with
// the same offsets; the gradients stay where autograd left them (one pointer per tensor, handed over in the kernel arguments), so a
// step is ONE launch instead of a multi-tensor launch per 30-odd tensors plus the step-counter adds.  The arithmetic follows
// torch/optim/adam.py `_single_tensor_adam` (lerp for the first moment, sqrt / bias_correction2_sqrt + eps, addcdiv by lr /
// bias_correction1) in fp32; a tensor whose gradient pointer is null is skipped like a parameter whose .grad is None.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct AdamGradPtrs { const float* g[ADAM_MAX_TENSORS]; };

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, const AdamHyper& h) {
  if (h.weight_decay != 0.0f) g = fmaf(h.weight_decay, p, g);
  m = m + (g - m) * (1.0f - h.beta1);                         // exp_avg.lerp_(grad, 1 - beta1)
  v = h.beta2 * v + (1.0f - h.beta2) * g * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / h.bias_correction2_sqrt + h.eps;
  p = p - h.step_size * (m / denom);                          // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// block b updates elements [blocks[b].y, blocks[b].y + blocks[b].z) of the flat buffers = elements blocks[b].w ... of tensor blocks[b].x
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                        const int4* __restrict__ blocks, AdamGradPtrs grads, AdamHyper h) {
  const int4 b = blocks[blockIdx.x];
  const float* g = grads.g[b.x];
  if (g == nullptr) return;
  g += b.w;
  const long base = b.y;
  const int n = b.z;
  if ((((uintptr_t)g) & 15) == 0 && (base & 3) == 0) {
    for (int i = 4 * threadIdx.x; i < n; i += 1024) {
      if (i + 4 <= n) {
        f32x4 pv = *(const f32x4*)(p + base + i), mv = *(const f32x4*)(m + base + i), vv = *(const f32x4*)(v + base + i);
        const f32x4 gv = *(const f32x4*)(g + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = pv[e], me = mv[e], ve = vv[e];
          adam_one(pe, me, ve, gv[e], h);
          pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        *(f32x4*)(p + base + i) = pv; *(f32x4*)(m + base + i) = mv; *(f32x4*)(v + base + i) = vv;
      } else {
        for (int e = i; e < n; ++e) adam_one(p[base + e], m[base + e], v[base + e], g[e], h);
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) adam_one(p[base + i], m[base + i], v[base + i], g[i], h);
  }
}

int launch_adam_step(float* p, float* m, float* v, const int* blocks, int n_blocks, const float* const* grads, int n_tensors,
                     const AdamHyper& h, hipStream_t stream) {
  if (n_blocks <= 0) return 0;
  AdamGradPtrs gp;
  for (int t = 0; t < ADAM_MAX_TENSORS; ++t) gp.g[t] = t < n_tensors ? grads[t] : nullptr;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, p, m, v, (const int4*)blocks, gp, h);
  return check_launch("adam_step_kernel");
}


}  // namespace crnerf
