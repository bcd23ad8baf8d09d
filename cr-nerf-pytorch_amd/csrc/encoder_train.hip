// Training twins of the appearance encoder: forward that keeps every layer output, and the backward
// pass -- weight / bias gradients of the seven convolutions and the gradient w.r.t. the input image (the encoder is also
// applied to the re-rendered image, train_mask_grid_sample.py:219, so d_img flows on into the decoder).
// Reference: encoder_sameoutputsize, models/linearStyleTransfer.py:208-276, differentiated by PyTorch autograd there.
// Same data layout as encoder.hip: activations pixel-major (HWC), one wave = one pixel x 64 channels.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace crnerf {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }   // ReflectionPad2d(1)
__device__ __forceinline__ float lrelu_grad(float y) { return y > 0.0f ? 1.0f : 0.2f; }                       // y = lrelu(pre): same sign

// forward kernels are encoder.hip's; redeclared here through small wrappers in that file
void enc_conv(int taps, bool act, const float* in, const float* wt, const float* b, float* out, int H, int W, int cin, int cout, hipStream_t st);
void enc_transpose_weights(const float* w, float* wt, int cout, int cin, int taps, hipStream_t st);
void enc_transpose_weights_all(const float* const* w, float* const* wt, const int* cout, const int* cin, const int* taps, hipStream_t st);
void enc_chw_to_hwc(const float* in, float* out, int C, int HW, hipStream_t st);
void enc_conv_gemm_pooled(const float* in, float* xcol, const float* w, const float* b, float* out, int Hs, int Ws, int cin, int cout, hipStream_t st);
void enc_chw_conv1(const float* img, const float* wt, const float* b, float* a0, float* y1, int HW, hipStream_t st);
void enc_maxpool2(const float* in, float* out, int H, int W, int C, hipStream_t st);
void enc_adaptive_avgpool(const float* in, float* out, int H, int W, int C, int S, hipStream_t st, int Hg, int yoff, int o0, int o1);
void enc_im2col(const float* in, float* X, int H, int W, int cin, hipStream_t st);
void enc_gemm_nt(bool act, const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M, int N, int K, hipStream_t st);
void enc_conv_gemm(int taps, const float* in, float* xcol, const float* w, const float* b, float* out, int H, int W, int cin, int cout, hipStream_t st);

static const int TCIN[7] = {3, 3, 64, 64, 128, 128, 128}, TCOUT[7] = {3, 64, 64, 128, 128, 128, 64}, TTAPS[7] = {1, 9, 9, 9, 9, 9, 1};

// A ROW BAND of an image (round 6: the encoder passes over the re-rendered images in ray-parallel training, one band of rows per rank -- DESIGN 4).
// The seven layers run on rows [row0, row0 + H) of an image of Hg rows as on an image of H rows (the band brings a halo wide enough that what the
// reflection padding at its cut edges gets wrong never reaches the rows it owns); only the FINAL pooling knows the whole: AdaptiveAvgPool2d(32)'s
// windows are the whole image's, and only the output rows [o0, o1) of the 32 x 32 map are produced -- n_out = (o1 - o0) * 32 pixels instead of
// 1,024.  The whole image is the band {Hg = H, row0 = 0, [0, 32)}.
struct EncBand { int Hg, row0, o0, o1; };
static int band_check(int H, const EncBand& b) {
  if (b.row0 < 0 || (b.row0 & 3) || b.row0 + H > b.Hg || b.o0 < 0 || b.o1 > 32 || b.o0 >= b.o1)
    return set_error(-3, "encoder band: rows must lie inside the image, start on a multiple of 4, and 0 <= o0 < o1 <= 32");
  const int H4g = (b.Hg / 2) / 2, y0 = (b.o0 * H4g) / 32, y1 = (b.o1 * H4g + 31) / 32;      // quarter-resolution rows the owned outputs read
  if (y0 < b.row0 / 4 || y1 > b.row0 / 4 + (H / 2) / 2) return set_error(-3, "encoder band: the output rows' pooling windows leave the band");
  return 0;
}

struct EncLayout {   // float offsets of the saved activations; x3..x6: the patch matrices of conv3..conv6 (kept for the weight gradients)
  size_t a0, y1, y2, y3, p3, y4, y5, p5, y6, p6, x3, x4, x5, x6, end;
  int H, W, H2, W2, H4, W4, n_out;
};
static EncLayout enc_layout(int H, int W, int n_out = 1024) {
  EncLayout L;
  L.H = H; L.W = W; L.H2 = H / 2; L.W2 = W / 2; L.H4 = L.H2 / 2; L.W4 = L.W2 / 2; L.n_out = n_out;
  const size_t n0 = (size_t)H * W, n2 = (size_t)L.H2 * L.W2, n4 = (size_t)L.H4 * L.W4;
  size_t o = 0;                                       // (p3 / p5: the pooled maps are not materialised any more -- they live in x4 / x6 only)
  L.a0 = o; o += n0 * 3;   L.y1 = o; o += n0 * 3;   L.y2 = o; o += n0 * 64;  L.y3 = o; o += n0 * 64;
  L.p3 = o;                L.y4 = o; o += n2 * 128; L.y5 = o; o += n2 * 128;
  L.p5 = o;                L.y6 = o; o += n4 * 128; L.p6 = o; o += (size_t)n_out * 128;
  L.x3 = o; o += n0 * 576; L.x4 = o; o += n2 * 576; L.x5 = o; o += n2 * 1152; L.x6 = o; o += n4 * 1152;
  L.end = o;
  return L;
}
static size_t enc_weight_floats() {
  size_t n = 0;
  for (int l = 0; l < 7; ++l) n += (size_t)TCIN[l] * TCOUT[l] * TTAPS[l];
  return n;
}
size_t encoder_train_saved_bytes(int H, int W, int n_out) { return (enc_layout(H, W, n_out).end + enc_weight_floats()) * sizeof(float); }
struct ScratchLayout { size_t ga, gb, g[7], X, wd, ws, ws_floats, end; };
static void enc_wgrad_specs(int H, int W, WgradSpec* sp, int n_out) {   // shapes only (pointers null): the seven weight-gradient jobs of one backward
  const long n0 = (long)H * W, n2 = (long)(H / 2) * (W / 2), n4 = (long)(H / 4) * (W / 4);
  const long P[7] = {n0, n0, n0, n2, n2, n4, n_out};
  for (int l = 0; l < 7; ++l) {
    const int K = TCIN[l] * TTAPS[l];
    sp[l] = WgradSpec{nullptr, TCOUT[l], TCOUT[l], nullptr, K, K, nullptr, K, (float*)1, wgrad_job_weight(TCOUT[l], K), 0, P[l]};
  }
}
static ScratchLayout enc_scratch(int H, int W, int n_out = 1024) {
  const size_t n0 = (size_t)H * W, n2 = (size_t)(H / 2) * (W / 2), n4 = (size_t)(H / 4) * (W / 4);
  const size_t gmap = n0 * 64 > (size_t)n_out * 128 ? n0 * 64 : (size_t)n_out * 128;   // largest gradient map (>= n2*128, n4*128)
  size_t xcol = n0 * 64 * 9;                                                           // conv3's patch matrix is the largest
  if (n2 * 128 * 9 > xcol) xcol = n2 * 128 * 9;
  WgradSpec sp[7];
  enc_wgrad_specs(H, W, sp, n_out);
  const size_t np[7] = {n0, n0, n0, n2, n2, n4, (size_t)n_out};
  ScratchLayout L;
  size_t o = 0;
  L.ga = o; o += gmap; L.gb = o; o += gmap;
  for (int l = 0; l < 7; ++l) { L.g[l] = o; o += np[l] * TCOUT[l]; }                   // lrelu'-scaled upstream gradient of every layer: read by the
  L.X = o; o += xcol; L.wd = o; o += 128 * 128 * 9;                                    // batched weight-gradient launch at the end of the backward
  L.ws_floats = wgrad_batch_ws_floats(sp, 7);
  L.ws = o; o += L.ws_floats;
  L.end = o;
  return L;
}
size_t encoder_train_scratch_bytes(int H, int W, int n_out) { return enc_scratch(H, W, n_out).end * sizeof(float); }

int launch_encoder_forward_train(const float* img, int H, int W, const float* const* w, void* saved, float* out, hipStream_t st) {
  return launch_encoder_forward_train_band(img, H, W, H, 0, 0, 32, w, saved, out, st);
}

// img: rows [row0, row0 + H) of the image, [3, H, W] contiguous; out[(o1 - o0) * 32, 64]
int launch_encoder_forward_train_band(const float* img, int H, int W, int Hg, int row0, int o0, int o1, const float* const* w, void* saved, float* out, hipStream_t st) {
  if (H < 8 || W < 8) return set_error(-2, "encoder: image must be at least 8x8 (two 2x2 max-pools and reflection padding)");
  const EncBand band{Hg, row0, o0, o1};
  if (int rc = band_check(H, band)) return rc;
  const int n_out = (o1 - o0) * 32;
  const EncLayout L = enc_layout(H, W, n_out);
  float* s = (float*)saved;
  float* wt[7];
  float* p = s + L.end;
  for (int l = 0; l < 7; ++l) { wt[l] = p; p += TCIN[l] * TCOUT[l] * TTAPS[l]; }
  enc_transpose_weights_all(w, wt, TCOUT, TCIN, TTAPS, st);     // one launch (seven before: 28 of the 570 launches of a 1,024-ray train.sh step)
  enc_chw_conv1(img, wt[0], w[1], s + L.a0, s + L.y1, H * W, st);      // NCHW -> pixel-major (kept: conv1's weight gradient reads it) + conv1
  enc_conv(9, true, s + L.y1, wt[1], w[3], s + L.y2, H, W, 3, 64, st);
  // cin >= 64: fp32 MFMA GEMMs over the patch matrices (encoder.hip), which stay in `saved` for the weight gradients
  enc_conv_gemm(9, s + L.y2, s + L.x3, w[4], w[5], s + L.y3, H, W, 64, 64, st);
  enc_conv_gemm_pooled(s + L.y3, s + L.x4, w[6], w[7], s + L.y4, H, W, 64, 128, st);            // max-pool + conv4: the pooled map exists only inside x4
  enc_conv_gemm(9, s + L.y4, s + L.x5, w[8], w[9], s + L.y5, L.H2, L.W2, 128, 128, st);
  enc_conv_gemm_pooled(s + L.y5, s + L.x6, w[10], w[11], s + L.y6, L.H2, L.W2, 128, 128, st);   // max-pool + conv6
  enc_adaptive_avgpool(s + L.y6, s + L.p6, L.H4, L.W4, 128, 32, st, (Hg / 2) / 2, row0 / 4, o0, o1);
  enc_conv_gemm(1, s + L.p6, nullptr, w[12], w[13], out, o1 - o0, 32, 128, 64, st);
  return check_launch("encoder_forward_train");
}

// ---------------------------------------------------------------- backward kernels
// g[px][o] = d_out[px][o] * lrelu'(y[px][o])  (y = null: plain copy)
__global__ void enc_act_grad_kernel(const float* __restrict__ d_out, const float* __restrict__ y, float* __restrict__ g, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  g[idx] = y ? d_out[idx] * lrelu_grad(y[idx]) : d_out[idx];
}

// The patch matrix X[px][c * 9 + tap] = in[reflect(px + tap)][c] (enc_im2col, encoder.hip) makes the weight gradient the
// point-reduction GEMM dW[o][c*9+tap] = sum_px g[px][o] X[px][c*9+tap] -- exactly the shape (and the output layout,
// [cout][cin][3][3]) of the MLP's MFMA wgrad kernel (mlp_train16.hip), which is reused as is.

// d_in[px][c] = sum over (patch row q, tap) with reflect(q + tap) == px of dX[q][c*9 + tap]: the adjoint of the reflection-padded
// gather, applied to the patch-matrix gradient dX = g W that the GEMM path produces (pixels in row / column 1 and n-2 also
// collect what the padding mirrored)
// yact != null: what is written is g = d_in * lrelu'(yact) -- the upstream gradient of the layer BELOW with its activation's derivative already in it
// (round 4: the separate enc_act_grad pass per layer is gone; `d_in` is then that layer's g buffer)
__global__ void enc_col2im_kernel(const float* __restrict__ dX, float* __restrict__ d_in, int H, int W, int cin, const float* __restrict__ yact) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * W * cin) return;
  const int c = (int)(idx % cin);
  const int px = (int)(idx / cin), py = px / W, pxx = px % W;
  int ty[3], tx[3], nty = 0, ntx = 0;                    // padded coordinates that map onto (py, pxx)
  ty[nty++] = py; if (py == 1) ty[nty++] = -1; if (py == H - 2) ty[nty++] = H;
  tx[ntx++] = pxx; if (pxx == 1) tx[ntx++] = -1; if (pxx == W - 2) tx[ntx++] = W;
  float acc = 0.0f;
  for (int a = 0; a < nty; ++a)
    for (int ky = 0; ky < 3; ++ky) {
      const int qy = ty[a] - ky + 1;
      if (qy < 0 || qy >= H) continue;
      for (int b = 0; b < ntx; ++b)
        for (int kx = 0; kx < 3; ++kx) {
          const int qx = tx[b] - kx + 1;
          if (qx < 0 || qx >= W) continue;
          acc += dX[((long)qy * W + qx) * cin * 9 + c * 9 + ky * 3 + kx];
        }
    }
  d_in[idx] = yact ? acc * lrelu_grad(yact[idx]) : acc;
}

// d_in[px][c] = sum over (output pixel q, tap) with reflect(q + tap) == px of sum_o g[q][o] * w[o][c][tap]   (g already carries lrelu')
// (the adjoint of the reflection-padded gather: pixels in row / column 1 and n-2 also collect what the padding mirrored)
// w: the layer's own [cout][cin][taps] weights, indexed in place (round 4: the [tap][o][c] re-layout launch in front of this kernel is gone -- it
// only ever serves the two 3-channel layers); yact as in enc_col2im_kernel
template <int TAPS>
__global__ __launch_bounds__(256) void enc_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                        float* __restrict__ d_in, int H, int W, int cin, int cout, const float* __restrict__ yact) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int px = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: g[q][o] becomes a scalar load
  if (px >= H * W) return;
  const int py = px / W, pxx = px % W;
  float acc = 0.0f;
  if (TAPS == 9) {
    int ty[3], tx[3], nty = 0, ntx = 0;                    // padded coordinates that map onto (py, pxx)
    ty[nty++] = py; if (py == 1) ty[nty++] = -1; if (py == H - 2) ty[nty++] = H;
    tx[ntx++] = pxx; if (pxx == 1) tx[ntx++] = -1; if (pxx == W - 2) tx[ntx++] = W;
    for (int a = 0; a < nty; ++a)
      for (int ky = 0; ky < 3; ++ky) {
        const int qy = ty[a] - ky + 1;
        if (qy < 0 || qy >= H) continue;
        for (int b = 0; b < ntx; ++b)
          for (int kx = 0; kx < 3; ++kx) {
            const int qx = tx[b] - kx + 1;
            if (qx < 0 || qx >= W) continue;
            const long q = (long)qy * W + qx;
            const float* gq = g + q * cout;
            const int cc = c < cin ? c : cin - 1;
            const float* wp = w + (long)cc * 9 + ky * 3 + kx;
#pragma unroll 8
            for (int o = 0; o < cout; ++o) acc = fmaf(gq[o], wp[(long)o * cin * 9], acc);
          }
      }
  } else {
    const float* gq = g + (long)px * cout;
    const int cc = c < cin ? c : cin - 1;
#pragma unroll 8
    for (int o = 0; o < cout; ++o) acc = fmaf(gq[o], w[(long)o * cin + cc], acc);
  }
  if (c < cin) d_in[(long)px * cin + c] = yact ? acc * lrelu_grad(yact[(long)px * cin + c]) : acc;
}

// conv2's data gradient (3 x 3, THREE input channels, cout <= 64): one THREAD per pixel computing all three channels.  enc_dgrad_kernel<9> maps the
// input channels to lanes -- with cin = 3 sixty-one lanes of every wave idle and the three that work walk a 576-step chain of scalar loads: 0.34 ms
// per call on a 256 x 256 image (1.36 ms of a 65,536-ray step for 0.2 GFLOP).  The same products in the same order per (pixel, channel) as that
// kernel: the same bits.  Weights through LDS (w[o][c][tap], 6.9 KB; the lanes of a wave read one address at a time except on the image border).
__global__ __launch_bounds__(64) void enc_dgrad9_c3_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ d_in, int H, int W,
                                                            int cout, const float* __restrict__ yact) {
  __shared__ float ws[64 * 27];
  for (int k = threadIdx.x; k < cout * 27; k += 64) ws[k] = w[k];
  __syncthreads();
  const int px = blockIdx.x * 64 + threadIdx.x;        // one wave per workgroup: a 32 x 32 map still spreads over 16 CUs
  if (px >= H * W) return;
  const int py = px / W, pxx = px % W;
  int ty[3], tx[3], nty = 0, ntx = 0;                    // padded coordinates that map onto (py, pxx)
  ty[nty++] = py; if (py == 1) ty[nty++] = -1; if (py == H - 2) ty[nty++] = H;
  tx[ntx++] = pxx; if (pxx == 1) tx[ntx++] = -1; if (pxx == W - 2) tx[ntx++] = W;
  float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
  for (int a = 0; a < nty; ++a)
    for (int ky = 0; ky < 3; ++ky) {
      const int qy = ty[a] - ky + 1;
      if (qy < 0 || qy >= H) continue;
      for (int b = 0; b < ntx; ++b)
        for (int kx = 0; kx < 3; ++kx) {
          const int qx = tx[b] - kx + 1;
          if (qx < 0 || qx >= W) continue;
          const float4* gq = (const float4*)(g + ((long)qy * W + qx) * cout);
          const float* wp = ws + ky * 3 + kx;
          for (int o4 = 0; o4 < cout / 4; ++o4) {
            const float4 gv = gq[o4];
            const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float* wo = wp + (4 * o4 + e) * 27;
              acc0 = fmaf(ge[e], wo[0], acc0);
              acc1 = fmaf(ge[e], wo[9], acc1);
              acc2 = fmaf(ge[e], wo[18], acc2);
            }
          }
        }
    }
  float* dst = d_in + (long)px * 3;
  if (yact) {
    const float* ya = yact + (long)px * 3;
    acc0 *= lrelu_grad(ya[0]); acc1 *= lrelu_grad(ya[1]); acc2 *= lrelu_grad(ya[2]);
  }
  dst[0] = acc0; dst[1] = acc1; dst[2] = acc2;
}

// MaxPool2d(2,2) backward: the gradient goes to the first maximum of the window in scan order (ATen's tie rule)
// act: `in` is the output of a LeakyReLU layer and d_in receives g = d(in) * lrelu'(in) (the pooled position's value is bv itself)
__global__ void enc_maxpool2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ d_out, float* __restrict__ d_in, int H, int W, int C, int act) {
  const int Ho = H / 2, Wo = W / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Ho * Wo * C) return;
  const int c = idx % C, px = idx / C, yy = px / Wo, xx = px % Wo;
  const long base = ((long)(2 * yy) * W + 2 * xx) * C + c;
  const long off[4] = {0, C, (long)W * C, (long)W * C + C};
  int best = 0;
  float bv = in[base];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float v = in[base + off[k]];
    if (v > bv) { bv = v; best = k; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) d_in[base + off[k]] = k == best ? (act ? d_out[idx] * lrelu_grad(bv) : d_out[idx]) : 0.0f;
}

// AdaptiveAvgPool2d(S) backward: every input position collects d_out / window_area from the windows that contain it
// band (EncBand): d_in covers rows [yoff, yoff + H) of a map of Hg rows, d_out the output rows [o0, o1) only
__global__ void enc_avgpool_bwd_kernel(const float* __restrict__ d_out, float* __restrict__ d_in, int H, int W, int C, int S, const float* __restrict__ yact,
                                       int Hg, int yoff, int o0, int o1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W * C) return;
  const int c = idx % C, px = idx / C, yy = px / W + yoff, xx = px % W;
  float acc = 0.0f;
  for (int oy = o0; oy < o1; ++oy) {
    const int y0 = (oy * Hg) / S, y1 = ((oy + 1) * Hg + S - 1) / S;
    if (yy < y0 || yy >= y1) continue;
    for (int ox = 0; ox < S; ++ox) {
      const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
      if (xx < x0 || xx >= x1) continue;
      acc += d_out[((long)(oy - o0) * S + ox) * C + c] / (float)((y1 - y0) * (x1 - x0));
    }
  }
  d_in[idx] = yact ? acc * lrelu_grad(yact[idx]) : acc;
}

struct BwdBufs { float* X; float* wd; WgradSpec* specs; int* nspec; };

// xsaved: the layer's patch matrix kept by the forward (GEMM layers) or null (built here); wt: the forward's [cin*taps][cout]
// re-layout of the weights (kept in `saved`): for cin >= 64 the data gradient is the GEMM dX[px][k] = sum_o g[px][o] wt[k][o] on the
// fp32 matrix cores + the gather over the padding's adjoint, instead of a 1,152-step dependent chain per pixel
// g: this layer's upstream gradient with its own lrelu' already applied (made by the producer above it; only the top layer runs enc_act_grad).
// d_in / yact: where the gradient w.r.t. this layer's input goes, and -- when that input is a LeakyReLU layer's output consumed directly --
// that output, so that what is written is already the NEXT layer's g.
template <int TAPS>
static void conv_bwd(const float* g, const float* in, const float* w, const BwdBufs& B, float* dW, float* db, float* d_in, const float* yact,
                     int H, int W, int cin, int cout, hipStream_t st, const float* xsaved = nullptr, const float* wt = nullptr) {
  const float* X = in;
  if (TAPS == 9) {
    if (xsaved) X = xsaved;
    else { enc_im2col(in, B.X, H, W, cin, st); X = B.X; }
  }
  // dW[o][k] = sum_px g[px][o] X[px][k] (+ bias sums): the MFMA point-reduction GEMM of mlp_train16.hip, queued -- all seven
  // layers run in ONE batched launch at the end of the backward (g and X of every layer stay untouched until then)
  B.specs[(*B.nspec)++] = WgradSpec{g, cout, cout, X, cin * TAPS, cin * TAPS, dW, cin * TAPS, db, wgrad_job_weight(cout, cin * TAPS), 0, (long)H * W};
  if (!d_in) return;
  if (wt && cin >= 64 && (cout & 7) == 0) {
    if (TAPS == 9) {
      enc_gemm_nt(false, g, cout, wt, cout, nullptr, B.X, cin * 9, H * W, cin * 9, cout, st);   // B.X is free: the patch matrix came from `saved`
      const long nd = (long)H * W * cin;
      hipLaunchKernelGGL(enc_col2im_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, B.X, d_in, H, W, cin, yact);
    } else {
      enc_gemm_nt(false, g, cout, wt, cout, nullptr, d_in, cin, H * W, cin, cout, st);          // (1 x 1: its consumer applies the derivative)
    }
    return;
  }
  if (TAPS == 9 && cin == 3 && cout <= 64 && (cout & 3) == 0 && ((uintptr_t)g & 15) == 0) {   // conv2: a thread per pixel (the lane-per-channel kernel leaves 61 of 64 lanes idle)
    hipLaunchKernelGGL(enc_dgrad9_c3_kernel, dim3((H * W + 63) / 64), dim3(64), 0, st, g, w, d_in, H, W, cout, yact);
    return;
  }
  hipLaunchKernelGGL((enc_dgrad_kernel<TAPS>), dim3((H * W + 3) / 4, (cin + 63) / 64), dim3(256), 0, st, g, w, d_in, H, W, cin, cout, yact);
}

// conv1's data gradient (1x1, 3 -> 3; enc_dgrad_kernel<1>'s products in its order) written straight into the photo's NCHW layout:
// d_img[c][px] = sum_o g[px][o] w[o][c] -- one launch instead of the pixel-major data gradient + its re-layout
__global__ void conv1_dgrad_chw_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ d_img, int HW) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= HW) return;
  const float g0 = g[3 * px], g1 = g[3 * px + 1], g2 = g[3 * px + 2];
#pragma unroll
  for (int c = 0; c < 3; ++c) d_img[(long)c * HW + px] = fmaf(g2, w[6 + c], fmaf(g1, w[3 + c], fmaf(g0, w[c], 0.0f)));
}

// saved: from launch_encoder_forward_train; out: its output (for lrelu7'); d_out[1024,64]; grads[14] in weight order;
// d_img[3,H,W] (NCHW) or null
int launch_encoder_backward(int H, int W, const float* const* w, const void* saved, const float* out, const float* d_out, void* scratch,
                            float* const* grads, float* d_img, hipStream_t st) {
  return launch_encoder_backward_band(H, W, H, 0, 0, 32, w, saved, out, d_out, scratch, grads, d_img, st);
}

// out / d_out: [(o1 - o0) * 32, 64]; d_img [3, H, W]: the gradient w.r.t. the band's rows (what its owned outputs contribute; the caller sums the bands)
int launch_encoder_backward_band(int H, int W, int Hg, int row0, int o0, int o1, const float* const* w, const void* saved, const float* out, const float* d_out,
                                 void* scratch, float* const* grads, float* d_img, hipStream_t st) {
  const EncBand band{Hg, row0, o0, o1};
  if (int rc = band_check(H, band)) return rc;
  const int n_out = (o1 - o0) * 32;
  const EncLayout L = enc_layout(H, W, n_out);
  const float* s = (const float*)saved;
  const ScratchLayout SL = enc_scratch(H, W, n_out);
  float* base = (float*)scratch;
  float* ga = base + SL.ga;
  float* gb = base + SL.gb;
  WgradSpec specs[7];
  int nspec = 0;
  const BwdBufs B{base + SL.X, base + SL.wd, specs, &nspec};
  const int n0 = H * W, n2 = L.H2 * L.W2, n4 = L.H4 * L.W4;
  const float* wt[7];                                    // the forward's transposed weights, behind the activations in `saved`
  {
    const float* p = s + L.end;
    for (int l = 0; l < 7; ++l) { wt[l] = p; p += TCIN[l] * TCOUT[l] * TTAPS[l]; }
  }
  float* g[7];
  for (int l = 0; l < 7; ++l) g[l] = base + SL.g[l];
  // the top layer's g from the caller's gradient; every other g[l] is written by the kernel that produces that layer's upstream gradient,
  // derivative included (seven elementwise passes and two memsets per backward before round 4)
  hipLaunchKernelGGL(enc_act_grad_kernel, dim3((unsigned)(((long)n_out * 64 + 255) / 256)), dim3(256), 0, st, d_out, out, g[6], (long)n_out * 64);
  conv_bwd<1>(g[6], s + L.p6, w[12], B, grads[12], grads[13], ga, nullptr, o1 - o0, 32, 128, 64, st, nullptr, wt[6]);             // conv7 -> d p6
  hipLaunchKernelGGL(enc_avgpool_bwd_kernel, dim3((n4 * 128 + 255) / 256), dim3(256), 0, st, ga, g[5], L.H4, L.W4, 128, 32, s + L.y6,
                     (Hg / 2) / 2, row0 / 4, o0, o1);                                                                               // -> g of conv6
  conv_bwd<9>(g[5], nullptr, w[10], B, grads[10], grads[11], ga, nullptr, L.H4, L.W4, 128, 128, st, s + L.x6, wt[5]);             // conv6 -> d p5
  if ((L.H2 | L.W2) & 1) (void)hipMemsetAsync(g[4], 0, (size_t)n2 * 128 * sizeof(float), st);   // (an odd row / column lies in no pooling window)
  hipLaunchKernelGGL(enc_maxpool2_bwd_kernel, dim3((n4 * 128 + 255) / 256), dim3(256), 0, st, s + L.y5, ga, g[4], L.H2, L.W2, 128, 1);   // -> g of conv5
  conv_bwd<9>(g[4], s + L.y4, w[8], B, grads[8], grads[9], g[3], s + L.y4, L.H2, L.W2, 128, 128, st, s + L.x5, wt[4]);             // conv5 -> g of conv4
  conv_bwd<9>(g[3], nullptr, w[6], B, grads[6], grads[7], gb, nullptr, L.H2, L.W2, 64, 128, st, s + L.x4, wt[3]);                 // conv4 -> d p3
  if ((H | W) & 1) (void)hipMemsetAsync(g[2], 0, (size_t)n0 * 64 * sizeof(float), st);
  hipLaunchKernelGGL(enc_maxpool2_bwd_kernel, dim3((n2 * 64 + 255) / 256), dim3(256), 0, st, s + L.y3, gb, g[2], H, W, 64, 1);           // -> g of conv3
  conv_bwd<9>(g[2], s + L.y2, w[4], B, grads[4], grads[5], g[1], s + L.y2, H, W, 64, 64, st, s + L.x3, wt[2]);                     // conv3 -> g of conv2
  conv_bwd<9>(g[1], s + L.y1, w[2], B, grads[2], grads[3], g[0], nullptr, H, W, 3, 64, st);                                         // conv2 -> g of conv1 (no activation)
  conv_bwd<1>(g[0], s + L.a0, w[0], B, grads[0], grads[1], nullptr, nullptr, H, W, 3, 3, st);                                        // conv1: weight / bias gradient
  if (d_img) hipLaunchKernelGGL(conv1_dgrad_chw_kernel, dim3((n0 + 255) / 256), dim3(256), 0, st, g[0], w[0], d_img, n0);            // its data gradient, NCHW
  if (int rc = wgrad_batch(specs, nspec, base + SL.ws, SL.ws_floats, st)) return rc;        // the seven weight / bias gradients: two launches
  return check_launch("encoder_backward");
}

}  // namespace crnerf
