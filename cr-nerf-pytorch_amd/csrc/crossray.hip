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
// Every kernel takes up to two "jobs" (content grid, style grid) so the single-GPU entry
// crnerf_crossray_decode_f32 is 6 launches from one host call (4 for grids of <= 4096 pixels: the row
// reduction rides in the fc launch, the fold in the apply launch).
//
// gram is the only piece with real arithmetic (18.4 k MAC/pixel): it runs on the fp32 MFMA with the
// same swapped-operand trick as the NeRF MLP (mlp_core.h): 32 pixels per wavefront, the conv chain's
// activations stay in registers, the 72 KiB of weights sit in LDS as A-operand fragments, and the
// 32x32 Gram update itself is 16 more MFMAs per tile (H H^T through a 4 KiB LDS transpose).  Grids of
// <= 4096 pixels (the headline 32x32 grid) take the cooperative variant: one tile per WORKGROUP, its
// four waves splitting every layer (gram_tiles_coop).
// (Tried and dropped in round 2: the whole small-grid decode as ONE cooperative launch of 64 co-resident
// workgroups with device-scope barriers in place of the launch boundaries -- 35.9 us against 37.2 us for
// the six launches, but every barrier's acquire invalidates the L2, and the render kernel of the next step
// then re-fetches its 5 MB of weights: +16 us there, a net loss.)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"
#include "crossray.h"

namespace crnerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------- channel sums (two jobs)
struct SumJob { const float* x; long HW; float* partial; int nblk; };

constexpr int SUM_THREADS = 1024;   // 16 channel quads x 64 pixel slots
__global__ __launch_bounds__(SUM_THREADS) void chansum_partial_kernel(SumJob j0, SumJob j1) {
  // thread = (pixel slot, channel quad); 16 channel quads x 64 pixel slots per block, one block per CU
  __shared__ f32x4 red[SUM_THREADS];
  const bool second = (int)blockIdx.x >= j0.nblk;
  const SumJob j = second ? j1 : j0;
  const int blk = second ? blockIdx.x - j0.nblk : blockIdx.x;
  const int cq = threadIdx.x & 15, ps = threadIdx.x >> 4;
  // eight independent row streams per thread x 16 waves per CU: 128 B in flight per lane, ~32 MB on the chip (HBM needs ~16 MB
  // by Little's law; one dependent add chain on 4 waves/CU reached 12 % of HBM, four streams 37 %).  The partial-row count
  // stays <= CROSSRAY_SUM_BLOCKS = 256 because the Gram kernel's prologue re-reads every partial row.
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  // the block owns a contiguous pixel range (whole 64-pixel rounds); the eight streams are rounds k, k+1, ..., k+7 -- the
  // tail is handled by clamping the row and zeroing its weight, never by a serial one-stream loop
  const long rounds = (j.HW + 63) / 64, per = (rounds + j.nblk - 1) / j.nblk;
  const long r0 = (long)blk * per, r1 = r0 + per < rounds ? r0 + per : rounds;
  const float* base = j.x + cq * 4;
  const long last = j.HW - 1;
  for (long rd = r0; rd < r1; rd += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long px = (rd + u) * 64 + ps;
      const bool ok = rd + u < r1 && px <= last;
      v[u] = *(const f32x4*)(base + (ok ? px : last) * 64);
      if (!ok) v[u] = f32x4{0, 0, 0, 0};
    }
    a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3]; a4 += v[4]; a5 += v[5]; a6 += v[6]; a7 += v[7];
  }
  const f32x4 acc = ((a0 + a4) + (a1 + a5)) + ((a2 + a6) + (a3 + a7));
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 32; s >= 1; s >>= 1) {
    if (ps < s) red[threadIdx.x] += red[threadIdx.x + s * 16];
    __syncthreads();
  }
  if (ps == 0) *(f32x4*)(j.partial + (long)blk * 64 + cq * 4) = red[cq];
}

struct RedJob { const float* partial; int rows; float* out; };
// out[c] = sum_r partial[r][c], cols columns per job, fixed summation tree (deterministic).  One 1024-thread block per 64
// columns: 16 row slices x 64 columns, four independent accumulators per thread, slices combined through LDS in a fixed order.
// (The first version walked the rows serially from one thread per column: 256 dependent L2 round trips = 60 us per call at an
// 800x800 grid, more than the 25 us channel-sum pass it finishes.)
constexpr int RED_THREADS = 1024;
__global__ __launch_bounds__(RED_THREADS) void reduce_rows_kernel(RedJob j0, RedJob j1, int cols) {
  __shared__ float red[RED_THREADS];
  const RedJob j = blockIdx.y ? j1 : j0;
  if (!j.out) return;
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (c < cols) {
    int r = sl;
    for (; r + 48 < j.rows; r += 64) {
      a0 += j.partial[(long)r * cols + c];
      a1 += j.partial[(long)(r + 16) * cols + c];
      a2 += j.partial[(long)(r + 32) * cols + c];
      a3 += j.partial[(long)(r + 48) * cols + c];
    }
    for (; r < j.rows; r += 16) a0 += j.partial[(long)r * cols + c];
  }
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int s = 8; s >= 1; s >>= 1) {
    if (sl < s) red[threadIdx.x] += red[threadIdx.x + s * 64];
    __syncthreads();
  }
  if (sl == 0 && c < cols) j.out[c] = red[cl];
}

static int chansum_blocks(long HW) {
  const long b = (HW + 63) / 64;
  return (int)(b < CROSSRAY_SUM_BLOCKS ? (b < 1 ? 1 : b) : CROSSRAY_SUM_BLOCKS);
}

int launch_crossray_chansum(const float* x, long HW, float* sum_out, float* workspace, hipStream_t stream) {
  if (HW <= 0) return set_error(-2, "crossray_chansum: empty grid");
  SumJob j{x, HW, workspace, chansum_blocks(HW)}, none{nullptr, 0, nullptr, 0};
  hipLaunchKernelGGL(chansum_partial_kernel, dim3(j.nblk), dim3(SUM_THREADS), 0, stream, j, none);
  RedJob r{workspace, j.nblk, sum_out}, rn{nullptr, 0, nullptr};
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1, 1), dim3(RED_THREADS), 0, stream, r, rn, 64);
  return check_launch("crossray_chansum");
}

// ---------------------------------------------------------------- Gram of the conv chain on the fp32 MFMA
// LDS (floats): A-operand fragments of the three 1x1 convs, biases, mean, per-wave transpose buffers
constexpr int GF_L1 = 0;                    // 32 fragments: 8 k-groups x 4 tiles   (64 -> 128)
constexpr int GF_L2 = GF_L1 + 32 * 256;     // 32 fragments: 16 k-groups x 2 tiles  (128 -> 64)
constexpr int GF_L3 = GF_L2 + 32 * 256;     //  8 fragments: 8 k-groups x 1 tile    (64 -> 32)
constexpr int GF_B1 = GF_L3 + 8 * 256;      // 128
constexpr int GF_B2 = GF_B1 + 128;          // 64
constexpr int GF_B3 = GF_B2 + 64;           // 32
constexpr int GF_MEAN = GF_B3 + 32;         // 64
constexpr int GF_HB = GF_MEAN + 64;         // 4 waves x [32 px][33]
constexpr int GF_FLOATS = GF_HB + 4 * 32 * 33;

struct GramJob {
  const float* x; long HW;
  const float* mean;          // [64] or null -> mean = (sum of chan_partial rows) * inv_count
  const float* chan_partial; int chan_rows; float inv_count;
  CnnTensors w;
  float* gram_partial;        // [nblk][1024]
  float* mean_out;            // [64] or null
  int nblk;
  int coop;                   // != 0: one tile per WORKGROUP (gram_tiles_coop; small grids), nblk <= tiles
};

__device__ __forceinline__ float lrelu02(float v) { return v > 0.0f ? v : 0.2f * v; }

template <int NT>
__device__ __forceinline__ void bias_init(f32x16 (&acc)[NT], const float* bias, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const f32x4*)(bias + 32 * t + 8 * q + 4 * h);
      acc[t][4 * q + 0] = b[0]; acc[t][4 * q + 1] = b[1]; acc[t][4 * q + 2] = b[2]; acc[t][4 * q + 3] = b[3];
    }
}

// acc[t] += W-fragments (LDS, fragment (v*NT + t)) x src, NG k-groups
template <int NT, int NG, int NSRC>
__device__ __forceinline__ void conv_mfma(const float* frags, int lane, const f32x16 (&src)[NSRC], f32x16 (&acc)[NT]) {
#pragma unroll
  for (int v = 0; v < NG; ++v)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 a = *(const f32x4*)(frags + (v * NT + t) * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = MFMA32(a[j], src[v >> 2][(v & 3) * 4 + j], acc[t]);
    }
}

// the three 1x1-conv matrices as MFMA A-operand fragments + biases into LDS.  frag(v,t)[lane = 32kk + i][j] = W[32t + i][8v + 4kk + j];
// the four j are contiguous in W, so one 16-byte load per lane-slot, all 18 of a thread in flight at once (one element per loop
// iteration was 36 us of serialized global-load latency -- as long as the render kernel of a 32x32 bf16 batch)
__device__ __forceinline__ void gram_stage_weights(const CnnTensors& w, float* sm, int tid) {
  static_assert(GF_B1 == 72 * 256, "fragment area");
  f32x4 stage[18];
#pragma unroll
  for (int it = 0; it < 18; ++it) {
    const int idx4 = tid + 256 * it;
    const int frag = idx4 >> 6, l = idx4 & 63, i = l & 31, kk = l >> 5;
    const float* src;
    if (frag < 32) { const int v = frag >> 2, t = frag & 3; src = w.w1 + (32 * t + i) * 64 + 8 * v + 4 * kk; }
    else if (frag < 64) { const int f = frag - 32, v = f >> 1, t = f & 1; src = w.w2 + (32 * t + i) * 128 + 8 * v + 4 * kk; }
    else { const int v = frag - 64; src = w.w3 + i * 64 + 8 * v + 4 * kk; }
    stage[it] = *(const f32x4*)src;
  }
#pragma unroll
  for (int it = 0; it < 18; ++it) *(f32x4*)(sm + 4 * (tid + 256 * it)) = stage[it];
  if (tid < 128) sm[GF_B1 + tid] = w.b1[tid];
  if (tid < 64) sm[GF_B2 + tid] = w.b2[tid];
  if (tid < 32) sm[GF_B3 + tid] = w.b3[tid];
}

// sm[GF_MEAN .. +64) = (sum of the `rows` partial rows) * inv_count: four row-interleaved partial sums per channel with the loads of
// a partial in flight together (a plain row loop is `rows` dependent L2 round trips -- 64 of them, ~25 us, for a 32x32 grid).
// Two __syncthreads inside; uses the (idle) transpose buffers.
__device__ __forceinline__ void gram_mean_from_partials(const float* chan_partial, int rows, float inv_count, float* mean_out, float* sm, int tid) {
  float part = 0.0f;
  const int c = tid & 63, q = tid >> 6;
#pragma unroll 8
  for (int r = q; r < rows; r += 4) part += chan_partial[r * 64 + c];
  sm[GF_HB + q * 64 + c] = part;
  __syncthreads();
  if (tid < 64) {
    const float m = ((sm[GF_HB + tid] + sm[GF_HB + 64 + tid]) + (sm[GF_HB + 128 + tid] + sm[GF_HB + 192 + tid])) * inv_count;
    sm[GF_MEAN + tid] = m;
    if (mean_out) mean_out[tid] = m;
  }
  __syncthreads();
}

// tiles blk*4 + wave, ... of the job through the conv chain; the workgroup's Gram partial -> J.gram_partial[blk]
__device__ __forceinline__ void gram_tiles(const GramJob& J, int blk, float* sm, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  float* hb = sm + GF_HB + wave * 32 * 33;
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) G[r] = 0.0f;

  const long tiles = (J.HW + 31) / 32;
  for (long tile = (long)blk * 4 + wave; tile < tiles; tile += (long)J.nblk * 4) {
    const long px = tile * 32 + p;
    const bool valid = px < J.HW;
    const float* row = J.x + (valid ? px : 0) * 64;
    f32x16 xin[2];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const f32x4 xv = *(const f32x4*)(row + 8 * v + 4 * h);
      const f32x4 mv = *(const f32x4*)(sm + GF_MEAN + 8 * v + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) xin[v >> 2][(v & 3) * 4 + j] = xv[j] - mv[j];      // cF - cMean, :59-65
    }
    f32x16 a1[4], a2[2], a3[1];
    bias_init<4>(a1, sm + GF_B1, h);
    conv_mfma<4, 8>(sm + GF_L1, lane, xin, a1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[t][r] = lrelu02(a1[t][r]);
    bias_init<2>(a2, sm + GF_B2, h);
    conv_mfma<2, 16>(sm + GF_L2, lane, a1, a2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[t][r] = lrelu02(a2[t][r]);
    bias_init<1>(a3, sm + GF_B3, h);
    conv_mfma<1, 8>(sm + GF_L3, lane, a2, a3);
    // transpose through LDS: hb[px][feat], feat = 8q + 4h + j; padded pixels contribute nothing
#pragma unroll
    for (int r = 0; r < 16; ++r) hb[p * 33 + 8 * (r >> 2) + 4 * h + (r & 3)] = valid ? a3[0][r] : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // G += H H^T : A[i][k] = H[feat i][px k], B[k][j] = H[feat j][px k] -> the same register
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = hb[(2 * s + h) * 33 + p];
      G = MFMA32(a, a, G);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // cross-wave reduction (fragment area is dead now); D layout: col j = p, row i = (r&3) + 8(r>>2) + 4h
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) sm[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + p] = G[r];
  __syncthreads();
  const f32x4 s = *(f32x4*)(sm + tid * 4) + *(f32x4*)(sm + 1024 + tid * 4) + *(f32x4*)(sm + 2048 + tid * 4) + *(f32x4*)(sm + 3072 + tid * 4);
  *(f32x4*)(J.gram_partial + (long)blk * 1024 + tid * 4) = s;
}

// ---- small grids: the four waves of a workgroup share ONE 32-pixel tile.  With a tile per wave (gram_tiles) a 32 x 32 grid keeps
// 16 workgroups busy for 11 us each -- the time of one wave's serial chain of 288 MFMAs -- and leaves the other 240 CUs idle.
// Here wave w computes output tile w of conv1 (128 features), tile w&1 / K-half w>>1 of conv2, K-quarter w of conv3 and steps
// 4w..4w+3 of the Gram update: 76 MFMAs per wave, activations exchanged through LDS ([feature][pixel] rows, the B-operand shape).
constexpr int GC_A1 = GF_HB;                 // [128][33] conv1 output
constexpr int GC_A2 = GC_A1 + 128 * 33;      // [64][33]  conv2 output
constexpr int GC_H = GC_A2 + 64 * 33;        // [32 px][33] conv3 output, pixel-major (the Gram's operand shape)
constexpr int GC_P2 = GC_H + 32 * 33;        // conv2 partial sums of the second K half: [2 tiles][16 regs][64 lanes]
constexpr int GC_P3 = GC_P2 + 2048;          // conv3 partial sums of K quarters 1..3
constexpr int GC_FLOATS = GC_P3 + 3072;
static_assert(GC_FLOATS * 4 <= 160 * 1024, "cooperative Gram tile must fit the LDS");

template <int NTT>
__device__ __forceinline__ void conv_mfma_lds(const float* frags, int lane, int t, int v0, int nv, const float* act, f32x16& acc) {
  const int p = lane & 31, h = lane >> 5;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < nv) {
      const int v = v0 + u;
      const f32x4 a = *(const f32x4*)(frags + (v * NTT + t) * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], act[(8 * v + 4 * h + j) * 33 + p], acc);
    }
  }
}

__device__ __forceinline__ void bias_init1(f32x16& acc, const float* bias, int t, int h, bool zero) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *(const f32x4*)(bias + 32 * t + 8 * q + 4 * h);
    acc[4 * q + 0] = zero ? 0.0f : b[0]; acc[4 * q + 1] = zero ? 0.0f : b[1]; acc[4 * q + 2] = zero ? 0.0f : b[2]; acc[4 * q + 3] = zero ? 0.0f : b[3];
  }
}

__device__ __forceinline__ void gram_tiles_coop(const GramJob& J, int blk, float* sm, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) G[r] = 0.0f;
  const long tiles = (J.HW + 31) / 32;
  for (long tile = blk; tile < tiles; tile += J.nblk) {
    const long px = tile * 32 + p;
    const bool valid = px < J.HW;
    const float* row = J.x + (valid ? px : 0) * 64;
    f32x16 xin[2];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const f32x4 xv = *(const f32x4*)(row + 8 * v + 4 * h);
      const f32x4 mv = *(const f32x4*)(sm + GF_MEAN + 8 * v + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) xin[v >> 2][(v & 3) * 4 + j] = xv[j] - mv[j];      // cF - cMean, :59-65
    }
    f32x16 acc;
    // conv1: output tile `wave`
    bias_init1(acc, sm + GF_B1, wave, h, false);
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const f32x4 a = *(const f32x4*)(sm + GF_L1 + (v * 4 + wave) * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], xin[v >> 2][(v & 3) * 4 + j], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[GC_A1 + (32 * wave + 8 * (r >> 2) + 4 * h + (r & 3)) * 33 + p] = lrelu02(acc[r]);
    __syncthreads();
    // conv2: output tile wave & 1, K half wave >> 1
    const int t2 = wave & 1, kh = wave >> 1;
    bias_init1(acc, sm + GF_B2, t2, h, kh != 0);
    conv_mfma_lds<2>(sm + GF_L2, lane, t2, 8 * kh, 8, sm + GC_A1, acc);
    if (kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sm[GC_P2 + t2 * 1024 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (!kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sm[GC_A2 + (32 * t2 + 8 * (r >> 2) + 4 * h + (r & 3)) * 33 + p] = lrelu02(acc[r] + sm[GC_P2 + t2 * 1024 + r * 64 + lane]);
    }
    __syncthreads();
    // conv3: K quarter `wave`
    bias_init1(acc, sm + GF_B3, 0, h, wave != 0);
    conv_mfma_lds<1>(sm + GF_L3, lane, 0, 2 * wave, 2, sm + GC_A2, acc);
    if (wave) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sm[GC_P3 + (wave - 1) * 1024 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (!wave) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (acc[r] + sm[GC_P3 + r * 64 + lane]) + (sm[GC_P3 + 1024 + r * 64 + lane] + sm[GC_P3 + 2048 + r * 64 + lane]);
        sm[GC_H + p * 33 + 8 * (r >> 2) + 4 * h + (r & 3)] = valid ? v : 0.0f;   // padded pixels contribute nothing
      }
    }
    __syncthreads();
    // G += H H^T, steps 4 wave .. 4 wave + 3 (k = pixels 8 wave .. 8 wave + 7)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = sm[GC_H + (2 * (4 * wave + u) + h) * 33 + p];
      G = MFMA32(a, a, G);
    }
    __syncthreads();   // the buffers are rewritten by the next tile
  }
  // cross-wave reduction (fragment area is dead now); D layout: col j = p, row i = (r&3) + 8(r>>2) + 4h
#pragma unroll
  for (int r = 0; r < 16; ++r) sm[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + p] = G[r];
  __syncthreads();
  const f32x4 sg = *(f32x4*)(sm + tid * 4) + *(f32x4*)(sm + 1024 + tid * 4) + *(f32x4*)(sm + 2048 + tid * 4) + *(f32x4*)(sm + 3072 + tid * 4);
  *(f32x4*)(J.gram_partial + (long)blk * 1024 + tid * 4) = sg;
}

__global__ __launch_bounds__(256, 1) void gram_mfma_kernel(GramJob j0, GramJob j1) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const bool second = (int)blockIdx.x >= j0.nblk;
  const GramJob J = second ? j1 : j0;
  const int blk = second ? blockIdx.x - j0.nblk : blockIdx.x;
  const int tid = threadIdx.x;
  gram_stage_weights(J.w, sm, tid);
  if (J.mean) {
    __syncthreads();
    if (tid < 64) { sm[GF_MEAN + tid] = J.mean[tid]; if (blk == 0 && J.mean_out) J.mean_out[tid] = J.mean[tid]; }
    __syncthreads();
  } else {
    gram_mean_from_partials(J.chan_partial, J.chan_rows, J.inv_count, blk == 0 ? J.mean_out : nullptr, sm, tid);
  }
  if (J.coop) gram_tiles_coop(J, blk, sm, tid);
  else gram_tiles(J, blk, sm, tid);
}

static int gram_blocks(long HW) {
  const long b = ((HW + 31) / 32 + 3) / 4;
  return (int)(b < CROSSRAY_GRAM_BLOCKS ? (b < 1 ? 1 : b) : CROSSRAY_GRAM_BLOCKS);
}

// small grids: one tile per workgroup, the four waves sharing it (gram_tiles_coop); at most `cap` workgroups
constexpr long GRAM_COOP_MAX_PIXELS = 4096;
static void gram_plan(GramJob& g, int cap = CROSSRAY_GRAM_BLOCKS) {
  if (g.nblk <= 0 || g.HW > GRAM_COOP_MAX_PIXELS) return;
  const long tiles = (g.HW + 31) / 32;
  g.coop = 1;
  g.nblk = (int)(tiles < cap ? tiles : cap);
}

static int launch_gram(const GramJob& a, const GramJob& b, hipStream_t stream) {
  const size_t shmem = (size_t)((a.coop || b.coop) ? GC_FLOATS : GF_FLOATS) * 4;
  if (int rc = ensure_dynamic_lds((const void*)gram_mfma_kernel, shmem, "gram_mfma_kernel")) return rc;
  hipLaunchKernelGGL(gram_mfma_kernel, dim3(a.nblk + b.nblk), dim3(256), shmem, stream, a, b);
  return 0;
}

int launch_crossray_gram(const float* x, long HW, const float* mean, const CnnTensors& w, float* gram_sum, float* workspace,
                         hipStream_t stream) {
  if (HW <= 0) return set_error(-2, "crossray_gram: empty grid");
  GramJob a{x, HW, mean, nullptr, 0, 0.0f, w, workspace, nullptr, gram_blocks(HW)};
  gram_plan(a);
  GramJob none{};
  none.nblk = 0;
  if (int rc = launch_gram(a, none, stream)) return rc;
  RedJob r{workspace, a.nblk, gram_sum}, rn{nullptr, 0, nullptr};
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(16, 1), dim3(RED_THREADS), 0, stream, r, rn, 1024);
  return check_launch("crossray_gram");
}

// ---------------------------------------------------------------- M = fc(G_sum / count): one wave per output row
struct FcJob { const float* gram_sum; float inv_count; const float* fc_w; const float* fc_b; float* out; };

__global__ __launch_bounds__(256) void gram_fc_kernel(FcJob j0, FcJob j1) {
  const FcJob j = blockIdx.y ? j1 : j0;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float* wr = j.fc_w + (long)row * 1024;
  float a = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x4 wv = *(const f32x4*)(wr + k * 256 + lane * 4);
    const f32x4 gv = *(const f32x4*)(j.gram_sum + k * 256 + lane * 4);
    a = fmaf(wv[0], gv[0] * j.inv_count, a); a = fmaf(wv[1], gv[1] * j.inv_count, a);
    a = fmaf(wv[2], gv[2] * j.inv_count, a); a = fmaf(wv[3], gv[3] * j.inv_count, a);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
  if (lane == 0) j.out[row] = a + j.fc_b[row];
}

int launch_crossray_matrix(const float* gram_sum, double count, const float* fc_w, const float* fc_b, float* out, hipStream_t stream) {
  if (!(count > 0)) return set_error(-2, "crossray_matrix: count must be positive");
  FcJob j{gram_sum, (float)(1.0 / count), fc_w, fc_b, out};
  hipLaunchKernelGGL(gram_fc_kernel, dim3(256, 1), dim3(256), 0, stream, j, j);
  return check_launch("crossray_matrix");
}

// Small grids: the row reduction of the partial Grams and both fc layers in one launch.  (32, 2) workgroups; each sums its job's
// <= 128 partial rows into LDS (4 KiB per row and workgroup from L2), then 32 rows of M = fc(G_sum / count), 8 per wave, the 16
// weight loads of four rows in flight together.  Workgroup 0 of a job also stores the summed Gram (the backward reads it).
struct FcSmallJob { const float* gram_partial; int rows; float* gram_sum; float inv_count; const float* fc_w; const float* fc_b; float* out; };

__global__ __launch_bounds__(256) void gram_fc_small_kernel(FcSmallJob j0, FcSmallJob j1) {
  __shared__ __attribute__((aligned(16))) float g[1024];
  const FcSmallJob j = blockIdx.y ? j1 : j0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = blockIdx.x;
  {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
    for (int r = 0; r < j.rows; ++r) acc += *(const f32x4*)(j.gram_partial + (long)r * 1024 + tid * 4);
    *(f32x4*)(g + tid * 4) = acc;
    if (li == 0) *(f32x4*)(j.gram_sum + tid * 4) = acc;
  }
  __syncthreads();
#pragma unroll
  for (int r0 = 0; r0 < 8; r0 += 4) {
    f32x4 wv[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int k = 0; k < 4; ++k) wv[rr][k] = *(const f32x4*)(j.fc_w + (long)(li * 32 + wave * 8 + r0 + rr) * 1024 + k * 256 + lane * 4);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 gv = *(const f32x4*)(g + k * 256 + lane * 4);
        acc = fmaf(wv[rr][k][0], gv[0] * j.inv_count, acc); acc = fmaf(wv[rr][k][1], gv[1] * j.inv_count, acc);
        acc = fmaf(wv[rr][k][2], gv[2] * j.inv_count, acc); acc = fmaf(wv[rr][k][3], gv[3] * j.inv_count, acc);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
      const int row = li * 32 + wave * 8 + r0 + rr;
      if (lane == 0) j.out[row] = acc + j.fc_b[row];
    }
  }
}

// ---------------------------------------------------------------- fold: A[3][64], v[3]  (single 256-thread block)
struct FoldLds {
  float S[32][33], Cm[32][33], T[32][33], U[3][32], P[3][64], Q[3][32];
  float Wrgb[3][64], Wun[64][33], Wcomp[32][65], bun[64], bcomp[32], cmean[64];
};

// Everything the fold reads (two 32x32 matrices, ~4.6k weights) goes to LDS in ONE batch of coalesced loads; the chain of small
// products then runs out of LDS.  (Read on demand, the last step alone was 160 dependent global loads in three threads: 8 us for
// 0.1 MFLOP.)  fold_load_weights: what does not depend on the Gram matrices (may run before they exist); fold_compute: the rest.
// Result: aff[0:192] = A, aff[192:195] = v (any address space the caller likes: LDS for the fused decode, global otherwise).
__device__ __forceinline__ void fold_load_weights(FoldLds& L, const FoldTensors& w, int t) {
  for (int e = t; e < 2048; e += 256) {
    L.Wun[e >> 5][e & 31] = w.unzip_w[e];         // [64][32]
    L.Wcomp[e >> 6][e & 63] = w.comp_w[e];        // [32][64]
  }
  if (t < 192) L.Wrgb[t >> 6][t & 63] = w.rgb_w[t];
  if (t < 32) L.bcomp[t] = w.comp_b[t];
}

__device__ __forceinline__ void fold_compute(FoldLds& L, const float* sM, const float* cM, const float* c_mean, const float* s_mean,
                                             const FoldTensors& w, float* aff, int t) {
  for (int e = t; e < 1024; e += 256) {
    L.S[e >> 5][e & 31] = sM[e];
    L.Cm[e >> 5][e & 31] = cM[e];
  }
  if (t < 64) { L.bun[t] = w.unzip_b[t] + s_mean[t]; L.cmean[t] = c_mean[t]; }
  __syncthreads();
  // rgb_pre = Wrgb (Wunzip (sMatrix cMatrix (Wcomp x ...))) (linearStyleTransfer.py:86-89 + the rgb conv), folded from the LEFT:
  // U = Wrgb Wunzip (3x32), then U sMatrix, then (U sMatrix) cMatrix -- three 3-row products instead of the 32x32x32
  // T = sMatrix cMatrix first (which was most of this kernel's arithmetic and one more pass over LDS)
  if (t < 96) {
    const int r = t >> 5, c = t & 31;
    float a = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) a = fmaf(L.Wrgb[r][k], L.Wun[k][c], a);
    L.U[r][c] = a;
  }
  __syncthreads();
  if (t < 96) {                                   // U S (3x32), kept in T[0..2]
    const int r = t >> 5, c = t & 31;
    float a = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) a = fmaf(L.U[r][k], L.S[k][c], a);
    L.T[r][c] = a;
  }
  __syncthreads();
  if (t < 96) {                                   // Q = (U S) cMatrix (3x32)
    const int r = t >> 5, c = t & 31;
    float a = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) a = fmaf(L.T[r][k], L.Cm[k][c], a);
    L.Q[r][c] = a;
  }
  __syncthreads();
  if (t < 192) {                                  // A = Q @ Wcomp (3x64)
    const int r = t >> 6, c = t & 63;
    float a = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) a = fmaf(L.Q[r][k], L.Wcomp[k][c], a);
    L.P[r][c] = a;
    aff[r * 64 + c] = a;
  }
  __syncthreads();
  // v = Q bcomp - A cMean + Wrgb (bunzip + sMean) + brgb: wave r sums row r, one term per lane
  if (t < 192) {
    const int r = t >> 6, k = t & 63;
    float a = (k < 32 ? L.Q[r][k] * L.bcomp[k] : 0.0f) - L.P[r][k] * L.cmean[k] + L.Wrgb[r][k] * L.bun[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    if (k == 0) aff[192 + r] = a + w.rgb_b[r];
  }
}

__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ sM, const float* __restrict__ cM, const float* __restrict__ c_mean,
                                                   const float* __restrict__ s_mean, FoldTensors w, float* __restrict__ affine) {
  __shared__ FoldLds L;
  const int t = threadIdx.x;
  if (sM) {
    fold_load_weights(L, w, t);
    fold_compute(L, sM, cM, c_mean, s_mean, w, affine, t);
  } else {
    // type == "content": decoder only                          linearStyleTransfer.py:285-287
    if (t < 192) affine[t] = w.rgb_w[t];
    if (t < 3) affine[192 + t] = w.rgb_b[t];
  }
}

int launch_crossray_fold(const float* sM, const float* cM, const float* c_mean, const float* s_mean, const FoldTensors& w,
                         float* affine, hipStream_t stream) {
  hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(256), 0, stream, sM, cM, c_mean, s_mean, w, affine);
  return check_launch("crossray_fold");
}

// ---------------------------------------------------------------- apply: rgb[c][px] = sigmoid(A[c] . x[px] + v[c])
// A: the folded affine in LDS.  4 lanes per pixel, 16 channels each; workgroup `blk` of `nblk` strides over the pixels.
__device__ __forceinline__ void apply_pixels(const float* __restrict__ x, long HW, const float* A, float* __restrict__ rgb, long plane_stride, int blk,
                                             int nblk, int tid) {
  const int q = tid & 3;
  for (long px = ((long)blk * 256 + tid) >> 2;; px += ((long)nblk * 256) >> 2) {
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

__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ x, long HW, const float* __restrict__ affine,
                                                    float* __restrict__ rgb, long plane_stride) {
  __shared__ float A[196];
  if (threadIdx.x < 195) A[threadIdx.x] = affine[threadIdx.x];
  __syncthreads();
  apply_pixels(x, HW, A, rgb, plane_stride, blockIdx.x, gridDim.x, threadIdx.x);
}

// Small grids: fold and apply in one launch -- every workgroup folds for itself (12 KiB of L2 reads, a few microseconds of
// arithmetic) instead of waiting for a one-workgroup launch to do it; workgroup 0 also stores the affine map (the backward reads it).
__global__ __launch_bounds__(256) void fold_apply_kernel(const float* __restrict__ sM, const float* __restrict__ cM, const float* __restrict__ c_mean,
                                                         const float* __restrict__ s_mean, FoldTensors w, float* __restrict__ affine,
                                                         const float* __restrict__ x, long HW, float* __restrict__ rgb, long plane_stride) {
  __shared__ FoldLds L;
  __shared__ float A[196];
  const int t = threadIdx.x;
  fold_load_weights(L, w, t);
  fold_compute(L, sM, cM, c_mean, s_mean, w, A, t);
  __syncthreads();
  if (blockIdx.x == 0 && t < 195) affine[t] = A[t];
  apply_pixels(x, HW, A, rgb, plane_stride, blockIdx.x, gridDim.x, t);
}

int launch_crossray_apply(const float* x, long HW, const float* affine, float* rgb, long plane_stride, hipStream_t stream) {
  if (HW <= 0) return 0;
  const long blocks = (HW * 4 + 255) / 256;
  const int grid = (int)(blocks < 2048 ? blocks : 2048);
  hipLaunchKernelGGL(apply_kernel, dim3(grid), dim3(256), 0, stream, x, HW, affine, rgb, plane_stride);
  return check_launch("crossray_apply");
}

// ---------------------------------------------------------------- single-GPU decode: 6 launches, one host call
// workspace layout (floats)
constexpr size_t WS_SUMP0 = 0;                                         // [CROSSRAY_SUM_BLOCKS][64]
constexpr size_t WS_SUMP1 = WS_SUMP0 + (size_t)CROSSRAY_SUM_BLOCKS * 64;
constexpr size_t WS_GRAMP0 = WS_SUMP1 + (size_t)CROSSRAY_SUM_BLOCKS * 64;  // [CROSSRAY_GRAM_BLOCKS][1024]
constexpr size_t WS_GRAMP1 = WS_GRAMP0 + (size_t)CROSSRAY_GRAM_BLOCKS * 1024;
constexpr size_t WS_STATS = WS_GRAMP1 + (size_t)CROSSRAY_GRAM_BLOCKS * 1024;
constexpr size_t ST_CMEAN = 0, ST_SMEAN = 64, ST_CGRAM = 128, ST_SGRAM = 128 + 1024, ST_CMAT = 128 + 2048, ST_SMAT = 128 + 3072,
                 ST_AFFINE = 128 + 4096, ST_END = ST_AFFINE + 256;
static_assert((WS_STATS + ST_END) * 4 <= CROSSRAY_WORKSPACE_BYTES, "workspace too small");

int launch_crossray_decode(const DecodeArgs& d, hipStream_t stream) {
  if (d.HW <= 0) return 0;
  float* ws = (float*)d.workspace;
  float* st = ws + WS_STATS;
  if (!d.style) {  // type == "content"
    if (int rc = launch_crossray_fold(nullptr, nullptr, nullptr, nullptr, d.lin, st + ST_AFFINE, stream)) return rc;
    return launch_crossray_apply(d.content, d.HW, st + ST_AFFINE, d.rgb, d.plane_stride, stream);
  }
  if (d.HWs <= 0) return set_error(-2, "crossray_decode: empty style grid");
  SumJob s0{d.content, d.HW, ws + WS_SUMP0, chansum_blocks(d.HW)}, s1{d.style, d.HWs, ws + WS_SUMP1, chansum_blocks(d.HWs)};
  hipLaunchKernelGGL(chansum_partial_kernel, dim3(s0.nblk + s1.nblk), dim3(SUM_THREADS), 0, stream, s0, s1);
  GramJob g0{d.content, d.HW, nullptr, ws + WS_SUMP0, s0.nblk, (float)(1.0 / (double)d.HW), d.cnet, ws + WS_GRAMP0, st + ST_CMEAN, gram_blocks(d.HW)};
  GramJob g1{d.style, d.HWs, nullptr, ws + WS_SUMP1, s1.nblk, (float)(1.0 / (double)d.HWs), d.snet, ws + WS_GRAMP1, st + ST_SMEAN, gram_blocks(d.HWs)};
  gram_plan(g0); gram_plan(g1);
  if (int rc = launch_gram(g0, g1, stream)) return rc;
  if (g0.coop && g1.coop) {   // small grids (the headline 32 x 32): four launches instead of six
    FcSmallJob f0{ws + WS_GRAMP0, g0.nblk, st + ST_CGRAM, (float)(1.0 / (double)d.HW), d.cnet_fc_w, d.cnet_fc_b, st + ST_CMAT};
    FcSmallJob f1{ws + WS_GRAMP1, g1.nblk, st + ST_SGRAM, (float)(1.0 / (double)d.HWs), d.snet_fc_w, d.snet_fc_b, st + ST_SMAT};
    hipLaunchKernelGGL(gram_fc_small_kernel, dim3(32, 2), dim3(256), 0, stream, f0, f1);
    hipLaunchKernelGGL(fold_apply_kernel, dim3((unsigned)((d.HW + 63) / 64)), dim3(256), 0, stream, st + ST_SMAT, st + ST_CMAT, st + ST_CMEAN, st + ST_SMEAN,
                       d.lin, st + ST_AFFINE, d.content, d.HW, d.rgb, d.plane_stride);
    return check_launch("crossray_decode");
  }
  RedJob r0{ws + WS_GRAMP0, g0.nblk, st + ST_CGRAM}, r1{ws + WS_GRAMP1, g1.nblk, st + ST_SGRAM};
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(16, 2), dim3(RED_THREADS), 0, stream, r0, r1, 1024);
  FcJob f0{st + ST_CGRAM, (float)(1.0 / (double)d.HW), d.cnet_fc_w, d.cnet_fc_b, st + ST_CMAT};
  FcJob f1{st + ST_SGRAM, (float)(1.0 / (double)d.HWs), d.snet_fc_w, d.snet_fc_b, st + ST_SMAT};
  hipLaunchKernelGGL(gram_fc_kernel, dim3(256, 2), dim3(256), 0, stream, f0, f1);
  hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(256), 0, stream, st + ST_SMAT, st + ST_CMAT, st + ST_CMEAN, st + ST_SMEAN, d.lin, st + ST_AFFINE);
  if (int rc = check_launch("crossray_decode")) return rc;
  return launch_crossray_apply(d.content, d.HW, st + ST_AFFINE, d.rgb, d.plane_stride, stream);
}

// Ray-sharded decode (SURVEY 8e option B): three host calls around the two all-reduces of
// parallel.decode_sharded.  xchg[0:64] = channel sums, xchg[64:1088] = Gram sums of the CONTENT grid:
//   phase 0: local content sums -> xchg[0:64]            (style sums stay in the workspace)       | all-reduce xchg[0:64]
//   phase 1: global sums + global count -> mean; local content Gram sums -> xchg[64:1088]; style Gram  | all-reduce xchg[64:]
//   phase 2: global Gram -> both fc layers, fold, apply on the local pixels -> rgb
// The style grid is replicated, so its statistics never need a collective.
int launch_crossray_decode_sharded(const DecodeArgs& d, int phase, float* xchg, double count_global, hipStream_t stream) {
  if (!d.style || d.HWs <= 0) return set_error(-2, "crossray_decode_sharded: needs the (replicated) style grid");
  if (!(count_global > 0) && phase > 0) return set_error(-2, "crossray_decode_sharded: global pixel count must be positive");
  float* ws = (float*)d.workspace;
  float* st = ws + WS_STATS;
  const bool have = d.HW > 0;                         // a rank may hold no pixels
  SumJob s0{d.content, d.HW, ws + WS_SUMP0, have ? chansum_blocks(d.HW) : 0}, s1{d.style, d.HWs, ws + WS_SUMP1, chansum_blocks(d.HWs)};
  if (phase == 0) {
    hipLaunchKernelGGL(chansum_partial_kernel, dim3(s0.nblk + s1.nblk), dim3(SUM_THREADS), 0, stream, s0, s1);
    if (have) {
      RedJob r0{ws + WS_SUMP0, s0.nblk, xchg}, rn{nullptr, 0, nullptr};
      hipLaunchKernelGGL(reduce_rows_kernel, dim3(1, 1), dim3(RED_THREADS), 0, stream, r0, rn, 64);
    } else if (hipMemsetAsync(xchg, 0, 64 * sizeof(float), stream) != hipSuccess) return set_error(-10, "hipMemsetAsync failed");
    return check_launch("crossray_decode_sharded phase 0");
  }
  if (phase == 1) {
    GramJob g0{d.content, d.HW, nullptr, xchg, 1, (float)(1.0 / count_global), d.cnet, ws + WS_GRAMP0, st + ST_CMEAN, have ? gram_blocks(d.HW) : 0};
    GramJob g1{d.style, d.HWs, nullptr, ws + WS_SUMP1, s1.nblk, (float)(1.0 / (double)d.HWs), d.snet, ws + WS_GRAMP1, st + ST_SMEAN, gram_blocks(d.HWs)};
    gram_plan(g0); gram_plan(g1);
    if (int rc = launch_gram(g0, g1, stream)) return rc;
    RedJob r0{ws + WS_GRAMP0, g0.nblk, have ? xchg + 64 : nullptr}, r1{ws + WS_GRAMP1, g1.nblk, st + ST_SGRAM};
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(16, 2), dim3(RED_THREADS), 0, stream, r0, r1, 1024);
    if (!have) {
      if (hipMemsetAsync(xchg + 64, 0, 1024 * sizeof(float), stream) != hipSuccess) return set_error(-10, "hipMemsetAsync failed");
      // the mean is needed by fold on every rank
      GramJob m0{d.style, 0, nullptr, xchg, 1, (float)(1.0 / count_global), d.cnet, ws + WS_GRAMP0, st + ST_CMEAN, 1}, none{};
      none.nblk = 0;
      if (int rc = launch_gram(m0, none, stream)) return rc;
    }
    return check_launch("crossray_decode_sharded phase 1");
  }
  FcJob f0{xchg + 64, (float)(1.0 / count_global), d.cnet_fc_w, d.cnet_fc_b, st + ST_CMAT};
  FcJob f1{st + ST_SGRAM, (float)(1.0 / (double)d.HWs), d.snet_fc_w, d.snet_fc_b, st + ST_SMAT};
  hipLaunchKernelGGL(gram_fc_kernel, dim3(256, 2), dim3(256), 0, stream, f0, f1);
  if (d.HW > 0 && d.HW <= GRAM_COOP_MAX_PIXELS) {   // small shards: fold and apply in one launch, as in the single-GPU decode
    hipLaunchKernelGGL(fold_apply_kernel, dim3((unsigned)((d.HW + 63) / 64)), dim3(256), 0, stream, st + ST_SMAT, st + ST_CMAT, st + ST_CMEAN, st + ST_SMEAN,
                       d.lin, st + ST_AFFINE, d.content, d.HW, d.rgb, d.plane_stride);
    return check_launch("crossray_decode_sharded phase 2");
  }
  hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(256), 0, stream, st + ST_SMAT, st + ST_CMAT, st + ST_CMEAN, st + ST_SMEAN, d.lin, st + ST_AFFINE);
  if (int rc = check_launch("crossray_decode_sharded phase 2")) return rc;
  return launch_crossray_apply(d.content, d.HW, st + ST_AFFINE, d.rgb, d.plane_stride, stream);
}

}  // namespace crnerf

// ================================================================= backward of the decode (training)
// What autograd derives from style_net.forward in the reference.  The forward stats (means, Grams, fc
// outputs, folded affine) are re-created by one more forward decode into the workspace; then
//   pre    : per pixel  d_pre = d_rgb * rgb (1-rgb),  dx = d_pre A            (direct path of x)
//   wgrad  : dA = sum_px d_pre^T x, dv = sum_px d_pre                          (point-reduction GEMM, mlp_train16.hip)
//   small  : dA, dv -> grads of compress / unzip / rgb convs, dT -> dsMatrix, dcMatrix, mean terms
//   fc     : d fc_w = dm (x) g,  d fc_b = dm,  dg = fc_w^T dm   for both CNNs
//   chain  : per pixel (content and style): recompute the 1x1-conv chain, dh3 = (dG + dG^T) h3, back through
//            the three convs; the per-layer deltas/activations go to HBM for
//   wgrad  : the six conv weight/bias gradients
//   finish : dx += dxc - mean(dxc) + dmean/HW   (centering, linearStyleTransfer.py:59-65), same for the style grid
namespace crnerf {

__global__ __launch_bounds__(256) void dec_bwd_pre_kernel(const float* __restrict__ x, long HW, const float* __restrict__ affine,
                                                          const float* __restrict__ d_rgb, long plane_stride,
                                                          float* __restrict__ d_pre /*[HW,4]*/, float* __restrict__ dx /*[HW,64]*/) {
  __shared__ float A[196];
  if (threadIdx.x < 195) A[threadIdx.x] = affine[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x & 3;
  const long px = ((long)blockIdx.x * 256 + threadIdx.x) >> 2;
  const bool valid = px < HW;
  float r = 0.0f, g = 0.0f, b = 0.0f;
  f32x4 xv[4];
  if (valid) {
    const float* row = x + px * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * q + 16 * k;
      xv[k] = *(const f32x4*)(row + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r = fmaf(A[c + e], xv[k][e], r); g = fmaf(A[64 + c + e], xv[k][e], g); b = fmaf(A[128 + c + e], xv[k][e], b);
      }
    }
  }
  r += __shfl_xor(r, 1); g += __shfl_xor(g, 1); b += __shfl_xor(b, 1);
  r += __shfl_xor(r, 2); g += __shfl_xor(g, 2); b += __shfl_xor(b, 2);
  if (!valid) return;
  float dp[3];
  const float pre[3] = {r + A[192], g + A[193], b + A[194]};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float s = 1.0f / (1.0f + expf(-pre[c]));
    dp[c] = d_rgb[c * plane_stride + px] * s * (1.0f - s);
  }
  if (q == 0) *(f32x4*)(d_pre + px * 4) = f32x4{dp[0], dp[1], dp[2], 0.0f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * q + 16 * k;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = dp[0] * A[c + e] + dp[1] * A[64 + c + e] + dp[2] * A[128 + c + e];
    *(f32x4*)(dx + px * 64 + c) = o;
  }
}

struct SmallBwd {
  const float* dA; const float* dv;                  // [3,64] (ld 64), [3]
  const float* c_mean; const float* s_mean; const float* sM; const float* cM;
  FoldTensors w;
  float* g_comp_w; float* g_comp_b; float* g_unz_w; float* g_unz_b; float* g_rgb_w; float* g_rgb_b;
  float* dm_s; float* dm_c;                          // [1024] each (= d fc_b)
  float* dc_mean; float* ds_mean;                    // [64] each: the mean terms that do not go through the Gram
};

__global__ __launch_bounds__(256) void dec_bwd_small_kernel(SmallBwd a) {
  __shared__ float T[32][33], U[3][32], Q[3][32], Am[3][64], dAp[3][64], dQ[3][32], dU[3][32], dT[32][33], dv[3];
  const int t = threadIdx.x;
  if (t < 3) dv[t] = a.dv[t];
  for (int e = t; e < 1024; e += 256) {
    const int i = e >> 5, j = e & 31;
    float s = 0.0f;
    for (int k = 0; k < 32; ++k) s = fmaf(a.sM[i * 32 + k], a.cM[k * 32 + j], s);
    T[i][j] = s;
  }
  if (t < 96) {
    const int r = t / 32, k = t % 32;
    float s = 0.0f;
    for (int c = 0; c < 64; ++c) s = fmaf(a.w.rgb_w[r * 64 + c], a.w.unzip_w[c * 32 + k], s);
    U[r][k] = s;
  }
  __syncthreads();
  if (t < 96) {
    const int r = t / 32, j = t % 32;
    float s = 0.0f;
    for (int k = 0; k < 32; ++k) s = fmaf(U[r][k], T[k][j], s);
    Q[r][j] = s;
  }
  __syncthreads();
  if (t < 192) {
    const int r = t / 64, c = t % 64;
    float s = 0.0f;
    for (int k = 0; k < 32; ++k) s = fmaf(Q[r][k], a.w.comp_w[k * 64 + c], s);
    Am[r][c] = s;
    dAp[r][c] = a.dA[r * 64 + c] - dv[r] * a.c_mean[c];
  }
  __syncthreads();
  if (t < 96) {
    const int r = t / 32, k = t % 32;
    float s = dv[r] * a.w.comp_b[k];
    for (int c = 0; c < 64; ++c) s = fmaf(dAp[r][c], a.w.comp_w[k * 64 + c], s);
    dQ[r][k] = s;
  }
  for (int e = t; e < 32 * 64; e += 256) {          // d compress.weight
    const int k = e / 64, c = e % 64;
    a.g_comp_w[e] = Q[0][k] * dAp[0][c] + Q[1][k] * dAp[1][c] + Q[2][k] * dAp[2][c];
  }
  if (t < 32) a.g_comp_b[t] = Q[0][t] * dv[0] + Q[1][t] * dv[1] + Q[2][t] * dv[2];
  if (t < 64) {
    a.dc_mean[t] = -(Am[0][t] * dv[0] + Am[1][t] * dv[1] + Am[2][t] * dv[2]);
    const float wv = a.w.rgb_w[t] * dv[0] + a.w.rgb_w[64 + t] * dv[1] + a.w.rgb_w[128 + t] * dv[2];
    a.g_unz_b[t] = wv;
    a.ds_mean[t] = wv;
  }
  if (t < 3) a.g_rgb_b[t] = dv[t];
  __syncthreads();
  if (t < 96) {
    const int r = t / 32, k = t % 32;
    float s = 0.0f;
    for (int j = 0; j < 32; ++j) s = fmaf(dQ[r][j], T[k][j], s);
    dU[r][k] = s;
  }
  for (int e = t; e < 1024; e += 256) {
    const int k = e >> 5, j = e & 31;
    dT[k][j] = U[0][k] * dQ[0][j] + U[1][k] * dQ[1][j] + U[2][k] * dQ[2][j];
  }
  __syncthreads();
  if (t < 192) {                                    // d feat_2_rgb weight
    const int r = t / 64, c = t % 64;
    float s = dv[r] * (a.w.unzip_b[c] + a.s_mean[c]);
    for (int k = 0; k < 32; ++k) s = fmaf(dU[r][k], a.w.unzip_w[c * 32 + k], s);
    a.g_rgb_w[r * 64 + c] = s;
  }
  for (int e = t; e < 64 * 32; e += 256) {          // d unzip.weight
    const int c = e / 32, k = e % 32;
    a.g_unz_w[e] = a.w.rgb_w[c] * dU[0][k] + a.w.rgb_w[64 + c] * dU[1][k] + a.w.rgb_w[128 + c] * dU[2][k];
  }
  for (int e = t; e < 1024; e += 256) {             // T = sM cM
    const int i = e >> 5, k = e & 31;
    float s = 0.0f, c2 = 0.0f;
    for (int j = 0; j < 32; ++j) {
      s = fmaf(dT[i][j], a.cM[k * 32 + j], s);      // dsM[i][k] = sum_j dT[i][j] cM[k][j]
      c2 = fmaf(a.sM[j * 32 + i], dT[j][k], c2);    // dcM[i][k] = sum_j sM[j][i] dT[j][k]
    }
    a.dm_s[e] = s;
    a.dm_c[e] = c2;
  }
}

struct FcBwd { const float* dm; const float* gram_sum; float inv_count; const float* fc_w; float* g_fc_w; float* g_fc_b; float* S; };
// grid (256, 2): rows 4b..4b+3 of the outer product d fc_w = dm (x) g; blocks 0..3 also form
// S = (dG + dG^T) * inv_count with dG = (fc_w^T dm).view(32,32): the factor the Gram backward applies per pixel
__global__ __launch_bounds__(256) void dec_bwd_fc_kernel(FcBwd j0, FcBwd j1) {
  const FcBwd j = blockIdx.y ? j1 : j0;
  const int t = threadIdx.x;
  for (int r = 0; r < 4; ++r) {
    const int i = blockIdx.x * 4 + r;
    const float d = j.dm[i];
    for (int c = t; c < 1024; c += 256) j.g_fc_w[(long)i * 1024 + c] = d * (j.gram_sum[c] * j.inv_count);
    if (t == 0) j.g_fc_b[i] = d;
  }
  if (blockIdx.x < 4) {
    const int col = blockIdx.x * 256 + t;
    float s = 0.0f;
    for (int i = 0; i < 1024; ++i) s = fmaf(j.fc_w[(long)i * 1024 + col], j.dm[i], s);
    j.S[1024 + col] = s;                               // raw dg, second half of the S buffer
  }
}
__global__ void dec_bwd_sym_kernel(float* S0, float inv0, float* S1, float inv1) {
  float* S = blockIdx.x ? S1 : S0;
  const float inv = blockIdx.x ? inv1 : inv0;
  for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
    const int a = e >> 5, b = e & 31;
    S[e] = (S[1024 + a * 32 + b] + S[1024 + b * 32 + a]) * inv;   // dL/dG_sum + its transpose, G = G_sum / count
  }
}

// per-pixel chain forward + backward; stores what wgrad needs
struct ChainJob {
  const float* x; long P; const float* mean; CnnTensors w; const float* S;
  float* xc; float* h1; float* h2; float* d1; float* d2; float* d3; float* dxc; int nblk;
};
constexpr int CH_W1 = 0, CH_W2 = 128 * 64, CH_W3 = CH_W2 + 64 * 128, CH_B1 = CH_W3 + 32 * 64, CH_B2 = CH_B1 + 128, CH_B3 = CH_B2 + 64,
              CH_MEAN = CH_B3 + 32, CH_S = CH_MEAN + 64, CH_FLOATS = CH_S + 1024;

__device__ __forceinline__ float dlrelu(float h) { return h > 0.0f ? 1.0f : 0.2f; }

__global__ __launch_bounds__(64) void dec_bwd_chain_kernel(ChainJob j0, ChainJob j1) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const bool second = (int)blockIdx.x >= j0.nblk;
  const ChainJob J = second ? j1 : j0;
  const int blk = second ? blockIdx.x - j0.nblk : blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < 128 * 64; i += 64) { sm[CH_W1 + i] = J.w.w1[i]; sm[CH_W2 + i] = J.w.w2[i]; }
  for (int i = tid; i < 32 * 64; i += 64) sm[CH_W3 + i] = J.w.w3[i];
  for (int i = tid; i < 128; i += 64) sm[CH_B1 + i] = J.w.b1[i];
  sm[CH_B2 + tid] = J.w.b2[tid];
  sm[CH_MEAN + tid] = J.mean[tid];
  if (tid < 32) sm[CH_B3 + tid] = J.w.b3[tid];
  for (int i = tid; i < 1024; i += 64) sm[CH_S + i] = J.S[i];
  __syncthreads();
  for (long px = (long)blk * 64 + tid; px < J.P; px += (long)J.nblk * 64) {
    float xin[64], h2[64];
    const float* row = J.x + px * 64;
#pragma unroll
    for (int c = 0; c < 64; ++c) { xin[c] = row[c] - sm[CH_MEAN + c]; J.xc[px * 64 + c] = xin[c]; }
#pragma unroll
    for (int o = 0; o < 64; ++o) h2[o] = sm[CH_B2 + o];
#pragma unroll 1
    for (int oc = 0; oc < 128; oc += 8) {               // layer 1 in chunks of 8 outputs, folded into layer 2
      float h1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float acc = sm[CH_B1 + oc + u];
        const float* wr = sm + CH_W1 + (oc + u) * 64;
#pragma unroll
        for (int c = 0; c < 64; ++c) acc = fmaf(wr[c], xin[c], acc);
        h1[u] = lrelu02(acc);
        J.h1[px * 128 + oc + u] = h1[u];
      }
#pragma unroll
      for (int o = 0; o < 64; ++o) {
        const float* wr = sm + CH_W2 + o * 128 + oc;
        float acc = h2[o];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(wr[u], h1[u], acc);
        h2[o] = acc;
      }
    }
#pragma unroll
    for (int o = 0; o < 64; ++o) { h2[o] = lrelu02(h2[o]); J.h2[px * 64 + o] = h2[o]; }
    float h3[32], d3[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) {
      float acc = sm[CH_B3 + o];
      const float* wr = sm + CH_W3 + o * 64;
#pragma unroll
      for (int c = 0; c < 64; ++c) acc = fmaf(wr[c], h2[c], acc);
      h3[o] = acc;
    }
#pragma unroll
    for (int a = 0; a < 32; ++a) {                      // dh3 = S h3,  S = (dG + dG^T) / count^... (see launch)
      float acc = 0.0f;
#pragma unroll
      for (int b = 0; b < 32; ++b) acc = fmaf(sm[CH_S + a * 32 + b], h3[b], acc);
      d3[a] = acc;
      J.d3[px * 32 + a] = acc;
    }
    float d2[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) d2[c] = 0.0f;
#pragma unroll 4
    for (int o = 0; o < 32; ++o) {
      const float* wr = sm + CH_W3 + o * 64;
#pragma unroll
      for (int c = 0; c < 64; ++c) d2[c] = fmaf(d3[o], wr[c], d2[c]);
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) { d2[c] *= dlrelu(h2[c]); J.d2[px * 64 + c] = d2[c]; }
    float dxc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) dxc[c] = 0.0f;
#pragma unroll 1
    for (int kc = 0; kc < 128; kc += 8) {
      float d1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d1[u] = 0.0f;
#pragma unroll 4
      for (int o = 0; o < 64; ++o) {
        const float* wr = sm + CH_W2 + o * 128 + kc;
#pragma unroll
        for (int u = 0; u < 8; ++u) d1[u] = fmaf(d2[o], wr[u], d1[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        d1[u] *= dlrelu(J.h1[px * 128 + kc + u]);
        J.d1[px * 128 + kc + u] = d1[u];
        const float* wr = sm + CH_W1 + (kc + u) * 64;
#pragma unroll
        for (int c = 0; c < 64; ++c) dxc[c] = fmaf(d1[u], wr[c], dxc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) J.dxc[px * 64 + c] = dxc[c];
  }
}

// ---- the same chain on the fp32 MFMA (round 2).  One thread per pixel (kernel above) is 40 k broadcast-LDS FMAs in a row:
// 0.3 ms per launch whatever the grid size, 0.9 ms of a 14.8 ms training step at the reference's 1,024-ray batch.  Here a wavefront
// owns 32 pixels, swapped-operand like the Gram kernel: D[feature][pixel] = M[feature][k] . src[k][pixel], activations and deltas
// stay in registers through the three forward and four backward products, the seven matrices (W1, W2, W3, S, W3^T, W2^T, W1^T)
// sit in LDS as A-operand fragments.  frag(v, t)[lane = 32 kk + i][j] = M[32 t + i][8 v + 4 kk + j].
constexpr int CF_L1 = 0;                    // W1   [128 x  64]: 8 k-groups x 4 tiles = 32 fragments
constexpr int CF_L2 = CF_L1 + 32 * 256;     // W2   [ 64 x 128]: 16 x 2 = 32
constexpr int CF_L3 = CF_L2 + 32 * 256;     // W3   [ 32 x  64]: 8 x 1 = 8
constexpr int CF_S = CF_L3 + 8 * 256;       // S    [ 32 x  32]: 4 x 1 = 4
constexpr int CF_T3 = CF_S + 4 * 256;       // W3^T [ 64 x  32]: 4 x 2 = 8
constexpr int CF_T2 = CF_T3 + 8 * 256;      // W2^T [128 x  64]: 8 x 4 = 32
constexpr int CF_T1 = CF_T2 + 32 * 256;     // W1^T [ 64 x 128]: 16 x 2 = 32
constexpr int CF_FRAGS = 148;
constexpr int CF_B1 = CF_T1 + 32 * 256, CF_B2 = CF_B1 + 128, CF_B3 = CF_B2 + 64, CF_MEAN = CF_B3 + 32, CF_FLOATS = CF_MEAN + 64;
static_assert(CF_B1 == CF_FRAGS * 256 && CF_FLOATS * 4 <= 160 * 1024, "chain fragments must fit the 160 KiB LDS");

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&a)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[t][r] = 0.0f;
}
// rows of a pixel-major [P][ld] buffer from the D[feature][pixel] register layout: lane (p, h) holds features 32t + 8q + 4h + 0..3
template <int NT>
__device__ __forceinline__ void store_tiles(float* base, long px, int ld, const f32x16 (&a)[NT], int h, bool valid) {
  if (!valid) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = {a[t][4 * q + 0], a[t][4 * q + 1], a[t][4 * q + 2], a[t][4 * q + 3]};
      *(f32x4*)(base + px * ld + 32 * t + 8 * q + 4 * h) = v;
    }
}

__global__ __launch_bounds__(256, 1) void dec_bwd_chain_mfma_kernel(ChainJob j0, ChainJob j1) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const bool second = (int)blockIdx.x >= j0.nblk;
  const ChainJob J = second ? j1 : j0;
  const int blk = second ? blockIdx.x - j0.nblk : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  // stage the 148 fragments: element (frag, lane-slot l = 32 kk + i) = 4 consecutive columns of row 32 t + i of the matrix
  for (int idx = tid; idx < CF_FRAGS * 64; idx += 256) {
    const int frag = idx >> 6, l = idx & 63, i = l & 31, kk = l >> 5;
    const float* M; int ld, nt, f; bool tr;
    if (frag < 32) { M = J.w.w1; ld = 64; nt = 4; f = frag; tr = false; }
    else if (frag < 64) { M = J.w.w2; ld = 128; nt = 2; f = frag - 32; tr = false; }
    else if (frag < 72) { M = J.w.w3; ld = 64; nt = 1; f = frag - 64; tr = false; }
    else if (frag < 76) { M = J.S; ld = 32; nt = 1; f = frag - 72; tr = false; }
    else if (frag < 84) { M = J.w.w3; ld = 64; nt = 2; f = frag - 76; tr = true; }
    else if (frag < 116) { M = J.w.w2; ld = 128; nt = 4; f = frag - 84; tr = true; }
    else { M = J.w.w1; ld = 64; nt = 2; f = frag - 116; tr = true; }
    const int v = f / nt, t = f - v * nt;
    const int row = 32 * t + i, col = 8 * v + 4 * kk;
    f32x4 val;
    if (!tr) val = *(const f32x4*)(M + row * ld + col);
    else val = f32x4{M[(col + 0) * ld + row], M[(col + 1) * ld + row], M[(col + 2) * ld + row], M[(col + 3) * ld + row]};   // M^T[row][col + j]
    *(f32x4*)(sm + 4 * idx) = val;
  }
  if (tid < 128) sm[CF_B1 + tid] = J.w.b1[tid];
  if (tid < 64) { sm[CF_B2 + tid] = J.w.b2[tid]; sm[CF_MEAN + tid] = J.mean[tid]; }
  if (tid < 32) sm[CF_B3 + tid] = J.w.b3[tid];
  __syncthreads();

  const long tiles = (J.P + 31) / 32;
  for (long tile = (long)wave * J.nblk + blk; tile < tiles; tile += (long)J.nblk * 4) {   // tiles spread over workgroups first
    const long px = tile * 32 + p;
    const bool valid = px < J.P;
    const float* row = J.x + (valid ? px : 0) * 64;
    f32x16 xin[2];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const f32x4 xv = *(const f32x4*)(row + 8 * v + 4 * h);
      const f32x4 mv = *(const f32x4*)(sm + CF_MEAN + 8 * v + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) xin[v >> 2][(v & 3) * 4 + j] = xv[j] - mv[j];
    }
    store_tiles<2>(J.xc, px, 64, xin, h, valid);
    f32x16 a1[4], a2[2], a3[1];
    bias_init<4>(a1, sm + CF_B1, h);
    conv_mfma<4, 8>(sm + CF_L1, lane, xin, a1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[t][r] = lrelu02(a1[t][r]);
    store_tiles<4>(J.h1, px, 128, a1, h, valid);
    bias_init<2>(a2, sm + CF_B2, h);
    conv_mfma<2, 16>(sm + CF_L2, lane, a1, a2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[t][r] = lrelu02(a2[t][r]);
    store_tiles<2>(J.h2, px, 64, a2, h, valid);
    bias_init<1>(a3, sm + CF_B3, h);
    conv_mfma<1, 8>(sm + CF_L3, lane, a2, a3);
    // backward: dh3 = S h3;  d2 = (W3^T dh3) . lrelu'(h2);  d1 = (W2^T d2) . lrelu'(h1);  dxc = W1^T d1
    f32x16 d3[1], d2[2], d1[4], dx[2];
    zero_tiles<1>(d3);
    conv_mfma<1, 4>(sm + CF_S, lane, a3, d3);
    store_tiles<1>(J.d3, px, 32, d3, h, valid);
    zero_tiles<2>(d2);
    conv_mfma<2, 4>(sm + CF_T3, lane, d3, d2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) d2[t][r] *= dlrelu(a2[t][r]);
    store_tiles<2>(J.d2, px, 64, d2, h, valid);
    zero_tiles<4>(d1);
    conv_mfma<4, 8>(sm + CF_T2, lane, d2, d1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) d1[t][r] *= dlrelu(a1[t][r]);
    store_tiles<4>(J.d1, px, 128, d1, h, valid);
    zero_tiles<2>(dx);
    conv_mfma<2, 16>(sm + CF_T1, lane, d1, dx);
    store_tiles<2>(J.dxc, px, 64, dx, h, valid);
  }
}

// dst[px][c] (+)= dxc[px][c] + (dmean[c] - colsum[c]) / P
// count: the pixels the mean was taken over (P, or the GLOBAL pixel count of a ray-sharded grid: colsum is then the global column sum)
__global__ void dec_bwd_finish_kernel(float* __restrict__ dst, const float* __restrict__ dxc, const float* __restrict__ dmean,
                                      const float* __restrict__ colsum, long P, int accumulate, float count) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * 64) return;
  const int c = (int)(idx & 63);
  const float v = dxc[idx] + (dmean[c] - colsum[c]) / count;
  dst[idx] = accumulate ? dst[idx] + v : v;
}

// the six 1x1-conv weight-gradient jobs of the two CNN chains (snet on the style grid, cnet on the content grid): shapes for the plan
static void dec_conv_wgrad_specs(long HW, long HWs, WgradSpec* sp) {
  int n = 0;
  for (int net = 0; net < 2; ++net) {
    const long P = net ? HWs : HW;
    sp[n++] = WgradSpec{nullptr, 128, 128, nullptr, 64, 64, nullptr, 64, (float*)1, wgrad_job_weight(128, 64), 0, P};
    sp[n++] = WgradSpec{nullptr, 64, 64, nullptr, 128, 128, nullptr, 128, (float*)1, wgrad_job_weight(64, 128), 0, P};
    sp[n++] = WgradSpec{nullptr, 32, 32, nullptr, 64, 64, nullptr, 64, (float*)1, wgrad_job_weight(32, 64), 0, P};
  }
}

size_t crossray_backward_workspace_floats(long HW, long HWs) {
  const long P = HW + HWs;
  size_t n = CROSSRAY_WORKSPACE_BYTES / 4;                       // forward decode workspace (stats live here)
  n += (size_t)3 * HW + 4 * (size_t)HW;                          // scratch rgb, d_pre[HW,4]
  n += (size_t)P * (64 + 128 + 64 + 128 + 64 + 32 + 64);         // xc, h1, h2, d1, d2, d3, dxc
  n += 4096 + 2 * 2048 + 1024 + 1152;                            // dA/dv, dm, S/dg buffers, mean terms, column sums, the sharded backward's copy of the forward's sums
  WgradSpec sp[6];
  dec_conv_wgrad_specs(HW, HWs, sp);
  size_t wg = wgrad_batch_ws_floats(sp, 6);
  if (wgrad_workspace_floats(HW, 4, 64) > wg) wg = wgrad_workspace_floats(HW, 4, 64);
  n += wg + 64 * (size_t)CROSSRAY_SUM_BLOCKS * 2;
  return n;
}

int launch_crossray_decode_backward(const DecodeArgs& d, const float* d_rgb, long d_plane_stride, float* workspace, float* d_content,
                                    float* d_style, float* const* grads, hipStream_t stream) {
  return launch_crossray_decode_backward_sharded(d, d_rgb, d_plane_stride, workspace, d_content, d_style, grads, -1, nullptr, (double)d.HW, nullptr, stream);
}

// Ray-sharded backward (round 6; the training twin of launch_crossray_decode_sharded): the content grid is this rank's block of pixels, the style
// grid is replicated.  What crosses pixels in the backward are the same two kinds of sums as in the forward -- the gradient of the folded affine
// (dA [3,64], dv [3]: sums over pixels of d_pre (x) x) and the column sums of the centred chain's input gradient -- so the pass is cut at them:
//   phase 0: the forward's statistics from the GLOBAL sums `fwd_xchg` (channel sums | Gram sums, as the forward's all-reduces left them), d_pre,
//            the direct path of x, dA | dv of the LOCAL pixels -> xb[0:256] | xb[256:320]                       | all-reduce xb[0:320]
//   phase 1: everything that is replicated arithmetic on global quantities (small matrices, both fc layers), the conv chains (local content
//            pixels, the style grid), the six conv weight gradients, the column sums of dxc: content -> xb[320:384]  | all-reduce xb[320:384]
//   phase 2: the centring terms with the global count and the global column sums.
// grads: the content chain's six conv gradients (grads[8..13]) are this rank's PART (sums over its pixels); every other gradient and d_style are
// computed from all-reduced quantities or from the replicated style grid and are the WHOLE gradient on every rank.  phase -1: one GPU, one call.
int launch_crossray_decode_backward_sharded(const DecodeArgs& d, const float* d_rgb, long d_plane_stride, float* workspace, float* d_content,
                                            float* d_style, float* const* grads, int phase, const float* fwd_xchg, double count_global, float* xb,
                                            hipStream_t stream) {
  if (d.HW <= 0 || d.HWs <= 0 || !d.style) return set_error(-2, "crossray_decode_backward: needs a content and a style grid");
  if (phase >= 0 && (!fwd_xchg || !xb || !(count_global > 0))) return set_error(-2, "crossray_decode_backward_sharded: needs the forward's global sums, the exchange buffer and the global pixel count");
  const bool all = phase < 0;
  const float inv_count = (float)(1.0 / count_global);
  const long HW = d.HW, HWs = d.HWs;
  float* p = workspace;
  float* fwd_ws = p; p += CROSSRAY_WORKSPACE_BYTES / 4;
  float* rgb = p; p += 3 * HW;
  float* d_pre = p; p += 4 * HW;
  auto take = [&](size_t n) { float* q = p; p += n; return q; };
  float* xc[2] = {take(HW * 64), take(HWs * 64)};
  float* h1[2] = {take(HW * 128), take(HWs * 128)};
  float* h2[2] = {take(HW * 64), take(HWs * 64)};
  float* d1[2] = {take(HW * 128), take(HWs * 128)};
  float* d2[2] = {take(HW * 64), take(HWs * 64)};
  float* d3[2] = {take(HW * 32), take(HWs * 32)};
  float* dxc[2] = {take(HW * 64), take(HWs * 64)};
  float* dA = take(256); float* dv = take(64);
  float* fx = take(64 + 1024 + 64);                  // sharded: a scratch copy of the forward's exchange block (phases 0 / 1 of the forward rewrite theirs)
  if (!all) { dA = xb; dv = xb + 256; }              // the all-reduced buffers ARE the operands of phase 1
  float* dm_s = take(1024); float* dm_c = take(1024);
  float* S_s = take(2048); float* S_c = take(2048);
  float* dmean_c = take(64); float* dmean_s = take(64); float* cs_c = take(64); float* cs_s = take(64);
  float* sum_ws = take(64 * (size_t)CROSSRAY_SUM_BLOCKS * 2);
  float* wws = p;
  // 0. forward again: stats + affine into fwd_ws
  DecodeArgs f = d;
  f.workspace = fwd_ws; f.rgb = rgb; f.plane_stride = HW;
  float* st = fwd_ws + WS_STATS;
  if (all || phase == 0) {
    if (all) {
      if (int rc = launch_crossray_decode(f, stream)) return rc;
    } else {
      // the sharded forward's three phases with the GLOBAL sums in place of what its all-reduces delivered: phase 0 / 1 write this rank's local
      // sums into the block they are given (a scratch copy), phase 1 / 2 read the global channel sums / Gram from where they are handed over
      if (int rc = launch_crossray_decode_sharded(f, 0, fx + 1088, count_global, stream)) return rc;          // (style partial sums into the workspace)
      if (hipMemcpyAsync(fx, fwd_xchg, 64 * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) return set_error(-10, "hipMemcpyAsync failed");
      if (int rc = launch_crossray_decode_sharded(f, 1, fx, count_global, stream)) return rc;                 // content mean (global), style Gram; fx[64:] = local Gram (unused)
      if (hipMemcpyAsync(fx + 64, fwd_xchg + 64, 1024 * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) return set_error(-10, "hipMemcpyAsync failed");
      if (hipMemcpyAsync(st + ST_CGRAM, fwd_xchg + 64, 1024 * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) return set_error(-10, "hipMemcpyAsync failed");
      if (int rc = launch_crossray_decode_sharded(f, 2, fx, count_global, stream)) return rc;                 // both fc layers, fold -> the affine
    }
    // 1. d_pre, direct dx; 2. dA, dv
    hipLaunchKernelGGL(dec_bwd_pre_kernel, dim3((unsigned)((HW * 4 + 255) / 256)), dim3(256), 0, stream, d.content, HW, st + ST_AFFINE, d_rgb,
                       d_plane_stride, d_pre, d_content);
    wgrad(d_pre, 4, 3, d.content, 64, 64, dA, 64, dv, HW, wws, stream);
    if (!all) return check_launch("crossray_decode_backward phase 0");
  }
  if (phase == 2) {
    hipLaunchKernelGGL(dec_bwd_finish_kernel, dim3((unsigned)((HW * 64 + 255) / 256)), dim3(256), 0, stream, d_content, dxc[0], dmean_c, xb + 320, HW, 1, (float)count_global);
    hipLaunchKernelGGL(dec_bwd_finish_kernel, dim3((unsigned)((HWs * 64 + 255) / 256)), dim3(256), 0, stream, d_style, dxc[1], dmean_s, cs_s, HWs, 0, (float)HWs);
    return check_launch("crossray_decode_backward phase 2");
  }
  // 3. small matrices
  SmallBwd sb{dA, dv, st + ST_CMEAN, st + ST_SMEAN, st + ST_SMAT, st + ST_CMAT, d.lin,
              grads[16], grads[17], grads[18], grads[19], grads[20], grads[21], dm_s, dm_c, dmean_c, dmean_s};
  hipLaunchKernelGGL(dec_bwd_small_kernel, dim3(1), dim3(256), 0, stream, sb);
  // 4. fc layers: snet = grads[6], [7]; cnet = grads[14], [15]
  FcBwd fs{dm_s, st + ST_SGRAM, (float)(1.0 / (double)HWs), d.snet_fc_w, grads[6], grads[7], S_s};
  FcBwd fc{dm_c, st + ST_CGRAM, inv_count, d.cnet_fc_w, grads[14], grads[15], S_c};
  hipLaunchKernelGGL(dec_bwd_fc_kernel, dim3(256, 2), dim3(256), 0, stream, fs, fc);
  // S = (dG + dG^T) / count with dG = dg.view(32,32): G = G_sum / count, G_sum = sum_px h3 h3^T
  hipLaunchKernelGGL(dec_bwd_sym_kernel, dim3(2), dim3(256), 0, stream, S_s, (float)(1.0 / (double)HWs), S_c, inv_count);
  // 5. conv chains
  const int nb_c = (int)((HW + 63) / 64 < 1024 ? (HW + 63) / 64 : 1024), nb_s = (int)((HWs + 63) / 64 < 1024 ? (HWs + 63) / 64 : 1024);
  ChainJob jc{d.content, HW, st + ST_CMEAN, d.cnet, S_c, xc[0], h1[0], h2[0], d1[0], d2[0], d3[0], dxc[0], nb_c};
  ChainJob js{d.style, HWs, st + ST_SMEAN, d.snet, S_s, xc[1], h1[1], h2[1], d1[1], d2[1], d3[1], dxc[1], nb_s};
  static const bool scalar_chain = getenv("CRNERF_CHAIN_SCALAR") != nullptr;     // A/B switch for the round-1 one-thread-per-pixel kernel
  if (scalar_chain) {
    const size_t shmem = (size_t)CH_FLOATS * 4;
    if (int rc = ensure_dynamic_lds((const void*)dec_bwd_chain_kernel, shmem, "dec_bwd_chain_kernel")) return rc;
    hipLaunchKernelGGL(dec_bwd_chain_kernel, dim3(nb_c + nb_s), dim3(64), shmem, stream, jc, js);
  } else {
    const long tc = (HW + 31) / 32, ts = (HWs + 31) / 32;
    jc.nblk = (int)(tc < 256 ? tc : 256);
    js.nblk = (int)(ts < 256 ? ts : 256);
    const size_t shmem = (size_t)CF_FLOATS * 4;
    if (int rc = ensure_dynamic_lds((const void*)dec_bwd_chain_mfma_kernel, shmem, "dec_bwd_chain_mfma_kernel")) return rc;
    hipLaunchKernelGGL(dec_bwd_chain_mfma_kernel, dim3(jc.nblk + js.nblk), dim3(256), shmem, stream, jc, js);
  }
  // 6. conv weight / bias gradients: snet = grads[0..5], cnet = grads[8..13]
  // (one batched launch + one reduction for the six jobs: at these grid sizes a launch per job is mostly ramp-up and drain)
  {
    WgradSpec sp[6];
    dec_conv_wgrad_specs(HW, HWs, sp);
    for (int net = 0; net < 2; ++net) {
      float* const* g = grads + (net ? 0 : 8);
      WgradSpec* q = sp + 3 * net;
      q[0].D = d1[net]; q[0].A = xc[net]; q[0].dst = g[0]; q[0].db = g[1];
      q[1].D = d2[net]; q[1].A = h1[net]; q[1].dst = g[2]; q[1].db = g[3];
      q[2].D = d3[net]; q[2].A = h2[net]; q[2].dst = g[4]; q[2].db = g[5];
    }
    if (int rc = wgrad_batch(sp, 6, wws, wgrad_batch_ws_floats(sp, 6), stream)) return rc;
  }
  // 7. centering terms
  SumJob s0{dxc[0], HW, sum_ws, chansum_blocks(HW)}, s1{dxc[1], HWs, sum_ws + 64 * CROSSRAY_SUM_BLOCKS, chansum_blocks(HWs)};
  hipLaunchKernelGGL(chansum_partial_kernel, dim3(s0.nblk + s1.nblk), dim3(SUM_THREADS), 0, stream, s0, s1);
  RedJob r0{s0.partial, s0.nblk, all ? cs_c : xb + 320}, r1{s1.partial, s1.nblk, cs_s};
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1, 2), dim3(RED_THREADS), 0, stream, r0, r1, 64);
  if (!all) return check_launch("crossray_decode_backward phase 1");       // (the content column sums wait for their all-reduce)
  hipLaunchKernelGGL(dec_bwd_finish_kernel, dim3((unsigned)((HW * 64 + 255) / 256)), dim3(256), 0, stream, d_content, dxc[0], dmean_c, cs_c, HW, 1, (float)HW);
  hipLaunchKernelGGL(dec_bwd_finish_kernel, dim3((unsigned)((HWs * 64 + 255) / 256)), dim3(256), 0, stream, d_style, dxc[1], dmean_s, cs_s, HWs, 0, (float)HWs);
  return check_launch("crossray_decode_backward");
}


// ---------------------------------------------------------------- decoder-only ("content") backward
// style_net.forward(content, None, type="content") = NeuralRenderer alone: rgb = sigmoid(W x + b), W [3,64]
// (linearStyleTransfer.py:285-287, nerf_decoder_stylenerf.py:279-291).  d_pre = d_rgb * rgb (1 - rgb);
// d_x = W^T d_pre; dW = sum_px d_pre (x) x and db = sum_px d_pre through the MFMA point-reduction GEMM.
__global__ __launch_bounds__(256) void content_bwd_kernel(const float* __restrict__ rgb, long rgb_stride, const float* __restrict__ d_rgb,
                                                          long d_stride, const float* __restrict__ W, long HW, float* __restrict__ d_pre,
                                                          float* __restrict__ d_x) {
  const long px = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = threadIdx.x & 63;
  if (px >= HW) return;
  float g[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float y = rgb[o * rgb_stride + px];
    g[o] = d_rgb[o * d_stride + px] * y * (1.0f - y);
  }
  if (c < 3) d_pre[px * 4 + c] = g[c];
  if (c == 3) d_pre[px * 4 + 3] = 0.0f;
  d_x[px * 64 + c] = fmaf(g[0], W[c], fmaf(g[1], W[64 + c], g[2] * W[128 + c]));
}

size_t content_backward_workspace_floats(long HW) { return (size_t)HW * 4 + wgrad_workspace_floats(HW, 3, 64); }

int launch_content_backward(const float* content, long HW, const float* W, const float* rgb, long rgb_stride, const float* d_rgb, long d_stride,
                            float* workspace, float* d_content, float* dW, float* db, hipStream_t stream) {
  if (HW <= 0) return 0;
  float* d_pre = workspace;                     // [HW,4], column 3 zero
  hipLaunchKernelGGL(content_bwd_kernel, dim3((unsigned)((HW + 3) / 4)), dim3(256), 0, stream, rgb, rgb_stride, d_rgb, d_stride, W, HW, d_pre, d_content);
  wgrad(d_pre, 4, 3, content, 64, 64, dW, 64, db, HW, workspace + (size_t)HW * 4, stream);
  return check_launch("content_backward");
}

}  // namespace crnerf
