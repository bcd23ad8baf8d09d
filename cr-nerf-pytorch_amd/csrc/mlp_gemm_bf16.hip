// Opt-in mixed-precision TRAINING twins of NeRF_sigma.forward (models/nerf.py:157-182): every nn.Linear except static_sigma as one
// points x features GEMM on the bf16 MFMA -- operands (fp32 in HBM, exactly the buffers of the fp32 twins) are rounded to bf16
// (RNE) in registers, products accumulate in fp32, biases / activations / the sigma head / all storage stay fp32.  The forward has
// the semantics of the bf16 inference entry points (include/crnerf.h "bf16 variants"; oracle mlp_forward_bf16); the data gradient
// is the same GEMM on the transposed matrices with the relu mask in the epilogue; the weight gradients are wgrad_kernel's bf16 path.
//
// Why un-fused: with the matrix work 16x cheaper these passes are bound by their activation traffic (1 KiB per point and layer in,
// 1 KiB out), not by the MFMA; a persistent workgroup keeps the layer's weight matrix in LDS as B-operand fragments and streams
// 256-point tiles (64 per wave) through it.
//   C[P x N] = act(A[P x K] . W[N x K]^T + bias)         A = up to two column segments (the skip / dir concatenations)
//   v_mfma_f32_32x32x16_bf16: a-operand lane (i, hh) = A[point i][k = 8 hh + e], b-operand lane (j, hh) = W[feature j][k = 8 hh + e];
//   accumulator register r of lane (j, hh) = C[point (r&3) + 8 (r>>2) + 4 hh][feature j].
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"
#include "layout.h"
#include "mlp_train16.h"

namespace crnerf {

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gb_f32x16 __attribute__((ext_vector_type(16)));

// ---- packed weights: per matrix a stream of 1 KiB fragments, frag(t, s)[lane = 32 hh + j][e] = bf16(M[32 t + j][16 s + 8 hh + e]),
// tile-major (t = feature tile of 32, s = k-step of 16); rows / columns beyond the matrix are zero.  The LDS image is the global image.
struct GemmMat { int off_frag; int N; int K; };   // fragment offset in the packed buffer, output features, padded contraction length
enum {
  GM_L1 = 0, GM_L2, GM_L3, GM_L4, GM_L5, GM_L6, GM_L7, GM_L8, GM_FINAL, GM_DIR, GM_RGB,        // forward: W[N x K]
  GM_T_RGB, GM_T_DIR, GM_T_FINAL, GM_T8, GM_T7, GM_T6, GM_T5, GM_T4, GM_T3, GM_T2,               // data gradient: (W restricted to the hidden inputs)^T
  GM_COUNT
};

constexpr int DIR_SEG_LO = 5;   // the direction embedding x[:, 93:120] is addressed as the 16-byte-aligned x[:, 88:120], first 5 columns masked
__host__ __device__ constexpr int gm_frags(int N, int K) { return ((N + 31) / 32) * (K / 16); }

struct GemmLayout {
  GemmMat m[GM_COUNT];
  int total_frags;
};

static GemmLayout make_layout() {
  GemmLayout L{};
  int off = 0;
  auto add = [&](int id, int N, int K) { L.m[id] = GemmMat{off, N, K}; off += gm_frags(N, K); };
  add(GM_L1, 256, 96);                                  // xyz_encoding_1: [emb 93 -> 96]
  for (int l = GM_L2; l <= GM_L4; ++l) add(l, 256, 256);
  add(GM_L5, 256, 352);                                 // xyz_encoding_5: [emb 96 | h4 256]
  for (int l = GM_L6; l <= GM_L8; ++l) add(l, 256, 256);
  add(GM_FINAL, 256, 256);
  add(GM_DIR, 128, 288);                                // dir_encoding: [final 256 | dir 27 -> 32]
  add(GM_RGB, 64, 128);
  add(GM_T_RGB, 128, 64);                               // d(dir act) = d_rgb . W_rgb
  add(GM_T_DIR, 256, 128);                              // d(final)   = d(dir) . W_dir[:, :256]
  add(GM_T_FINAL, 256, 256);
  for (int l = GM_T8; l <= GM_T2; ++l) add(l, 256, 256);   // xyz_encoding_8 .. 2 (layer 5: its hidden block W[:, 93:])
  L.total_frags = off;
  return L;
}
static const GemmLayout& layout() { static const GemmLayout L = make_layout(); return L; }

size_t gemm_packed_bytes() { return (size_t)layout().total_frags * 1024 + 4096; }   // + fp32 consts: biases are read from the tensors

struct PackJob {
  const float* W; int ld;        // source matrix, row-major
  int rows, cols;                // valid extent of M (after the optional transpose)
  int col0;                      // first source column (forward) / first source column of the block being transposed
  int transpose;                 // M[r][c] = W[c][col0 + r]   (else M[r][c] = W[r][seg(c)])
  int seg0_cols, seg0_pad;       // forward with two segments: columns [0, seg0_cols) come first, padded to seg0_pad
  int seg1_lo;                   // ... and the second segment starts seg1_lo columns into its padded range (see GemmSeg::lo)
  int N, K, off_frag;
};

__global__ __launch_bounds__(256) void gemm_pack_kernel(PackJob j, uint4* __restrict__ packed) {
  const int ks = j.K / 16;
  const int nfrag = ((j.N + 31) / 32) * ks;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < nfrag * 64; idx += gridDim.x * 256) {
    const int frag = idx >> 6, lane = idx & 63, jn = lane & 31, hh = lane >> 5;
    const int t = frag / ks, s = frag - t * ks;
    const int r = 32 * t + jn;
    union { gb_bf16x2 h[4]; uint4 u; } v;
    float e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = 16 * s + 8 * hh + q;
      float val = 0.0f;
      if (r < j.rows) {
        if (j.transpose) {
          if (c < j.cols) val = j.W[(long)c * j.ld + j.col0 + r];
        } else {
          // padded column c -> source column: segment 0 occupies [0, seg0_pad) (valid < seg0_cols), the rest follows
          int src = -1;
          if (c < j.seg0_pad) { if (c < j.seg0_cols) src = c; }
          else { const int c1 = c - j.seg0_pad - j.seg1_lo; if (c1 >= 0 && c1 + j.seg0_cols < j.cols) src = c1 + j.seg0_cols; }
          if (src >= 0) val = j.W[(long)r * j.ld + src];
        }
      }
      e[q] = val;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v.h[q] = gb_bf16x2{(__bf16)e[2 * q], (__bf16)e[2 * q + 1]};
    packed[(long)(j.off_frag + frag) * 64 + lane] = v.u;
  }
}

int launch_pack_mlp_gemm(const MlpTensors& t, void* packed, hipStream_t st) {
  const GemmLayout& L = layout();
  auto fwd = [&](int id, const float* W, int ld, int rows, int cols, int seg0_cols, int seg0_pad, int seg1_lo = 0) {
    PackJob j{W, ld, rows, cols, 0, 0, seg0_cols, seg0_pad, seg1_lo, L.m[id].N, L.m[id].K, L.m[id].off_frag};
    hipLaunchKernelGGL(gemm_pack_kernel, dim3(64), dim3(256), 0, st, j, (uint4*)packed);
  };
  auto tr = [&](int id, const float* W, int ld, int rows, int cols, int col0) {   // M = (W[:, col0 : col0 + rows])^T, M is rows x cols
    PackJob j{W, ld, rows, cols, col0, 1, 0, 0, 0, L.m[id].N, L.m[id].K, L.m[id].off_frag};
    hipLaunchKernelGGL(gemm_pack_kernel, dim3(64), dim3(256), 0, st, j, (uint4*)packed);
  };
  fwd(GM_L1, t.w[0], 93, 256, 93, 93, 96);
  for (int l = 1; l < 8; ++l) {
    if (l == 4) fwd(GM_L5, t.w[4], 349, 256, 349, 93, 96);
    else fwd(GM_L1 + l, t.w[l], 256, 256, 256, 256, 256);
  }
  fwd(GM_FINAL, t.w_final, 256, 256, 256, 256, 256);
  fwd(GM_DIR, t.w_dir, 283, 128, 283, 256, 256, DIR_SEG_LO);
  fwd(GM_RGB, t.w_rgb, 128, 64, 128, 128, 128);
  tr(GM_T_RGB, t.w_rgb, 128, 128, 64, 0);
  tr(GM_T_DIR, t.w_dir, 283, 256, 128, 0);
  tr(GM_T_FINAL, t.w_final, 256, 256, 256, 0);
  for (int l = 7; l >= 1; --l) tr(GM_T8 + (7 - l), t.w[l], l == 4 ? 349 : 256, 256, 256, l == 4 ? 93 : 0);
  return check_launch("gemm_pack_kernel");
}

// ---- the GEMM
// A column segment: `pad` (multiple of 16) padded columns of rows p + row * ld, of which [lo, hi) are real and the rest read as zero.
// p + row * ld is 16-byte aligned and all `pad` columns lie inside the row's allocation (every load is an unconditional float4).
// The direction embedding x[:, 93:120] is not 16-byte aligned: it is addressed as x[:, 88:120] with lo = 5 (DIR_SEG_LO).
struct GemmSeg { const float* p; int ld; int lo; int hi; int pad; };
static_assert((XYZ_DIM - DIR_SEG_LO) % 4 == 0 && XYZ_DIM - DIR_SEG_LO + 32 == IN_DIM, "dir segment = the last 32 columns of the [P,120] input row");
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };
enum { SEG_NONE = 0, SEG_VEC = 1 };                              // second A segment: absent / present
struct GemmJob {
  GemmSeg a0, a1;                 // a0: 16-byte aligned rows, ld % 4 == 0 (vector loads; the tail of its last k-step may read up to 7 floats
                                  // past `cols` inside the row's allocation and zeroes them)
  const uint4* frags;             // fragment stream of the feature tiles of this pass
  int N, K;                       // features of this pass (<= 32 NT), K = a0.pad + a1.pad
  int col_off;                    // first output column / bias index of this pass
  const float* bias;              // indexed by col_off + feature, or null
  int act;
  const uint32_t* bits_in;                    // epilogue: C *= relu' from the activity bits the forward wrote (below), or null
  uint32_t* bits_out;                         // ACT_RELU forward: activity bits of the output, or null.  Layout per point: 32 bytes =
                                              // [hh = 0,1][tile t = 0..7] u16, bit 4 q + e <-> feature 32 t + 8 q + 4 hh + e: exactly what
                                              // lane (point, hh) produces / consumes, so 16 bytes per lane and tile of 64 points
                                              // (the activation itself would be another 1 KiB per point and layer of HBM reads)
  const float* r1_row; const float* r1_col;   // epilogue: C += r1_row[row] * r1_col[col]   (the sigma head's branch into d(h8)) or null
  float* out; int ldo;
  long P;
  int dbg;                        // timing experiments only (CRNERF_GEMM_DBG): 1 = no epilogue, 2 = no operand loads, 4 = no MFMAs
};

constexpr int GEMM_LDS_BYTES = 128 * 1024 + 2048;   // fragments + bias / rank-1 column vector
constexpr int GEMM_MT = 1;        // 32-point tiles per wave: 1 keeps the kernel under 256 registers -> two waves per SIMD, so one wave's
                                  // store phase overlaps another's load phase (MT = 2 at one wave per SIMD: 8.4 ms per 2^20-point forward)
constexpr int GEMM_WAVES = 8;
constexpr int GEMM_TILE = 32 * GEMM_MT;
constexpr int GEMM_PF = 4;        // k-steps per prefetch chunk: 4 x 4 KiB per wave in flight (~16 MB on the chip)

__device__ __forceinline__ void gemm_load_a(const float* p, int ld, int lo, int hi, const long (&rws)[GEMM_MT], int cl, float (&v)[GEMM_MT][8]) {
  // UNCONDITIONAL loads (a branch around a load costs a vmcnt(0) at the join and serialises the prefetch); padding columns are read
  // from inside the row and zeroed by a select
#pragma unroll
  for (int m = 0; m < GEMM_MT; ++m) {
    const float* rp = p + rws[m] * ld;
    const float4 x0 = *(const float4*)(rp + cl), x1 = *(const float4*)(rp + cl + 4);
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[m][e] = (cl + e >= lo && cl + e < hi) ? x[e] : 0.0f;
  }
}

// NT feature tiles of 32 per pass: 8 (N = 256), 4 (N = 128), 2 (N = 64).  ACT / MASK (relu' bits in) / R1 (rank-1 term) are compile-time:
// as run-time switches they became ~1,400 branches and 200 spilled registers in the epilogue, and every store sat behind a spill
// reload's s_waitcnt vmcnt(0) -- i.e. behind the previous store's completion.
template <int NT, int A1KIND, int ACT, bool MASK, bool R1>
__global__ __launch_bounds__(64 * GEMM_WAVES, GEMM_WAVES / 4) void linear_bf16_kernel(GemmJob j) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hh = lane >> 5;
  const int ks = j.K / 16, ks0 = j.a0.pad / 16;
  {   // stage the weight fragments: the LDS image is the global image
    const int n16 = NT * ks * 64;
    uint4* dst = (uint4*)gsm;
    for (int idx = tid; idx < n16; idx += 64 * GEMM_WAVES) dst[idx] = j.frags[idx];
    float* eb = (float*)(gsm + (size_t)NT * ks * 1024);   // behind the fragments: [0, 256) bias of this pass' features, [256, 512) rank-1 column vector
    if (tid < 32 * NT) {
      const bool ok = tid < j.N;
      eb[tid] = (j.bias && ok) ? j.bias[j.col_off + tid] : 0.0f;
      eb[256 + tid] = (j.r1_col && ok) ? j.r1_col[j.col_off + tid] : 0.0f;
    }
  }
  __syncthreads();
  const float* eb = (const float*)(gsm + (size_t)NT * ks * 1024);
  const long tiles = (j.P + GEMM_TILE - 1) / GEMM_TILE;
  const int chunks = (ks + GEMM_PF - 1) / GEMM_PF;
  const long tstride = (long)gridDim.x * GEMM_WAVES;
  // operand staging lives ACROSS tiles: the first chunk of the next tile is requested in the last k-chunk of the current one, so loads
  // are in flight during the epilogue's store phase too
  float nxt[GEMM_PF][GEMM_MT][8];
#pragma unroll
  for (int u = 0; u < GEMM_PF; ++u)
#pragma unroll
    for (int m = 0; m < GEMM_MT; ++m)
#pragma unroll
      for (int e = 0; e < 8; ++e) nxt[u][m][e] = 1.0f;
  auto rows_of = [&](long tile, long (&rw)[GEMM_MT]) {
#pragma unroll
    for (int m = 0; m < GEMM_MT; ++m) { const long rr = tile * GEMM_TILE + 32 * m + i; rw[m] = rr < j.P ? rr : j.P - 1; }
  };
  // k-step s reads columns 16 s + 8 hh .. +7 of the segmented input row; steps past the end re-read the last step (unused)
  auto fetch = [&](int c, const long (&rw)[GEMM_MT], float (&buf)[GEMM_PF][GEMM_MT][8]) {
#pragma unroll
    for (int u = 0; u < GEMM_PF; ++u) {
      const int s = c * GEMM_PF + u < ks ? c * GEMM_PF + u : ks - 1;
      const bool first = A1KIND == SEG_NONE || s < ks0;            // segment by selects: one load path, no branch
      gemm_load_a(first ? j.a0.p : j.a1.p, first ? j.a0.ld : j.a1.ld, first ? j.a0.lo : j.a1.lo, first ? j.a0.hi : j.a1.hi, rw,
                  16 * (first ? s : s - ks0) + 8 * hh, buf[u]);
    }
  };
  {
    const long t0 = (long)blockIdx.x * GEMM_WAVES + wave;
    long rw0[GEMM_MT];
    rows_of(t0 < tiles ? t0 : 0, rw0);
    if (!(j.dbg & 2)) fetch(0, rw0, nxt);
  }
  for (long tile = (long)blockIdx.x * GEMM_WAVES + wave; tile < tiles; tile += tstride) {
    const long row0 = tile * GEMM_TILE;
    gb_f32x16 acc[GEMM_MT][NT];
#pragma unroll
    for (int m = 0; m < GEMM_MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;
    long rws[GEMM_MT];   // rows of this lane's point tiles (clamped: always readable)
    rows_of(tile, rws);
    uint4 bin[GEMM_MT];
#pragma unroll
    for (int m = 0; m < GEMM_MT; ++m) bin[m] = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (MASK) {
#pragma unroll
      for (int m = 0; m < GEMM_MT; ++m) bin[m] = *(const uint4*)(j.bits_in + (rws[m] * 2 + hh) * 4);
    }
    long rws_next[GEMM_MT];                                // the next tile of this wave (clamped to a valid tile when there is none)
    rows_of(tile + tstride < tiles ? tile + tstride : tile, rws_next);
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
      gb_bf16x8 curb[GEMM_PF][GEMM_MT];                    // this chunk's operands, rounded (v_cvt_pk_bf16_f32); frees the fp32 staging
#pragma unroll
      for (int u = 0; u < GEMM_PF; ++u)
#pragma unroll
        for (int m = 0; m < GEMM_MT; ++m) {
          union { gb_bf16x2 h[4]; gb_bf16x8 v8; } cv;
#pragma unroll
          for (int q = 0; q < 4; ++q) cv.h[q] = gb_bf16x2{(__bf16)nxt[u][m][2 * q], (__bf16)nxt[u][m][2 * q + 1]};
          curb[u][m] = cv.v8;
        }
      if (!(j.dbg & 2)) {                                  // the next chunk's operands fly while this one's MFMAs run; after the last chunk,
        const bool last = c + 1 == chunks;                   // the NEXT TILE's first chunk (rows by select: one load path, no branch)
        long rsel[GEMM_MT];
#pragma unroll
        for (int m = 0; m < GEMM_MT; ++m) rsel[m] = last ? rws_next[m] : rws[m];
        fetch(last ? 0 : c + 1, rsel, nxt);
      }
#pragma unroll
      for (int u = 0; u < GEMM_PF; ++u) {
        const int s = c * GEMM_PF + u;
        if (s < ks && !(j.dbg & 4)) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const gb_bf16x8 bfr = *(const gb_bf16x8*)(gsm + ((size_t)(t * ks + s) * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < GEMM_MT; ++m)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, curb[u][m], acc[m][t], 0, 0, 0);   // swapped: D[feature][point]
          }
        }
      }
    }
    // epilogue.  Swapped operands leave lane (p, hh) with point row0 + 32 m + p and, in registers 4 q .. 4 q + 3 of tile t, the four
    // consecutive features 32 t + 8 q + 4 hh + 0..3: one 16-byte store per quad (un-swapped, a lane holds ONE feature of 16 points:
    // 256 dword stores per tile and wave).
    if (j.dbg & 1) continue;
    const bool vec_out = (j.ldo & 3) == 0;
    const int t_off = j.col_off >> 5;
#pragma unroll
    for (int m = 0; m < GEMM_MT; ++m) {
      const long row = row0 + 32 * m + i;
      const bool row_ok = row < j.P;
      float* orow = j.out + rws[m] * j.ldo + j.col_off + 4 * hh;
      const float r1 = R1 ? j.r1_row[rws[m]] : 0.0f;
      const uint32_t bw_in[4] = {bin[m].x, bin[m].y, bin[m].z, bin[m].w};
      uint32_t bw_out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tg = t_off + t;                                 // tile index within the point's 256 features
        const uint32_t half_in = MASK ? bw_in[(tg >> 1) & 3] >> ((tg & 1) * 16) : 0xffffu;
        uint32_t half_out = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int fcol = 32 * t + 8 * q + 4 * hh;               // < 32 NT; features beyond N (multiple of 32 here) do not exist
          const float4 b = *(const float4*)(eb + fcol);
          float v[4] = {acc[m][t][4 * q + 0] + b.x, acc[m][t][4 * q + 1] + b.y, acc[m][t][4 * q + 2] + b.z, acc[m][t][4 * q + 3] + b.w};
          if (R1) {
            const float4 rc = *(const float4*)(eb + 256 + fcol);
            v[0] = fmaf(r1, rc.x, v[0]); v[1] = fmaf(r1, rc.y, v[1]); v[2] = fmaf(r1, rc.z, v[2]); v[3] = fmaf(r1, rc.w, v[3]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (ACT == ACT_RELU) { v[e] = fmaxf(v[e], 0.0f); half_out |= (v[e] > 0.0f ? 1u : 0u) << (4 * q + e); }
            else if (ACT == ACT_SIGMOID) v[e] = sigmoid_ref(v[e]);
            if (MASK) v[e] = ((half_in >> (4 * q + e)) & 1u) ? v[e] : 0.0f;
          }
          if (row_ok) {
            float* o = orow + 32 * t + 8 * q;
            if (vec_out) *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
          }
        }
        bw_out[(tg >> 1) & 3] |= half_out << ((tg & 1) * 16);
      }
      if (ACT == ACT_RELU && j.bits_out && row_ok) {              // the words this pass covers: 8 tiles = 16 bytes, 4 tiles = 8 bytes
        uint32_t* bo = j.bits_out + (row * 2 + hh) * 4;
        if (NT == 8) *(uint4*)bo = make_uint4(bw_out[0], bw_out[1], bw_out[2], bw_out[3]);
        else if (NT == 4) { const int w0 = (t_off >> 1) & 3; *(uint2*)(bo + w0) = make_uint2(bw_out[w0], bw_out[w0 + 1]); }
      }
    }
  }
}

static int run_gemm(const GemmJob& j_in, int a1kind, hipStream_t st) {
  if (j_in.P <= 0) return 0;
  static const int dbg = getenv("CRNERF_GEMM_DBG") ? atoi(getenv("CRNERF_GEMM_DBG")) : 0;
  GemmJob j = j_in;
  j.dbg = dbg;
  const int nt = (j.N + 31) / 32;
  const long tiles = (j.P + GEMM_TILE - 1) / GEMM_TILE;
  const int cus = num_cus();
  const long wg = (tiles + GEMM_WAVES - 1) / GEMM_WAVES;
  const int grid = (int)(wg < cus ? wg : cus);
  const size_t shmem = (size_t)nt * (j.K / 16) * 1024 + 2048;
  if (shmem > GEMM_LDS_BYTES) return set_error(-2, "linear_bf16: weight tile exceeds LDS");
  const bool mask = j.bits_in != nullptr, r1 = j.r1_row != nullptr;
  if (j.N != 32 * nt) return set_error(-2, "linear_bf16: feature count must be a multiple of 32");
#define CRNERF_GEMM(NTV, KIND, ACTV, MASKV, R1V)                                                                                            \
  if (nt == NTV && a1kind == KIND && j.act == ACTV && mask == MASKV && r1 == R1V) {                                                         \
    if (int rc = ensure_dynamic_lds((const void*)linear_bf16_kernel<NTV, KIND, ACTV, MASKV, R1V>, GEMM_LDS_BYTES, "linear_bf16_kernel")) return rc; \
    hipLaunchKernelGGL((linear_bf16_kernel<NTV, KIND, ACTV, MASKV, R1V>), dim3(grid), dim3(64 * GEMM_WAVES), shmem, st, j);                  \
    return 0;                                                                                                                               \
  }
  CRNERF_GEMM(8, SEG_NONE, ACT_RELU, false, false)       // xyz_encoding_1..4, 6..8
  CRNERF_GEMM(4, SEG_VEC, ACT_RELU, false, false)        // xyz_encoding_5 (two passes), dir_encoding
  CRNERF_GEMM(8, SEG_NONE, ACT_NONE, false, false)       // xyz_encoding_final; d(final)
  CRNERF_GEMM(2, SEG_NONE, ACT_SIGMOID, false, false)    // static_rgb
  CRNERF_GEMM(4, SEG_NONE, ACT_NONE, true, false)        // d(dir act)
  CRNERF_GEMM(8, SEG_NONE, ACT_NONE, true, true)         // d(h8): + the sigma head's branch
  CRNERF_GEMM(8, SEG_NONE, ACT_NONE, true, false)        // d(h7..h1)
#undef CRNERF_GEMM
  return set_error(-2, "linear_bf16: unsupported shape");
}

// sigma = softplus(w_sigma . h8 + b) on the un-rounded fp32 activations (models/nerf.py:146,172); one point per lane group of 16
__global__ __launch_bounds__(256) void sigma_head_kernel(const float* __restrict__ h8, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ out, long P) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const long p = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;
  if (p >= P) return;
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 x = *(const float4*)(h8 + p * ACT_W + 64 * q + 4 * sub), ww = *(const float4*)(w + 64 * q + 4 * sub);
    s = fmaf(x.x, ww.x, s); s = fmaf(x.y, ww.y, s); s = fmaf(x.z, ww.z, s); s = fmaf(x.w, ww.w, s);
  }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 16);
  if (sub == 0) out[p * OUT_DIM + FEAT_DIM] = softplus_ref(s + b[0]);
}

// d_rgb_pre = d_out[:, :64] * f (1 - f);  d_sig_pre = d_out[:, 64] * (1 - exp(-sigma))    (sigmoid', softplus')
__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ out, const float* __restrict__ d_out, float* __restrict__ d_rgb,
                                                        float* __restrict__ d_sig, long P) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * OUT_DIM) return;
  const long p = idx / OUT_DIM;
  const int c = (int)(idx - p * OUT_DIM);
  const float f = out[idx], g = d_out[idx];
  if (c < FEAT_DIM) d_rgb[p * FEAT_DIM + c] = g * f * (1.0f - f);
  else d_sig[p] = g * (1.0f - expf(-f));
}

static const uint4* frag_ptr(const void* packed, int id) { return (const uint4*)packed + (size_t)layout().m[id].off_frag * 64; }

int launch_mlp_forward_train_mixed(const MlpTensors& t, const void* packed, const float* x, float* out, float* acts, long P, hipStream_t st) {
  if (P <= 0) return 0;
  const GemmLayout& L = layout();
  auto A = [&](int slot) { return acts + (size_t)slot * P * ACT_W; };
  const GemmSeg none{nullptr, 0, 0, 0, 0};
  // tiles [t0, t0 + nt) of matrix `id` as one pass (a 256 x 352 matrix does not fit the 128 KiB of LDS: the skip layer runs as two
  // 128-feature passes and reads its input twice)
  auto bits = [&](int slot) { return (uint32_t*)(acts + (size_t)ACT_SLOTS * P * ACT_W) + (size_t)slot * P * 8; };   // 32 bytes per point and slot
  auto gemm = [&](int id, int t0, int nt, GemmSeg a0, GemmSeg a1, int a1kind, const float* bias, int act, float* o, int ldo, uint32_t* bo) {
    const int ks = L.m[id].K / 16;
    const int n = L.m[id].N - 32 * t0 < 32 * nt ? L.m[id].N - 32 * t0 : 32 * nt;
    GemmJob j{a0, a1, frag_ptr(packed, id) + (size_t)t0 * ks * 64, n, L.m[id].K, 32 * t0, bias, act, nullptr, bo, nullptr, nullptr, o, ldo, P, 0};
    return run_gemm(j, a1kind, st);
  };
  const GemmSeg emb{x, IN_DIM, 0, XYZ_DIM, 96};
  if (int rc = gemm(GM_L1, 0, 8, emb, none, SEG_NONE, t.b[0], ACT_RELU, A(0), ACT_W, bits(0))) return rc;
  for (int l = 1; l < 8; ++l) {
    const GemmSeg h{A(l - 1), ACT_W, 0, 256, 256};
    if (l == 4) {
      if (int rc = gemm(GM_L5, 0, 4, emb, h, SEG_VEC, t.b[4], ACT_RELU, A(4), ACT_W, bits(4))) return rc;
      if (int rc = gemm(GM_L5, 4, 4, emb, h, SEG_VEC, t.b[4], ACT_RELU, A(4), ACT_W, bits(4))) return rc;
    } else if (int rc = gemm(GM_L1 + l, 0, 8, h, none, SEG_NONE, t.b[l], ACT_RELU, A(l), ACT_W, bits(l))) return rc;
  }
  hipLaunchKernelGGL(sigma_head_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, st, A(7), t.w_sigma, t.b_sigma, out, P);
  if (int rc = gemm(GM_FINAL, 0, 8, GemmSeg{A(7), ACT_W, 0, 256, 256}, none, SEG_NONE, t.b_final, ACT_NONE, A(8), ACT_W, nullptr)) return rc;
  if (int rc = gemm(GM_DIR, 0, 4, GemmSeg{A(8), ACT_W, 0, 256, 256}, GemmSeg{x + XYZ_DIM - DIR_SEG_LO, IN_DIM, DIR_SEG_LO, 32, 32}, SEG_VEC, t.b_dir, ACT_RELU, A(9), ACT_W,
                    bits(9)))
    return rc;
  if (int rc = gemm(GM_RGB, 0, 2, GemmSeg{A(9), ACT_W, 0, 128, 128}, none, SEG_NONE, t.b_rgb, ACT_SIGMOID, out, OUT_DIM, nullptr)) return rc;
  return check_launch("mlp_forward_train_mixed");
}

int launch_mlp_backward_mixed(const MlpTensors& t, const void* packed, const float* x, const float* out, const float* d_out, const float* acts,
                              void* scratch, float* const* grads, long P, hipStream_t st) {
  if (P <= 0) return 0;
  const GemmLayout& L = layout();
  float* deltas = (float*)scratch;
  float* d_rgb = deltas + (size_t)ACT_SLOTS * P * ACT_W;
  float* d_sig = d_rgb + (size_t)P * FEAT_DIM;
  float* ws = d_sig + P;
  auto D = [&](int slot) { return deltas + (size_t)slot * P * ACT_W; };
  hipLaunchKernelGGL(head_grad_kernel, dim3((unsigned)((P * OUT_DIM + 255) / 256)), dim3(256), 0, st, out, d_out, d_rgb, d_sig, P);
  const GemmSeg none{nullptr, 0, 0, 0, 0};
  auto bits = [&](int slot) { return (const uint32_t*)(acts + (size_t)ACT_SLOTS * P * ACT_W) + (size_t)slot * P * 8; };
  auto dg = [&](int id, GemmSeg a0, const uint32_t* mask, const float* r1r, const float* r1c, float* o) {
    GemmJob j{a0, none, frag_ptr(packed, id), L.m[id].N, L.m[id].K, 0, nullptr, ACT_NONE, mask, nullptr, r1r, r1c, o, ACT_W, P, 0};
    return run_gemm(j, SEG_NONE, st);
  };
  if (int rc = dg(GM_T_RGB, GemmSeg{d_rgb, FEAT_DIM, 0, 64, 64}, bits(9), nullptr, nullptr, D(9))) return rc;        // through static_rgb, relu' of dir act
  if (int rc = dg(GM_T_DIR, GemmSeg{D(9), ACT_W, 0, 128, 128}, nullptr, nullptr, nullptr, D(8))) return rc;        // through dir_encoding[:, :256] (final is linear)
  if (int rc = dg(GM_T_FINAL, GemmSeg{D(8), ACT_W, 0, 256, 256}, bits(7), d_sig, t.w_sigma, D(7))) return rc;          // through final + the sigma head, relu' of h8
  for (int l = 7; l >= 1; --l)                                                                                    // through xyz_encoding_{l+1}, relu' of h_l
    if (int rc = dg(GM_T8 + (7 - l), GemmSeg{D(l), ACT_W, 0, 256, 256}, bits(l - 1), nullptr, nullptr, D(l - 1))) return rc;
  if (int rc = check_launch("mlp_backward_mixed")) return rc;
  return launch_mlp_wgrads(x, acts, deltas, d_rgb, d_sig, ws, grads, P, st, 1);
}

}  // namespace crnerf
