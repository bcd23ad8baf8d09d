// Opt-in mixed-precision TRAINING twins of NeRF_sigma.forward (models/nerf.py:157-182): every nn.Linear except static_sigma as one
// points x features GEMM on the bf16 MFMA.  Operands are bf16 (RNE), products accumulate in fp32, biases / activations / the sigma
// head are evaluated in fp32 on the accumulators -- and what travels through HBM between the layers is bf16: the saved
// activations, the layer deltas of the backward, the embedded input.  (Rounding at the store instead of at the next load changes
// nothing for the GEMMs -- the operand was rounded anyway -- and halves the traffic these passes are bound by; round 2's first
// version kept fp32 buffers and rounded in registers: 1 KiB in + 1 KiB out per point and layer.)
// The forward has the semantics of the bf16 inference entry points (include/crnerf.h "bf16 variants"; oracle mlp_forward_bf16);
// the data gradient is the same GEMM on the transposed matrices with the relu mask in the epilogue; the weight gradients multiply
// the stored bf16 deltas and activations (wgrad_b_kernel below).  oracle/cpu_ref.py mlp_forward_bf16_train restates all three.
//
// Why un-fused: with the matrix work 16x cheaper these passes are bound by their activation traffic, not by the MFMA; a persistent
// workgroup keeps the layer's weight matrix in LDS as MFMA operand fragments and streams 32-point tiles (one per wave) through it.
//   C[P x N] = act(A[P x K] . W[N x K]^T + bias)         A = up to two column segments (the skip / dir concatenations)
//   v_mfma_f32_16x16x32_bf16, swapped operands: a-operand lane (i, kq) = W[feature i][k = 8 kq + e], b-operand lane (j, kq) =
//   A[point j][k = 8 kq + e] (i, j = lane & 15, kq = lane >> 4); accumulator register r of lane (j, q4) = C[point j][feature 4 q4 + r].
//   Why the 16-wide shape: a load or store instruction moves 16 bytes per lane, and FOUR lanes share a point here -- 64 contiguous
//   bytes per row and instruction, 16 rows -- against 32 bytes x 32 rows with the 32x32x16 shape (two lanes per point).  A row copy in
//   the two patterns (tools/ubench/row_patterns.hip): reads 6.2 / 6.3 TB/s, writes 4.5 / 3.8, read + write 5.0 / 4.4.
//
// Storage order of an activation / delta row (256 bf16 = 512 B): within every group of 32 features, feature 16 b + 4 q4 + r sits at
// position 8 q4 + 4 b + r -- the 8 values lane (j, q4) holds of a group (rows 4 q4 .. 4 q4 + 3 of its two 16-feature tiles) are ONE
// 16-byte piece, and the four lanes of a point write 64 contiguous bytes per store instruction.  Consumers never see the
// permutation: the packed weight fragments carry it in their k index, the weight-gradient kernel un-permutes when it writes dW.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"
#include "layout.h"
#include "mlp_train16.h"

namespace crnerf {

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gb_f32x16 __attribute__((ext_vector_type(16)));
typedef float gb_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;   // raw storage

// feature 16 b + 4 q4 + r of a group of 32  ->  position 8 q4 + 4 b + r, and back (b: which of the group's two 16-feature MFMA tiles,
// q4 = lane >> 4: the lane quarter that holds accumulator rows 4 q4 .. 4 q4 + 3)
__host__ __device__ constexpr int perm32(int f) { return (f & ~31) | (((f >> 2) & 3) << 3) | (((f >> 4) & 1) << 2) | (f & 3); }
__host__ __device__ constexpr int unperm32(int c) { return (c & ~31) | (((c >> 2) & 1) << 4) | (((c >> 3) & 3) << 2) | (c & 3); }
// "fused" storage order of the rows the bf16 pair renderer's training twin writes (mlp_core_bf16p.h): feature 32T + 8c + 4h + i sits at
// position 32T + 16(c>>1) + 8h + 4(c&1) + i, i.e. bits 2 and 3 of the index trade places (an involution that keeps groups of four together)
__host__ __device__ constexpr int perm_fused(int f) { return (f & ~12) | (((f >> 2) & 1) << 3) | (((f >> 3) & 1) << 2); }
static_assert(perm_fused(perm_fused(91)) == 91 && perm_fused(32 * 3 + 8 * 2 + 4 * 1 + 3) == 32 * 3 + 16 * 1 + 8 * 1 + 4 * 0 + 3 &&
              perm_fused(8 * 1 + 4 * 0 + 2) == 4 + 2, "fused storage permutation");
// column orders of a weight-gradient operand: where column c of the stored row lives in the reference's feature order (-1: padding)
enum { ORD_REF = 0, ORD_PERM32 = 1, ORD_FUSED = 2, ORD_XYZ_SLOTS = 3, ORD_DIR_SLOTS = 4 };
__host__ __device__ inline int ord_to_ref(int ord, int c) {
  return ord == ORD_PERM32 ? unperm32(c) : ord == ORD_FUSED ? perm_fused(c) : ord == ORD_XYZ_SLOTS ? posenc_slot_to_col_b(c, XYZ_FREQS)
       : ord == ORD_DIR_SLOTS ? posenc_slot_to_col_b(c, DIR_FREQS) : c;
}
static_assert(unperm32(perm32(77)) == 77 && perm32(unperm32(200)) == 200 && perm32(16 * 1 + 4 * 2 + 3) == 8 * 2 + 4 + 3 && perm32(16 * 0 + 4 * 3 + 1) == 24 + 1,
              "storage permutation");

constexpr int XB_W = 128;        // embedded input as bf16: [0, 93) xyz embedding, [93, 96) zero, [96, 123) direction embedding, [123, 128) zero
constexpr int XB_DIR = 96;
constexpr int DRGB_W = 64;

__device__ __forceinline__ uint32_t gb_pk(float a, float b) {
  const gb_bf16x2 v = {(__bf16)a, (__bf16)b};   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float gb_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float gb_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- packed weights: per matrix a stream of 1 KiB fragments, frag(t, s)[lane = 16 kq + i][e] = bf16(M[16 t + i][32 s + 8 kq + e]),
// tile-major (t = feature tile of 16, s = k-step of 32); rows / columns beyond the matrix are zero.  The LDS image is the global image.
struct GemmMat { int off_frag; int N; int K; };   // fragment offset in the packed buffer, output features, padded contraction length
enum {
  GM_L1 = 0, GM_L2, GM_L3, GM_L4, GM_L5, GM_L6, GM_L7, GM_L8, GM_FINAL, GM_DIR, GM_RGB,        // forward: W[N x K]
  GM_T_RGB, GM_T_DIR, GM_T_FINAL, GM_T8, GM_T7, GM_T6, GM_T5, GM_T4, GM_T3, GM_T2,               // data gradient: (W restricted to the hidden inputs)^T
  GM_COUNT
};

__host__ __device__ constexpr int gm_frags(int N, int K) { return ((N + 15) / 16) * (K / 32); }

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

size_t gemm_packed_bytes() { return (size_t)layout().total_frags * 1024 + 4096; }

struct PackJob {
  const float* W; int ld;        // source matrix, row-major
  int rows, cols;                // forward: output features / source columns; transpose: rows of M / rows of the source block
  int col0;                      // transpose: first source column of the block
  int transpose;                 // M[r][c] = W[k(c)][col0 + r]   (else M[r][c] = W[r][src(c)])
  int seg0_cols, seg0_pad;       // forward: source columns [0, seg0_cols) occupy the padded range [0, seg0_pad), the rest follows
  int perm0, perm1;              // the k range of segment 0 / 1 indexes a row in storage order (an activation or delta slot), not reference order
  int N, K, off_frag;
};

__global__ __launch_bounds__(256) void gemm_pack_kernel(PackJob j, uint4* __restrict__ packed) {
  const int ks = j.K / 32;
  const int nfrag = ((j.N + 15) / 16) * ks;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < nfrag * 64; idx += gridDim.x * 256) {
    const int frag = idx >> 6, lane = idx & 63, jn = lane & 15, kq = lane >> 4;
    const int t = frag / ks, s = frag - t * ks;
    const int r = 16 * t + jn;
    float e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = 32 * s + 8 * kq + q;
      float val = 0.0f;
      if (r < j.rows) {
        if (j.transpose) {
          const int cc = j.perm0 ? unperm32(c) : c;
          if (cc < j.cols) val = j.W[(long)cc * j.ld + j.col0 + r];
        } else {
          int src = -1;
          if (c < j.seg0_pad) { const int cc = j.perm0 ? unperm32(c) : c; if (cc < j.seg0_cols) src = cc; }
          else { const int c1 = c - j.seg0_pad, cc = j.perm1 ? unperm32(c1) : c1; if (cc + j.seg0_cols < j.cols) src = cc + j.seg0_cols; }
          if (src >= 0) val = j.W[(long)r * j.ld + src];
        }
      }
      e[q] = val;
    }
    packed[(long)(j.off_frag + frag) * 64 + lane] = make_uint4(gb_pk(e[0], e[1]), gb_pk(e[2], e[3]), gb_pk(e[4], e[5]), gb_pk(e[6], e[7]));
  }
}

int launch_pack_mlp_gemm(const MlpTensors& t, void* packed, hipStream_t st) {
  const GemmLayout& L = layout();
  auto fwd = [&](int id, const float* W, int ld, int rows, int cols, int seg0_cols, int seg0_pad, int perm0, int perm1) {
    PackJob j{W, ld, rows, cols, 0, 0, seg0_cols, seg0_pad, perm0, perm1, L.m[id].N, L.m[id].K, L.m[id].off_frag};
    hipLaunchKernelGGL(gemm_pack_kernel, dim3(64), dim3(256), 0, st, j, (uint4*)packed);
  };
  auto tr = [&](int id, const float* W, int ld, int rows, int cols, int col0, int permk) {   // M = (W[:, col0 : col0 + rows])^T, M is rows x cols
    PackJob j{W, ld, rows, cols, col0, 1, 0, 0, permk, 0, L.m[id].N, L.m[id].K, L.m[id].off_frag};
    hipLaunchKernelGGL(gemm_pack_kernel, dim3(64), dim3(256), 0, st, j, (uint4*)packed);
  };
  fwd(GM_L1, t.w[0], 93, 256, 93, 93, 96, 0, 0);
  for (int l = 1; l < 8; ++l) {
    if (l == 4) fwd(GM_L5, t.w[4], 349, 256, 349, 93, 96, 0, 1);          // [embedding | h4 (storage order)]
    else fwd(GM_L1 + l, t.w[l], 256, 256, 256, 256, 256, 1, 0);
  }
  fwd(GM_FINAL, t.w_final, 256, 256, 256, 256, 256, 1, 0);
  fwd(GM_DIR, t.w_dir, 283, 128, 283, 256, 256, 1, 0);                    // [final (storage order) | direction embedding]
  fwd(GM_RGB, t.w_rgb, 128, 64, 128, 128, 128, 1, 0);
  tr(GM_T_RGB, t.w_rgb, 128, 128, 64, 0, 0);                              // k = d_rgb, reference order
  tr(GM_T_DIR, t.w_dir, 283, 256, 128, 0, 1);
  tr(GM_T_FINAL, t.w_final, 256, 256, 256, 0, 1);
  for (int l = 7; l >= 1; --l) tr(GM_T8 + (7 - l), t.w[l], l == 4 ? 349 : 256, 256, 256, l == 4 ? 93 : 0, 1);
  return check_launch("gemm_pack_kernel");
}

// ---- embedded input x[P,120] fp32 -> xb[P,128] bf16 (16-byte units; both embeddings start on a 16-byte boundary and are zero-padded)
__global__ __launch_bounds__(256) void embed_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb, long P) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * (XB_W / 8)) return;
  const long p = idx >> 4;
  const int u = (int)(idx & 15);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * u + e;
    const int src = c < XYZ_DIM ? c : (c >= XB_DIR && c < XB_DIR + DIR_DIM ? c - (XB_DIR - XYZ_DIM) : -1);
    v[e] = src >= 0 ? x[p * IN_DIM + src] : 0.0f;
  }
  *(uint4*)(xb + p * XB_W + 8 * u) = make_uint4(gb_pk(v[0], v[1]), gb_pk(v[2], v[3]), gb_pk(v[4], v[5]), gb_pk(v[6], v[7]));
}

// ---- the GEMM
// A column segment: `pad` (multiple of 32) bf16 columns of rows p + row * ld, all physically present (padding columns hold zeros,
// or meet zero weight columns), 16-byte aligned.
struct GemmSeg { const bf16_t* p; int ld; int pad; };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };
struct GemmJob {
  GemmSeg a0, a1;
  const uint4* frags;             // fragment stream of the feature tiles of this pass
  int N, K;                       // features of this pass (<= 32 NT), K = a0.pad + a1.pad
  int col_off;                    // first output feature / bias index of this pass (multiple of 32)
  const float* bias;              // indexed by col_off + feature, or null
  int act;
  const uint32_t* bits_in;        // epilogue: C *= relu' from the activity bits the forward wrote (below), or null
  uint32_t* bits_out;             // ACT_RELU forward: activity bits of the output, or null.  Layout per point: 32 bytes =
                                  // [q4 = 0..3][group u = 0..7] u8, bit 4 b + r <-> feature 32 u + 16 b + 4 q4 + r: exactly what
                                  // lane (point, q4) produces / consumes, 8 bytes per lane
  const float* r1_row;            // epilogue: C += r1_row[row] * colvec[feature]   (the sigma head's branch into d(h8)) or null
  const float* colvec;            // [features]: the rank-1 column vector, or (SIG) static_sigma.weight
  const float* sig_b; float* sig_out;   // SIG: sig_out[row] = softplus(colvec . relu(C[row]) + sig_b[0]) on the fp32 accumulators -- a compact [P]
                                        // array (sixteen lanes write 64 contiguous bytes); written straight into column 64 of the [P,65] result,
                                        // one 4-byte piece per 260-byte row, the store cost 60 us per 2^20 points
  const float* sig_in;                  // ACT_SIGMOID: copied into column 64 of the fp32 rows this pass writes (the rows are then complete)
  bf16_t* out; int ldo;           // bf16 output rows in storage order (ACT_NONE / ACT_RELU)
  float* out_f; int ldo_f;        // fp32 output rows in reference order (ACT_SIGMOID: the [P,65] result)
  long P;
  int dbg;                        // timing experiments only (CRNERF_GEMM_DBG): 1 = no epilogue, 2 = no operand loads, 4 = no MFMAs
};

constexpr int GEMM_LDS_BYTES = 128 * 1024 + 2048;   // fragments + bias / column vector
constexpr int GEMM_WAVES = 8;     // two per SIMD (< 256 registers): one wave's store phase overlaps another's load phase
constexpr int GEMM_TILE = 32;     // points per wave tile: two groups of 16 (each weight fragment read feeds two MFMAs)
constexpr int GEMM_PF = 4;        // k-steps (of 32) per prefetch chunk: 8 x 16 B per lane = 8 KiB per wave in flight (~16 MB on the chip)

// NT groups of 32 features per pass: 8 (N = 256), 4 (N = 128), 2 (N = 64).  SEG2 / ACT / MASK (relu' bits in) / R1 (rank-1 term) / SIG
// (fused sigma head) are compile-time: as run-time switches they became ~1,400 branches and 200 spilled registers in the epilogue,
// and every store sat behind a spill reload's s_waitcnt vmcnt(0) -- i.e. behind the previous store's completion.
template <int NT, bool SEG2, int ACT, bool MASK, bool R1, bool SIG>
__global__ __launch_bounds__(64 * GEMM_WAVES, GEMM_WAVES / 4) void linear_bf16_kernel(GemmJob j) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jp = lane & 15, q4 = lane >> 4;       // b-operand: point jp, k quarter q4; accumulator: point jp, feature rows 4 q4 .. 4 q4 + 3
  const int ks = j.K / 32, ks0 = j.a0.pad / 32;
  constexpr int PF = GEMM_PF;
  constexpr int NT16 = 2 * NT;
  {   // stage the weight fragments: the LDS image is the global image
    const int n16 = NT16 * ks * 64;
    uint4* dst = (uint4*)gsm;
    for (int idx = tid; idx < n16; idx += 64 * GEMM_WAVES) dst[idx] = j.frags[idx];
    float* eb = (float*)(gsm + (size_t)NT16 * ks * 1024);   // behind the fragments: [0, 256) bias of this pass' features, [256, 512) column vector
    if (tid < 32 * NT) {
      const bool ok = tid < j.N;
      eb[tid] = (j.bias && ok) ? j.bias[j.col_off + tid] : 0.0f;
      eb[256 + tid] = (j.colvec && ok) ? j.colvec[j.col_off + tid] : 0.0f;
    }
  }
  __syncthreads();
  const float* eb = (const float*)(gsm + (size_t)NT16 * ks * 1024);
  const long tiles = (j.P + GEMM_TILE - 1) / GEMM_TILE;
  const int chunks = (ks + PF - 1) / PF;
  const long tstride = (long)gridDim.x * GEMM_WAVES;
  // operand staging lives ACROSS tiles: the first chunk of the next tile is requested in the last k-chunk of the current one, so loads
  // are in flight during the epilogue's store phase too.  Loads are UNCONDITIONAL (a branch around a load costs a vmcnt(0) at the
  // join and serialises the prefetch): rows are clamped, k-steps past the end re-read the last one, segments are chosen by selects.
  uint4 nxt[PF][2];
#pragma unroll
  for (int u = 0; u < PF; ++u) nxt[u][0] = nxt[u][1] = make_uint4(0u, 0u, 0u, 0u);
  auto row_of = [&](long tile, int g) { const long rr = tile * GEMM_TILE + 16 * g + jp; return rr < j.P ? rr : j.P - 1; };
  auto fetch = [&](int c, long rw0, long rw1, uint4 (&buf)[PF][2]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = c * PF + u < ks ? c * PF + u : ks - 1;
      const bool first = !SEG2 || s < ks0;
      const bf16_t* p = first ? j.a0.p : j.a1.p;
      const int ld = first ? j.a0.ld : j.a1.ld;
      const int col = 32 * (first ? s : s - ks0) + 8 * q4;
      buf[u][0] = *(const uint4*)(p + rw0 * ld + col);
      buf[u][1] = *(const uint4*)(p + rw1 * ld + col);
    }
  };
  {
    const long t0 = (long)blockIdx.x * GEMM_WAVES + wave;
    const long tt = t0 < tiles ? t0 : 0;
    if (!(j.dbg & 2)) fetch(0, row_of(tt, 0), row_of(tt, 1), nxt);
  }
  for (long tile = (long)blockIdx.x * GEMM_WAVES + wave; tile < tiles; tile += tstride) {
    gb_f32x4 acc[2][NT16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int t = 0; t < NT16; ++t) acc[g][t] = gb_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const long rw0 = row_of(tile, 0), rw1 = row_of(tile, 1);       // clamped: always readable
    const long tn = tile + tstride < tiles ? tile + tstride : tile;
    const long rn0 = row_of(tn, 0), rn1 = row_of(tn, 1);
    uint2 bin[2] = {make_uint2(~0u, ~0u), make_uint2(~0u, ~0u)};
    if (MASK) {
      bin[0] = *(const uint2*)(j.bits_in + (rw0 * 4 + q4) * 2);
      bin[1] = *(const uint2*)(j.bits_in + (rw1 * 4 + q4) * 2);
    }
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
      uint4 cur[PF][2];
#pragma unroll
      for (int u = 0; u < PF; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; }
      if (!(j.dbg & 2)) {                                  // the next chunk's operands fly while this one's MFMAs run; after the last chunk,
        const bool last = c + 1 == chunks;                 // the NEXT TILE's first chunk
        fetch(last ? 0 : c + 1, last ? rn0 : rw0, last ? rn1 : rw1, nxt);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int s = c * PF + u;
        if (s < ks && !(j.dbg & 4)) {
          const gb_bf16x8 b0 = __builtin_bit_cast(gb_bf16x8, cur[u][0]), b1 = __builtin_bit_cast(gb_bf16x8, cur[u][1]);
#pragma unroll
          for (int t = 0; t < NT16; ++t) {
            const gb_bf16x8 wf = *(const gb_bf16x8*)(gsm + ((size_t)(t * ks + s) * 64 + lane) * 16);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, b0, acc[0][t], 0, 0, 0);   // swapped: D[feature][point]
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, b1, acc[1][t], 0, 0, 0);
          }
        }
      }
    }
    // epilogue.  Swapped operands leave lane (jp, q4) with points 16 g + jp and, in accumulator t, features 16 t + 4 q4 + 0..3; the two
    // tiles of a 32-feature group are positions 32 u + 8 q4 + 0..7 of the stored row (perm32): one 16-byte store per group.
    if (j.dbg & 1) continue;
    const int u_off = j.col_off >> 5;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const long row = tile * GEMM_TILE + 16 * g + jp;
      const bool row_ok = row < j.P;
      const long rw = g ? rw1 : rw0;
      const float r1 = R1 ? j.r1_row[rw] : 0.0f;
      const unsigned long long bits_in = ((unsigned long long)bin[g].y << 32) | bin[g].x;
      unsigned long long bits_o = 0ull;
      float sg = 0.0f;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int ug = u_off + u;                                   // group index within the point's 256 features
        const uint32_t byte_in = MASK ? (uint32_t)(bits_in >> (8 * ug)) & 0xffu : 0xffu;
        uint32_t byte_out = 0u;
        float v[8];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          const int fcol = 32 * u + 16 * b2 + 4 * q4;               // < 32 NT; features beyond N (multiple of 32 here) do not exist
          const float4 b = *(const float4*)(eb + fcol);
          v[4 * b2 + 0] = acc[g][2 * u + b2][0] + b.x; v[4 * b2 + 1] = acc[g][2 * u + b2][1] + b.y;
          v[4 * b2 + 2] = acc[g][2 * u + b2][2] + b.z; v[4 * b2 + 3] = acc[g][2 * u + b2][3] + b.w;
          float4 cv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (R1 || SIG) cv = *(const float4*)(eb + 256 + fcol);
          if (R1) {
            v[4 * b2 + 0] = fmaf(r1, cv.x, v[4 * b2 + 0]); v[4 * b2 + 1] = fmaf(r1, cv.y, v[4 * b2 + 1]);
            v[4 * b2 + 2] = fmaf(r1, cv.z, v[4 * b2 + 2]); v[4 * b2 + 3] = fmaf(r1, cv.w, v[4 * b2 + 3]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float& x = v[4 * b2 + e];
            if (ACT == ACT_RELU) { x = fmaxf(x, 0.0f); byte_out |= (x > 0.0f ? 1u : 0u) << (4 * b2 + e); }
            else if (ACT == ACT_SIGMOID) x = sigmoid_ref(x);
            if (MASK) x = ((byte_in >> (4 * b2 + e)) & 1u) ? x : 0.0f;
          }
          if (SIG) {   // static_sigma on the un-rounded relu output (nerf.py:146,172)
            sg = fmaf(cv.x, v[4 * b2 + 0], sg); sg = fmaf(cv.y, v[4 * b2 + 1], sg); sg = fmaf(cv.z, v[4 * b2 + 2], sg); sg = fmaf(cv.w, v[4 * b2 + 3], sg);
          }
        }
        bits_o |= (unsigned long long)byte_out << (8 * ug);
        if (row_ok) {
          if (ACT == ACT_SIGMOID) {
            float* o = j.out_f + row * j.ldo_f + j.col_off + 32 * u + 4 * q4;
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
              for (int e = 0; e < 4; ++e) o[16 * b2 + e] = v[4 * b2 + e];
          } else {
            bf16_t* o = j.out + row * j.ldo + j.col_off + 32 * u + 8 * q4;   // the four lanes of a point write 64 contiguous bytes
            *(uint4*)o = make_uint4(gb_pk(v[0], v[1]), gb_pk(v[2], v[3]), gb_pk(v[4], v[5]), gb_pk(v[6], v[7]));
          }
        }
      }
      if (SIG) {
        sg += __shfl_xor(sg, 16);
        sg += __shfl_xor(sg, 32);
        if (row_ok && q4 == 0) j.sig_out[row] = softplus_ref(sg + j.sig_b[0]);
      }
      if (ACT == ACT_SIGMOID && j.sig_in) {
        const float sgv = j.sig_in[rw];
        if (row_ok && q4 == 0) j.out_f[row * j.ldo_f + FEAT_DIM] = sgv;
      }
      if (ACT == ACT_RELU && j.bits_out && row_ok) {              // the bytes this pass covers: 8 groups = 8 bytes, 4 groups = 4 bytes
        uint32_t* bo = j.bits_out + (row * 4 + q4) * 2;
        if (NT == 8) *(uint2*)bo = make_uint2((uint32_t)bits_o, (uint32_t)(bits_o >> 32));
        else if (NT == 4) bo[u_off >> 2] = (uint32_t)(bits_o >> (8 * u_off));
      }
    }
  }
}

static int run_gemm(const GemmJob& j_in, hipStream_t st) {
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
  const bool seg2 = j.a1.p != nullptr, mask = j.bits_in != nullptr, r1 = j.r1_row != nullptr, sig = j.sig_out != nullptr;
  if (j.N != 32 * nt || (j.col_off & 31)) return set_error(-2, "linear_bf16: feature count and offset must be multiples of 32");
#define CRNERF_GEMM(NTV, SEG2V, ACTV, MASKV, R1V, SIGV)                                                                                             \
  if (nt == NTV && seg2 == SEG2V && j.act == ACTV && mask == MASKV && r1 == R1V && sig == SIGV) {                                                  \
    if (int rc = ensure_dynamic_lds((const void*)linear_bf16_kernel<NTV, SEG2V, ACTV, MASKV, R1V, SIGV>, GEMM_LDS_BYTES, "linear_bf16_kernel")) \
      return rc;                                                                                                                                    \
    hipLaunchKernelGGL((linear_bf16_kernel<NTV, SEG2V, ACTV, MASKV, R1V, SIGV>), dim3(grid), dim3(64 * GEMM_WAVES), shmem, st, j);               \
    return 0;                                                                                                                                       \
  }
  CRNERF_GEMM(8, false, ACT_RELU, false, false, false)      // xyz_encoding_1..4, 6, 7
  CRNERF_GEMM(8, false, ACT_RELU, false, false, true)       // xyz_encoding_8 + static_sigma
  CRNERF_GEMM(4, true, ACT_RELU, false, false, false)       // xyz_encoding_5 (two passes), dir_encoding
  CRNERF_GEMM(8, false, ACT_NONE, false, false, false)      // xyz_encoding_final; d(final)
  CRNERF_GEMM(2, false, ACT_SIGMOID, false, false, false)   // static_rgb
  CRNERF_GEMM(4, false, ACT_NONE, true, false, false)       // d(dir act)
  CRNERF_GEMM(8, false, ACT_NONE, true, true, false)        // d(h8): + the sigma head's branch
  CRNERF_GEMM(8, false, ACT_NONE, true, false, false)       // d(h7..h1)
#undef CRNERF_GEMM
  return set_error(-2, "linear_bf16: unsupported shape");
}

// d_rgb_pre = d_out[:, :64] * f (1 - f) -> bf16;  d_sig_pre = d_out[:, 64] * (1 - exp(-sigma)) -> fp32    (sigmoid', softplus')
__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ out, const float* __restrict__ d_out, bf16_t* __restrict__ d_rgb,
                                                        float* __restrict__ d_sig, long P) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per point and pair of features (+ one for sigma)
  if (idx >= P * 33) return;
  const long p = idx / 33;
  const int c = (int)(idx - p * 33);
  const float* o = out + p * OUT_DIM;
  const float* g = d_out + p * OUT_DIM;
  if (c < 32) {
    const float f0 = o[2 * c], f1 = o[2 * c + 1];
    *(uint32_t*)(d_rgb + p * DRGB_W + 2 * c) = gb_pk(g[2 * c] * f0 * (1.0f - f0), g[2 * c + 1] * f1 * (1.0f - f1));
  } else {
    d_sig[p] = g[FEAT_DIM] * (1.0f - expf(-o[FEAT_DIM]));
  }
}

// ---- weight gradients from the stored bf16 rows: C[m][n] = sum_p D[p][m] * A[p][n] on v_mfma_f32_32x32x16_bf16 (k = 16 points).
// Workgroup = a 256 x 256 block of C over a chunk of points; a wave owns a 128 x 128 sub-block (4 x 4 MFMA tiles) -- MFMA tile t,
// lane i <-> column 4 i + t, so a lane's four columns of a point are ONE 8-byte load (256 contiguous bytes per half-wave) -- and
// assembles the 8-point operand with v_perm_b32; lane (i, kk) supplies points 8 kk .. 8 kk + 7 of the k-step.  Blocks narrower than
// 256 leave waves without a sub-block: those split the chunk's POINTS instead (wp sub-chunks, each with its own partial-sum slot),
// and columns beyond M / N are loaded from a valid address and dropped at the output -- the matrix work is 16x cheaper than in
// fp32, so a whole-tile MFMA on a 27-column block costs nothing next to the row traffic.  (A first version loaded the narrow blocks
// column by column, 2 bytes per lane: 26.8 ms of wgrad in a 16,384-ray step against 15.1 ms for the fp32-storage kernel.)
// Partial sums per (chunk, sub-chunk) are reduced by wgrad_reduce_kernel (mlp_train16.hip), deterministically.
// permD / permA: the operand's columns are in storage order -- the output index is un-permuted when dW is written.
struct WgradBJob {
  const bf16_t* D; int ldd; int M; int Dw; int permD;    // Dw / Aw: readable columns (multiple of 4, >= M / N)
  const bf16_t* A; int lda; int N; int Aw; int permA;
  float* partial;                      // [nchunk * wp][M][N], reference order
  float* bias_partial;                 // [nchunk * wp][M] column sums of D (bias gradient) or null
  long P; int chunk;
};

__host__ __device__ inline int wgb_wp(int M, int N) { return 4 / (((M < 256 ? M : 256) + 127) / 128 * (((N < 256 ? N : 256) + 127) / 128)); }

__global__ __launch_bounds__(256, 1) void wgrad_b_kernel(WgradBJob j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int bm = j.M - (int)blockIdx.y * 256 < 256 ? j.M - (int)blockIdx.y * 256 : 256;   // live extent of this block
  const int bn = j.N - (int)blockIdx.z * 256 < 256 ? j.N - (int)blockIdx.z * 256 : 256;
  const int wm = (bm + 127) / 128, wn = (bn + 127) / 128, wp = 4 / (wm * wn);
  const int wmi = wave % wm, wni = (wave / wm) % wn, wpi = wave / (wm * wn);
  const int m0 = blockIdx.y * 256 + wmi * 128, n0 = blockIdx.z * 256 + wni * 128;
  // Points are dealt out in 16-point k-steps, ROUND-ROBIN over the grid's workgroups (and, in narrow blocks, over the waves that split
  // the points): at any moment the whole chip reads one neighbourhood of the two operand arrays.  (Contiguous per-workgroup chunks
  // -- 512 far-apart streams -- ran the full blocks at 4.5 TB/s)
  const long kstride = (long)gridDim.x * wp;                     // in k-steps
  const long p0 = ((long)blockIdx.x * wp + wpi) * 16;            // first point of this wave's first k-step (may lie beyond P: no work)
  const long p1 = j.P;
  const long pstep = kstride * 16;
  const long slot = (long)blockIdx.x * wp + wpi;
  gb_f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  const int mc = m0 + 4 * i, nc = n0 + 4 * i;                                             // this lane's first column of each operand
  const long plast = j.P - 1;
  const bool do_bias = j.bias_partial && blockIdx.z == 0 && wni == 0;
  gb_f32x4 bsum = {0.0f, 0.0f, 0.0f, 0.0f};
  // one k-step: operands d[e], a[e] = this lane's four columns of points 8 kk + e of the step.  Rows past the (sub-)chunk are zeroed
  // HERE, where they are used -- a select right behind the load sits behind an s_waitcnt for it and exposes the memory latency in
  // every k-step; the other operand's rows are clamped to real (finite) rows, so 0 x them is 0.
#define CRNERF_WGB_MULTIPLY(D8, A8, PB)                                                                                              \
  {                                                                                                                                  \
    uint2 dm[8];                                                                                                                     \
    const int left = (int)(p1 - (PB) < 16 ? p1 - (PB) : 16) - 8 * kk;   /* valid points of this lane's eight; 8 or more except in the tail */ \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                                  \
      const bool keep = e < left;                                                                                                    \
      dm[e] = make_uint2(keep ? D8[e].x : 0u, keep ? D8[e].y : 0u);                                                                  \
    }                                                                                                                                \
    if (do_bias) {                                                                                                                   \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                                \
        bsum[0] += gb_lo(dm[e].x); bsum[1] += gb_hi(dm[e].x); bsum[2] += gb_lo(dm[e].y); bsum[3] += gb_hi(dm[e].y);                  \
      }                                                                                                                              \
    }                                                                                                                                \
    gb_bf16x8 df[4], af[4];                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                                  \
      uint32_t wd[4], wa[4];                                                                                                         \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {   /* column t of the lane's four, points 2q, 2q+1 -> one dword of the operand */ \
        const uint32_t sel = (t & 1) ? 0x07060302u : 0x05040100u;                                                                    \
        wd[q] = __builtin_amdgcn_perm((t >> 1) ? dm[2 * q + 1].y : dm[2 * q + 1].x, (t >> 1) ? dm[2 * q].y : dm[2 * q].x, sel);      \
        wa[q] = __builtin_amdgcn_perm((t >> 1) ? A8[2 * q + 1].y : A8[2 * q + 1].x, (t >> 1) ? A8[2 * q].y : A8[2 * q].x, sel);      \
      }                                                                                                                              \
      df[t] = __builtin_bit_cast(gb_bf16x8, make_uint4(wd[0], wd[1], wd[2], wd[3]));                                                 \
      af[t] = __builtin_bit_cast(gb_bf16x8, make_uint4(wa[0], wa[1], wa[2], wa[3]));                                                 \
    }                                                                                                                                \
    _Pragma("unroll") for (int a4 = 0; a4 < 4; ++a4)                                                                                 \
      _Pragma("unroll") for (int b4 = 0; b4 < 4; ++b4)                                                                               \
        acc[a4][b4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[a4], af[b4], acc[a4][b4], 0, 0, 0);                                 \
  }
  // addresses: 32-bit element offsets from the (sub-)chunk's first row, advanced by a constant per k-step and clamped to the last row of
  // the array (a 64-bit multiply per load was ~10 VALU instructions x 32 loads per k-step -- as many cycles as the 16 MFMAs)
  if (wp == 1 && p0 < p1) {
    // Full 256 x 256 block: the four waves multiply the SAME points, so the workgroup fetches each 512-byte row of both operands
    // ONCE -- 16-byte loads, whole rows per instruction -- into LDS and every wave picks its 8-byte column pieces from there (read
    // directly, each row half is fetched by two waves, 8 bytes per lane: 3.5 TB/s of HBM reads against 6.3 for a plain row copy,
    // tools/ubench/row_patterns.hip).  Double-buffered, one __syncthreads per 16-point k-step.
    __shared__ uint4 sh[2][2][16][33];           // [buffer][D, A][point][32 units of 16 B + pad]
    const int u = threadIdx.x & 31, rrow = threadIdx.x >> 5;
    const bf16_t* dch = j.D + p0 * j.ldd + blockIdx.y * 256 + 8 * u;
    const bf16_t* ach = j.A + p0 * j.lda + blockIdx.z * 256 + 8 * u;
    const uint32_t dlast = (uint32_t)(plast - p0) * (uint32_t)j.ldd, alast = (uint32_t)(plast - p0) * (uint32_t)j.lda;
    uint32_t do0 = (uint32_t)rrow * j.ldd, do1 = (uint32_t)(rrow + 8) * j.ldd, ao0 = (uint32_t)rrow * j.lda, ao1 = (uint32_t)(rrow + 8) * j.lda;
    const uint32_t dstep = (uint32_t)pstep * j.ldd, astep = (uint32_t)pstep * j.lda;
    uint4 g0, g1, g2, g3, h0, h1, h2, h3;          // the rows of k-steps n + 1 (g) and n + 2 (h) in flight while n is multiplied:
                                                   // one k-step is 16 KiB per workgroup = 4 MB on the chip, ~1 us of HBM at 4 TB/s
#define CRNERF_WGB_GFETCH(R0, R1, R2, R3)                                       \
    R0 = *(const uint4*)(dch + (do0 < dlast ? do0 : dlast));                    \
    R1 = *(const uint4*)(ach + (ao0 < alast ? ao0 : alast));                    \
    R2 = *(const uint4*)(dch + (do1 < dlast ? do1 : dlast));                    \
    R3 = *(const uint4*)(ach + (ao1 < alast ? ao1 : alast));                    \
    do0 += dstep; do1 += dstep; ao0 += astep; ao1 += astep;
#define CRNERF_WGB_LSTORE(B, R0, R1, R2, R3)                                    \
    sh[B][0][rrow][u] = R0; sh[B][1][rrow][u] = R1; sh[B][0][rrow + 8][u] = R2; sh[B][1][rrow + 8][u] = R3;
#define CRNERF_WGB_STEP(PB, BUF, F0, F1, F2, F3, S0, S1, S2, S3)                \
    {                                                                           \
      CRNERF_WGB_GFETCH(F0, F1, F2, F3)        /* k-step n + 2 -> the free register set */ \
      uint2 dcur[8], acur[8];                                                   \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                           \
        dcur[e] = ((const uint2*)&sh[BUF][0][8 * kk + e][0])[wmi * 32 + i];     \
        acur[e] = ((const uint2*)&sh[BUF][1][8 * kk + e][0])[wni * 32 + i];     \
      }                                                                         \
      CRNERF_WGB_MULTIPLY(dcur, acur, PB)                                       \
      CRNERF_WGB_LSTORE(BUF ^ 1, S0, S1, S2, S3) /* k-step n + 1, fetched one step ago */  \
      __syncthreads();                                                          \
    }
    // the two register sets swap roles every k-step (loop unrolled by two): rotating them through copies (g = h) made every
    // iteration wait for the loads it had just issued
    CRNERF_WGB_GFETCH(g0, g1, g2, g3)
    CRNERF_WGB_LSTORE(0, g0, g1, g2, g3)
    CRNERF_WGB_GFETCH(g0, g1, g2, g3)
    __syncthreads();
    for (long pb = p0; pb < p1; pb += 2 * pstep) {
      CRNERF_WGB_STEP(pb, 0, h0, h1, h2, h3, g0, g1, g2, g3)
      if (pb + pstep < p1) CRNERF_WGB_STEP(pb + pstep, 1, g0, g1, g2, g3, h0, h1, h2, h3)
    }
#undef CRNERF_WGB_STEP
#undef CRNERF_WGB_GFETCH
#undef CRNERF_WGB_LSTORE
  } else if (wp != 1 && p0 < p1) {
    const bf16_t* dch = j.D + p0 * j.ldd + (mc < j.Dw ? mc : 0);
    const bf16_t* ach = j.A + p0 * j.lda + (nc < j.Aw ? nc : 0);
    const uint32_t dlast = (uint32_t)(plast - p0) * (uint32_t)j.ldd, alast = (uint32_t)(plast - p0) * (uint32_t)j.lda;
    const uint32_t dstep = (uint32_t)pstep * j.ldd, astep = (uint32_t)pstep * j.lda;
    uint32_t dof = (uint32_t)(8 * kk) * j.ldd, aof = (uint32_t)(8 * kk) * j.lda;      // offset of point 8 kk of the step being fetched
    uint2 dcur[8], acur[8], dnxt[8], anxt[8];
#define CRNERF_WGB_FETCH16(D8, A8)                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                        \
      const uint32_t od = dof + (uint32_t)e * j.ldd, oa = aof + (uint32_t)e * j.lda;                       \
      D8[e] = *(const uint2*)(dch + (od < dlast ? od : dlast));                                            \
      A8[e] = *(const uint2*)(ach + (oa < alast ? oa : alast));                                            \
    }                                                                                                      \
    dof += dstep; aof += astep;
    CRNERF_WGB_FETCH16(dcur, acur)
    for (long pb = p0; pb < p1; pb += 2 * pstep) {     // the two operand sets swap roles (no copies: a copy waits for its loads)
      CRNERF_WGB_FETCH16(dnxt, anxt)
      CRNERF_WGB_MULTIPLY(dcur, acur, pb)
      if (pb + pstep < p1) {
        CRNERF_WGB_FETCH16(dcur, acur)
        CRNERF_WGB_MULTIPLY(dnxt, anxt, pb + pstep)
      }
    }
#undef CRNERF_WGB_FETCH16
  }
#undef CRNERF_WGB_MULTIPLY
  if (do_bias) {
#pragma unroll
    for (int t = 0; t < 4; ++t) bsum[t] += __shfl_xor(bsum[t], 32);
    if (kk == 0 && mc < j.M) *(gb_f32x4*)(j.bias_partial + slot * j.M + ord_to_ref(j.permD, mc)) = bsum;   // M is a multiple of 4
  }
  float* outp = j.partial + slot * j.M * j.N;
  const bool slots = j.permA >= ORD_XYZ_SLOTS;                  // embedding slots: every column finds its own place (or is padding)
  const int fn = slots ? 0 : ord_to_ref(j.permA, nc);           // the row permutations keep groups of four together
  int fs[4] = {0, 0, 0, 0};
  if (slots) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fs[b] = nc + b < j.Aw ? ord_to_ref(j.permA, nc + b) : -1;
  }
  const bool vec = (j.N & 3) == 0 && !slots;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + a;
      if (m >= j.M || nc >= (slots ? j.Aw : j.N)) continue;
      float* o = outp + (long)ord_to_ref(j.permD, m) * j.N + fn;
      if (vec) *(gb_f32x4*)o = gb_f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
      else if (slots) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (fs[b] >= 0) o[fs[b]] = acc[a][b][r];
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (nc + b < j.N) o[b] = acc[a][b][r];
      }
    }
}

// static_sigma: dW[n] = sum_p d_sig[p] * h8[p][n], db = sum_p d_sig[p] -- a weighted column sum over the stored h8 rows (512 B per
// point).  Thread = (16-byte unit of the row, one of 8 row slots); four rows per thread in flight; partial[block][257] in storage order.
constexpr int SIGW_BLOCKS = 256;
__global__ __launch_bounds__(256) void sigma_wgrad_kernel(const bf16_t* __restrict__ h8, const float* __restrict__ d_sig, long P, long chunk,
                                                          float* __restrict__ partial) {
  __shared__ float red[8][260];
  const int u = threadIdx.x & 31, rs = threadIdx.x >> 5;
  const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bs = 0.0f;
  for (long p = p0 + rs; p < p1; p += 32) {
    uint4 v[4];
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long pp = p + 8 * e;
      const bool ok = pp < p1;
      const long pc = ok ? pp : p1 - 1;
      v[e] = *(const uint4*)(h8 + pc * ACT_W + 8 * u);
      g[e] = ok ? d_sig[pc] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[0] = fmaf(g[e], gb_lo(v[e].x), a[0]); a[1] = fmaf(g[e], gb_hi(v[e].x), a[1]);
      a[2] = fmaf(g[e], gb_lo(v[e].y), a[2]); a[3] = fmaf(g[e], gb_hi(v[e].y), a[3]);
      a[4] = fmaf(g[e], gb_lo(v[e].z), a[4]); a[5] = fmaf(g[e], gb_hi(v[e].z), a[5]);
      a[6] = fmaf(g[e], gb_lo(v[e].w), a[6]); a[7] = fmaf(g[e], gb_hi(v[e].w), a[7]);
      bs += g[e];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rs][8 * u + k] = a[k];
  if (u == 0) red[rs][256] = bs;
  __syncthreads();
  for (int c = threadIdx.x; c < 257; c += 256) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][c];
    partial[(long)blockIdx.x * 257 + c] = s;
  }
}

__global__ __launch_bounds__(320) void sigma_wgrad_finish_kernel(const float* __restrict__ sums, float* __restrict__ dw, float* __restrict__ db, int fused) {
  const int f = threadIdx.x;
  if (f < 256) dw[f] = sums[fused ? perm_fused(f) : perm32(f)];
  else if (f == 256) db[0] = sums[256];
}

static int wgb_chunk(long P) {   // points per workgroup (multiple of the 16-point k-step), as wg_chunk of mlp_train16.hip
  long c = (P + 255) / 256;
  c = (c + 15) / 16 * 16;
  return (int)(c < 128 ? 128 : c);
}

static int wgrad_b(const bf16_t* D, int ldd, int M, int Dw, int permD, const bf16_t* A, int lda, int N, int Aw, int permA, float* dst, int ldc,
                   float* db, long P, float* ws, hipStream_t st) {
  if ((unsigned long long)P * (unsigned)(ldd > lda ? ldd : lda) >= (1ull << 32))
    return set_error(-2, "mlp_backward_mixed: more than 2^24 points per call (the weight-gradient kernel addresses rows with 32-bit offsets)");
  const int chunk = wgb_chunk(P);
  const int nchunk = (int)((P + chunk - 1) / chunk);
  const int slots = nchunk * wgb_wp(M, N);
  float* bws = ws + (size_t)slots * M * N;
  WgradBJob j{D, ldd, M, Dw, permD, A, lda, N, Aw, permA, ws, db ? bws : nullptr, P, chunk};
  hipLaunchKernelGGL(wgrad_b_kernel, dim3(nchunk, (M + 255) / 256, (N + 255) / 256), dim3(256), 0, st, j);
  return launch_wgrad_reduce(ws, slots, M, N, dst, ldc, bws, db, st);
}

// ---- buffers.  acts: [10][P][256] bf16 activations (storage order) | [10][P] x 32 B relu bits | xb [P][128] bf16 | sigma [P] fp32.
//      scratch: [10][P][256] bf16 deltas | d_rgb [P][64] bf16 | d_sig [P] fp32 | (16-byte aligned) weight-gradient workspace.
static size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }
size_t mlp_train_mixed_acts_bytes(long P) { return (size_t)P * (ACT_SLOTS * ACT_W * 2 + ACT_SLOTS * 32 + XB_W * 2 + 4); }
static size_t mixed_ws_offset(long P) { return align16((size_t)P * (ACT_SLOTS * ACT_W * 2 + DRGB_W * 2 + 4)); }
size_t mlp_train_mixed_scratch_bytes(long P) {
  const int chunk = wgb_chunk(P);
  const size_t nchunk = (size_t)((P + chunk - 1) / chunk);
  return mixed_ws_offset(P) + (nchunk * (256 * 256 + 256) + (size_t)SIGW_BLOCKS * 257 + 512) * 4;
}

static const uint4* frag_ptr(const void* packed, int id) { return (const uint4*)packed + (size_t)layout().m[id].off_frag * 64; }

int launch_mlp_forward_train_mixed(const MlpTensors& t, const void* packed, const float* x, float* out, void* acts, long P, hipStream_t st) {
  if (P <= 0) return 0;
  const GemmLayout& L = layout();
  bf16_t* abase = (bf16_t*)acts;
  auto A = [&](int slot) { return abase + (size_t)slot * P * ACT_W; };
  uint32_t* bits_base = (uint32_t*)(abase + (size_t)ACT_SLOTS * P * ACT_W);
  auto bits = [&](int slot) { return bits_base + (size_t)slot * P * 8; };   // 32 bytes per point and slot
  bf16_t* xb = (bf16_t*)(bits_base + (size_t)ACT_SLOTS * P * 8);
  float* sig_tmp = (float*)(xb + (size_t)P * XB_W);
  hipLaunchKernelGGL(embed_bf16_kernel, dim3((unsigned)((P * (XB_W / 8) + 255) / 256)), dim3(256), 0, st, x, xb, P);
  const GemmSeg none{nullptr, 0, 0};
  // tiles [t0, t0 + nt) of matrix `id` as one pass (a 256 x 352 matrix does not fit the 128 KiB of LDS: the skip layer runs as two
  // 128-feature passes and reads its input twice)
  auto gemm = [&](int id, int t0, int nt, GemmSeg a0, GemmSeg a1, const float* bias, int act, bf16_t* o, uint32_t* bo, bool sig = false) {
    const int ks = L.m[id].K / 16;
    const int n = L.m[id].N - 32 * t0 < 32 * nt ? L.m[id].N - 32 * t0 : 32 * nt;
    GemmJob j{a0, a1, frag_ptr(packed, id) + (size_t)t0 * ks * 64, n, L.m[id].K, 32 * t0, bias, act, nullptr, bo, nullptr,
              sig ? t.w_sigma : nullptr, sig ? t.b_sigma : nullptr, sig ? sig_tmp : nullptr, act == ACT_SIGMOID ? sig_tmp : nullptr, o, ACT_W, out, OUT_DIM, P, 0};
    return run_gemm(j, st);
  };
  const GemmSeg emb{xb, XB_W, 96};
  if (int rc = gemm(GM_L1, 0, 8, emb, none, t.b[0], ACT_RELU, A(0), bits(0))) return rc;
  for (int l = 1; l < 8; ++l) {
    const GemmSeg h{A(l - 1), ACT_W, 256};
    if (l == 4) {
      if (int rc = gemm(GM_L5, 0, 4, emb, h, t.b[4], ACT_RELU, A(4), bits(4))) return rc;
      if (int rc = gemm(GM_L5, 4, 4, emb, h, t.b[4], ACT_RELU, A(4), bits(4))) return rc;
    } else if (int rc = gemm(GM_L1 + l, 0, 8, h, none, t.b[l], ACT_RELU, A(l), bits(l), l == 7)) return rc;   // layer 8 carries static_sigma
  }
  if (int rc = gemm(GM_FINAL, 0, 8, GemmSeg{A(7), ACT_W, 256}, none, t.b_final, ACT_NONE, A(8), nullptr)) return rc;
  if (int rc = gemm(GM_DIR, 0, 4, GemmSeg{A(8), ACT_W, 256}, GemmSeg{xb + XB_DIR, XB_W, 32}, t.b_dir, ACT_RELU, A(9), bits(9))) return rc;
  if (int rc = gemm(GM_RGB, 0, 2, GemmSeg{A(9), ACT_W, 128}, none, t.b_rgb, ACT_SIGMOID, nullptr, nullptr)) return rc;
  return check_launch("mlp_forward_train_mixed");
}

// acts_layout: 0 = written by launch_mlp_forward_train_mixed (rows in perm32 order, xb in reference column order); 1 = written by the fused
// renderer's training twin (crnerf_render_rays_train_bf16: rows in "fused" order, xb in embedding-slot order).  The deltas this function
// produces are perm32 either way; the activity bits have one layout.
int launch_mlp_backward_mixed(const MlpTensors& t, const void* packed, const float* x, const float* out, const float* d_out, const void* acts,
                              void* scratch, float* const* grads, long P, hipStream_t st, int acts_layout) {
  if (P <= 0) return 0;
  const int oA = acts_layout ? ORD_FUSED : ORD_PERM32;
  const int oX = acts_layout ? ORD_XYZ_SLOTS : ORD_REF, oDir = acts_layout ? ORD_DIR_SLOTS : ORD_REF;
  (void)x;                                   // the bf16 copy the forward left in `acts` is what the weight gradients read
  const GemmLayout& L = layout();
  const bf16_t* abase = (const bf16_t*)acts;
  auto A = [&](int slot) { return abase + (size_t)slot * P * ACT_W; };
  const uint32_t* bits_base = (const uint32_t*)(abase + (size_t)ACT_SLOTS * P * ACT_W);
  auto bits = [&](int slot) { return bits_base + (size_t)slot * P * 8; };
  const bf16_t* xb = (const bf16_t*)(bits_base + (size_t)ACT_SLOTS * P * 8);
  bf16_t* deltas = (bf16_t*)scratch;
  auto D = [&](int slot) { return deltas + (size_t)slot * P * ACT_W; };
  bf16_t* d_rgb = deltas + (size_t)ACT_SLOTS * P * ACT_W;
  float* d_sig = (float*)(d_rgb + (size_t)P * DRGB_W);
  float* ws = (float*)((char*)scratch + mixed_ws_offset(P));
  hipLaunchKernelGGL(head_grad_kernel, dim3((unsigned)((P * 33 + 255) / 256)), dim3(256), 0, st, out, d_out, d_rgb, d_sig, P);
  const GemmSeg none{nullptr, 0, 0};
  auto dg = [&](int id, GemmSeg a0, const uint32_t* mask, const float* r1r, const float* r1c, bf16_t* o) {
    GemmJob j{a0, none, frag_ptr(packed, id), L.m[id].N, L.m[id].K, 0, nullptr, ACT_NONE, mask, nullptr, r1r, r1c, nullptr, nullptr, nullptr, o, ACT_W,
              nullptr, 0, P, 0};
    return run_gemm(j, st);
  };
  if (int rc = dg(GM_T_RGB, GemmSeg{d_rgb, DRGB_W, 64}, bits(9), nullptr, nullptr, D(9))) return rc;           // through static_rgb, relu' of dir act
  if (int rc = dg(GM_T_DIR, GemmSeg{D(9), ACT_W, 128}, nullptr, nullptr, nullptr, D(8))) return rc;           // through dir_encoding[:, :256] (final is linear)
  if (int rc = dg(GM_T_FINAL, GemmSeg{D(8), ACT_W, 256}, bits(7), d_sig, t.w_sigma, D(7))) return rc;         // through final + the sigma head, relu' of h8
  for (int l = 7; l >= 1; --l)                                                                                // through xyz_encoding_{l+1}, relu' of h_l
    if (int rc = dg(GM_T8 + (7 - l), GemmSeg{D(l), ACT_W, 256}, bits(l - 1), nullptr, nullptr, D(l - 1))) return rc;
  if (int rc = check_launch("mlp_backward_mixed")) return rc;
  // weight / bias gradients of the eleven nn.Linear (grads in crnerf.h tensor order)
  if (int rc = wgrad_b(D(0), ACT_W, 256, 256, 1, xb, XB_W, XYZ_DIM, 96, oX, grads[0], XYZ_DIM, grads[1], P, ws, st)) return rc;          // xyz_encoding_1
  for (int l = 1; l < 8; ++l) {
    if (l == 4) {                                                                                              // xyz_encoding_5: cat([xyz, h4])
      if (int rc = wgrad_b(D(4), ACT_W, 256, 256, 1, xb, XB_W, XYZ_DIM, 96, oX, grads[8], XYZ_DIM + 256, grads[9], P, ws, st)) return rc;
      if (int rc = wgrad_b(D(4), ACT_W, 256, 256, 1, A(3), ACT_W, 256, 256, oA, grads[8] + XYZ_DIM, XYZ_DIM + 256, nullptr, P, ws, st)) return rc;
    } else {
      if (int rc = wgrad_b(D(l), ACT_W, 256, 256, 1, A(l - 1), ACT_W, 256, 256, oA, grads[2 * l], 256, grads[2 * l + 1], P, ws, st)) return rc;
    }
  }
  if (int rc = wgrad_b(D(8), ACT_W, 256, 256, 1, A(7), ACT_W, 256, 256, oA, grads[16], 256, grads[17], P, ws, st)) return rc;             // xyz_encoding_final
  {                                                                                                            // static_sigma
    const long chunk = (P + SIGW_BLOCKS - 1) / SIGW_BLOCKS;
    const int nblk = (int)((P + chunk - 1) / chunk);
    float* part = ws;
    float* sums = ws + (size_t)SIGW_BLOCKS * 257;
    hipLaunchKernelGGL(sigma_wgrad_kernel, dim3(nblk), dim3(256), 0, st, A(7), d_sig, P, chunk, part);
    if (int rc = launch_wgrad_reduce(part, nblk, 1, 257, sums, 257, nullptr, nullptr, st)) return rc;
    hipLaunchKernelGGL(sigma_wgrad_finish_kernel, dim3(1), dim3(320), 0, st, sums, grads[18], grads[19], acts_layout);
  }
  if (int rc = wgrad_b(D(9), ACT_W, 128, 128, 1, A(8), ACT_W, 256, 256, oA, grads[20], 256 + DIR_DIM, grads[21], P, ws, st)) return rc;   // dir_encoding: cat([final, dir])
  if (int rc = wgrad_b(D(9), ACT_W, 128, 128, 1, xb + XB_DIR, XB_W, DIR_DIM, 32, oDir, grads[20] + 256, 256 + DIR_DIM, nullptr, P, ws, st)) return rc;
  if (int rc = wgrad_b(d_rgb, DRGB_W, FEAT_DIM, DRGB_W, 0, A(9), ACT_W, 128, 128, oA, grads[22], 128, grads[23], P, ws, st)) return rc;   // static_rgb
  return check_launch("mlp_backward_mixed wgrad");
}

}  // namespace crnerf
