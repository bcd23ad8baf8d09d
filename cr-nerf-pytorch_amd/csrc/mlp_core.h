// NeRF_sigma forward for one 32-point tile per wavefront, fp32 MFMA, activations register-resident.
//
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182 (8x256 trunk with a skip at
// layer 5, softplus sigma head, 256->256 linear, dir layer 283->128 relu, 128->64 sigmoid).
//
// MI355X design (not a translation of the reference's eager addmm chain):
//   * swapped-operand GEMM: D[feature][point] = W[feature][k] * act[k][point] with
//     v_mfma_f32_32x32x2_f32.  Lane (p = lane&31, h = lane>>5) owns point p; the C/D layout leaves
//     it holding features 32t + 8q + 4h + j (reg 4q+j of tile t), which is exactly the set of
//     k-values it must supply as the B operand of the next layer -- so a 32-point tile's
//     activations NEVER leave the register file across the 11 layers (no LDS round trip, no HBM).
//   * the A operand (weights) is shared by the 4 waves of a workgroup (one wave per SIMD) and is
//     streamed HBM/L2 -> LDS in 16 KiB stages with global_load_lds (direct-to-LDS DMA) through a
//     ring of RING_SLOTS stages, PF_DIST stages ahead; fragments are pre-packed (layout.h) so a
//     wave's ds_read_b128 is lane-linear and bank-conflict-free.
//   * one s_barrier per 64 MFMAs (4096 MFMA cycles); the barrier for stage s also certifies that
//     stage s+1 has landed, so reads may run ahead of the next barrier.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

namespace crnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) char gbl_char;

constexpr int RING_SLOTS = 6;
constexpr int PF_DIST = 4;
static_assert(PF_DIST >= 2 && PF_DIST <= RING_SLOTS - 2, "ring too shallow for the lookahead protocol");

// LDS map: byte offsets inside the single dynamic __shared__ array of every kernel using the core
constexpr int LDS_CONST0 = 0;
constexpr int LDS_CONST1 = CONST_BYTES;
constexpr int LDS_RING = 2 * CONST_BYTES;
constexpr int LDS_SCRATCH = LDS_RING + RING_SLOTS * STAGE_BYTES;  // 120,832

#define CRNERF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Streams packed weights into the LDS ring.  All members except pf_ptr / rd_addr are wave-uniform.
struct WeightPipe {
  lds_char* lds;
  gbl_char* base[2];   // per-lane pointers into the two packed streams (wave*4 KiB + lane*16 added)
  gbl_char* pf_ptr;    // next stage to fetch
  int pf_left;         // stages left in the pass being fetched
  int pf_pass;         // pass index inside the cycle
  int passes0;         // the first passes0 passes of a cycle use base[0], the rest base[1]
  int passes;          // passes per cycle
  uint32_t pf_slot;
  uint32_t rd_slot;
  uint32_t rd_addr;    // per-lane LDS byte address of fragment 0 of the stage being consumed
  uint32_t lane16;
  uint32_t wave4k;

  __device__ __forceinline__ void issue() {
    lds_char* dst = lds + LDS_RING + pf_slot * STAGE_BYTES + wave4k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(pf_ptr + i * FRAG_BYTES, dst + i * FRAG_BYTES, 16, 0, 0);
    pf_slot = (pf_slot + 1 == RING_SLOTS) ? 0u : pf_slot + 1;
    pf_ptr += STAGE_BYTES;
    if (--pf_left == 0) {
      pf_left = STAGES_PER_PASS;
      pf_pass = (pf_pass + 1 == passes) ? 0 : pf_pass + 1;
      pf_ptr = (pf_pass < passes0) ? base[0] : base[1];
    }
  }

  // Call once, all waves, before the first stage_begin().
  __device__ __forceinline__ void start(lds_char* lds_, gbl_char* stream0, gbl_char* stream1, int passes0_,
                                        int passes_, int lane, int wave) {
    lds = lds_;
    lane16 = (uint32_t)lane * 16u;
    wave4k = (uint32_t)wave * 4096u;
    base[0] = stream0 + wave4k + lane16;
    base[1] = stream1 + wave4k + lane16;
    passes0 = passes0_;
    passes = passes_;
    pf_pass = 0;
    pf_left = STAGES_PER_PASS;
    pf_ptr = (passes0 > 0) ? base[0] : base[1];
    pf_slot = 0;
    rd_slot = 0;
    rd_addr = 0;
#pragma unroll
    for (int s = 0; s < PF_DIST; ++s) issue();
    // stage 0 must have landed before the first stage_begin() (which only certifies stage 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (PF_DIST - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
  }

  // Entering stage s: my pieces of stage s+1 have landed -> barrier -> everyone's have, and every
  // wave is done with stage s-1, whose slot is (far) behind the one refilled here.
  __device__ __forceinline__ void stage_begin() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (PF_DIST - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue();
    rd_addr = LDS_RING + rd_slot * STAGE_BYTES + lane16;
    rd_slot = (rd_slot + 1 == RING_SLOTS) ? 0u : rd_slot + 1;
  }

  __device__ __forceinline__ f32x4 read_frag(int i) const {
    return *(const __attribute__((address_space(3))) f32x4*)(lds + rd_addr + i * FRAG_BYTES);
  }
};

// acc[t][4q+j] = bias[32t + 8q + 4h + j]
template <int NT>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NT], const lds_float* bias, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const __attribute__((address_space(3))) f32x4*)(bias + 32 * t + 8 * q + 4 * h);
      acc[t][4 * q + 0] = b[0];
      acc[t][4 * q + 1] = b[1];
      acc[t][4 * q + 2] = b[2];
      acc[t][4 * q + 3] = b[3];
    }
  }
}

// k-groups [G0, G0+NG) of a layer with NT output tiles; B operands come from src (group g of the
// segment = registers 4(g%4)..+3 of tile g/4).
template <int NT, int G0, int NG, int NSRC>
__device__ __forceinline__ void mma_segment(WeightPipe& p, const f32x16 (&src)[NSRC], f32x16 (&acc)[NT]) {
  static_assert((NG + 3) / 4 <= NSRC, "source too small");
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int phi = (G0 + g) * NT + t;
      if ((phi % STAGE_FRAGS) == 0) p.stage_begin();
      const f32x4 a = p.read_frag(phi % STAGE_FRAGS);
      const int st = g >> 2, r0 = (g & 3) * 4;
      acc[t] = CRNERF_MFMA(a[0], src[st][r0 + 0], acc[t]);
      acc[t] = CRNERF_MFMA(a[1], src[st][r0 + 1], acc[t]);
      acc[t] = CRNERF_MFMA(a[2], src[st][r0 + 2], acc[t]);
      acc[t] = CRNERF_MFMA(a[3], src[st][r0 + 3], acc[t]);
    }
  }
}

template <int NT, int NDST>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NT], f32x16 (&act)[NDST], float floor_) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) act[t][r] = fmaxf(acc[t][r], floor_);
}

__device__ __forceinline__ float softplus_ref(float x) {
  // nn.Softplus(beta=1, threshold=20), models/nerf.py:146
  return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// One 32-point tile through one model.  pe/dv are the positional embeddings in B-operand order
// (posenc.h).  Returns feat[t][4q+j] = rgb feature 32t+8q+4h+j of point p, and sigma (both halves).
__device__ __forceinline__ void mlp_tile(WeightPipe& p, int model, const f32x16 (&pe)[3], const f32x16 (&dv)[1],
                                         f32x16 (&feat)[2], float& sigma, int h) {
  const lds_float* C = (const lds_float*)(p.lds + (model ? LDS_CONST1 : LDS_CONST0));
  const float NEG_INF = -__builtin_huge_valf();
  f32x16 act[8], acc[8];

  init_acc<8>(acc, C + C_BIAS, h);                       // xyz_encoding_1
  mma_segment<8, 0, G_XYZ>(p, pe, acc);
  store_act<8>(acc, act, 0.0f);
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {                          // xyz_encoding_2..4
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_segment<8, 0, G_HID>(p, act, acc);
    store_act<8>(acc, act, 0.0f);
  }
  init_acc<8>(acc, C + C_BIAS + 4 * W_HIDDEN, h);        // xyz_encoding_5 = Linear(cat[xyz, h])
  mma_segment<8, 0, G_XYZ>(p, pe, acc);
  mma_segment<8, G_XYZ, G_HID>(p, act, acc);
  store_act<8>(acc, act, 0.0f);
#pragma unroll 1
  for (int l = 5; l < 8; ++l) {                          // xyz_encoding_6..8
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_segment<8, 0, G_HID>(p, act, acc);
    store_act<8>(acc, act, 0.0f);
  }
  {                                                      // static_sigma: 256 -> 1 on the VALU
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *(const __attribute__((address_space(3))) f32x4*)(C + C_WSIG + 32 * t + 8 * q + 4 * h);
        s = fmaf(w[0], act[t][4 * q + 0], s);
        s = fmaf(w[1], act[t][4 * q + 1], s);
        s = fmaf(w[2], act[t][4 * q + 2], s);
        s = fmaf(w[3], act[t][4 * q + 3], s);
      }
    s += __shfl_xor(s, 32);
    sigma = softplus_ref(s + C[C_BSIG]);
  }
  init_acc<8>(acc, C + C_BFIN, h);                       // xyz_encoding_final (no activation)
  mma_segment<8, 0, G_HID>(p, act, acc);
  store_act<8>(acc, act, NEG_INF);
  {
    f32x16 acc4[4];                                      // dir_encoding = relu(Linear(cat[final, dir]))
    init_acc<4>(acc4, C + C_BDIR, h);
    mma_segment<4, 0, G_HID>(p, act, acc4);
    mma_segment<4, G_HID, G_DIR>(p, dv, acc4);
    store_act<4>(acc4, act, 0.0f);
  }
  {
    f32x16 acc2[2];                                      // static_rgb = sigmoid(Linear)
    init_acc<2>(acc2, C + C_BRGB, h);
    mma_segment<2, 0, G_HALF>(p, act, acc2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) feat[t][r] = sigmoid_ref(acc2[t][r]);
  }
}

// Copy both models' consts blocks into LDS (all threads of a 256-thread workgroup).
__device__ __forceinline__ void load_consts(lds_char* lds, const char* packed0, const char* packed1) {
  const int tid = threadIdx.x;
  for (int i = tid; i < CONST_BYTES / 16; i += blockDim.x) {
    const f32x4 a = *(const f32x4*)(packed0 + i * 16);
    const f32x4 b = *(const f32x4*)(packed1 + i * 16);
    *(__attribute__((address_space(3))) f32x4*)(lds + LDS_CONST0 + i * 16) = a;
    *(__attribute__((address_space(3))) f32x4*)(lds + LDS_CONST1 + i * 16) = b;
  }
  __syncthreads();
}

}  // namespace crnerf
