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

constexpr int RING_SLOTS = 6;   // stages c-1 (being refilled), c (read), c+1 (read ahead), c+2 (certified), c+3, c+4 (in flight)
static_assert(RING_SLOTS >= 5, "ring too shallow for the read-ahead protocol");

// LDS map: byte offsets inside the single dynamic __shared__ array of every kernel using the core
constexpr int LDS_CONST0 = 0;
constexpr int LDS_CONST1 = CONST_BYTES;
constexpr int LDS_RING = 2 * CONST_BYTES;
constexpr int LDS_SCRATCH = LDS_RING + RING_SLOTS * STAGE_BYTES;  // 120,832

// Phase timer for tuning builds (-DCRNERF_TIMING): wave 0 of block 0 accumulates shader-clock cycles
// per phase into crnerf_timing[]; compiled out otherwise.
enum { T_PROLOGUE = 0, T_MMA, T_EPILOGUE, T_SIGMA, T_COMPOSITE, T_RAYLEVEL, T_TOTAL, T_X0, T_X1, T_X2, T_X3, T_X4, T_X5, T_X6, T_X7, T_REAL, T_COUNT };   // T_REAL: 100 MHz ticks from start() to flush()
#ifdef CRNERF_TIMING
static __device__ unsigned long long crnerf_timing[T_COUNT];   // one copy per translation unit
struct PhaseTimer {
  unsigned long long last, acc[T_COUNT], real0;
  bool on;
  __device__ __forceinline__ void start(bool on_) {
    on = on_;
    for (int i = 0; i < T_COUNT; ++i) acc[i] = 0;
    real0 = __builtin_amdgcn_s_memrealtime();
    last = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void tick(int phase) {
    __builtin_amdgcn_sched_barrier(0);   // the compiler may otherwise move MFMAs / VALU work across the clock read
    const unsigned long long t = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    acc[phase] += t - last;
    acc[T_TOTAL] += t - last;
    last = t;
  }
  __device__ __forceinline__ void flush() {
    acc[T_REAL] = __builtin_amdgcn_s_memrealtime() - real0;
    if (on)
      for (int i = 0; i < T_COUNT; ++i) crnerf_timing[i] = acc[i];
  }
};
#else
struct PhaseTimer {
  __device__ __forceinline__ void start(bool) {}
  __device__ __forceinline__ void tick(int) {}
  __device__ __forceinline__ void flush() {}
};
#endif

#define CRNERF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// One LDS-DMA piece: lanes copy 16 B each, global (base + voff + imm) -> LDS (lds_addr + imm + 16 * lane).  Written as
// asm on purpose: with the builtin, hipcc's waitcnt model stops counting LDS reads across an LDS-DMA instruction and
// the next use of ANY prefetched fragment becomes s_waitcnt lgkmcnt(0) -- the read-ahead ring is drained every fourth
// k-step.  (vmcnt for these loads is counted by hand in start()/advance() anyway.)  The asm rewrites M0 without telling
// the compiler (M0 is a reserved register, clang rejects it as a clobber): kernels that contain it must not index
// register arrays dynamically (s_set_gpr_idx / v_movrel keep their index in M0) -- checked by grepping the ISA.
__device__ __forceinline__ void glds16(uint32_t lds_addr, const char* base, uint32_t voff, int imm) {
#ifdef CRNERF_EXP_GLOADONLY   // (timing experiments only; garbage) the same VMEM request into a dummy register: issue + L2 traffic, no LDS write
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  u4 dummy;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dummy) : "v"(voff), "s"(base), "n"(imm) : "memory");
  return;
#endif
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds_addr), "v"(voff), "s"(base), "n"(imm)
               : "memory");
}


// Streams packed weights into the LDS ring.  All members except pf_ptr / rd_addr are wave-uniform.
//
// Protocol (c = stage being consumed):  fragments of stage c AND of stage c+1 may be read (the
// latter so that the first fragments of the next stage are already in registers when the barrier
// falls -- no LDS latency is exposed at stage boundaries).  advance() moves c -> c+1; it must be
// called after the last read of stage c has been issued: it waits until this wave's pieces of stage
// c+2 have landed (counted vmcnt, never 0) and barriers (=> everyone's have, and nobody still reads
// stage c-1).  Stage c-1's slot is then refilled with stage c+RING_SLOTS-1 by four issue_piece()
// calls spread over the consumption of stage c+1.
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
  uint32_t rd_addr;    // per-lane LDS byte address of fragment 0 of stage c
  uint32_t lane16;
  uint32_t wave4k;

  // One of the 4 LDS-DMA pieces (1 KiB each) this wave contributes to the stage being fetched.  A
  // global_load_lds costs ~60-180 issue cycles (MI355X guide), so the pieces are issued one at a
  // time between MFMA groups (mma_layer) instead of back to back behind the barrier.
  __device__ __forceinline__ void issue_piece(int i) {
    lds_char* dst = lds + LDS_RING + pf_slot * STAGE_BYTES + wave4k;
#ifndef CRNERF_EXP_NOGLDS   // (timing experiments only: results are garbage without the loads)
    __builtin_amdgcn_global_load_lds(pf_ptr + i * FRAG_BYTES, dst + i * FRAG_BYTES, 16, 0, 0);
#endif
    if (i == 3) {
      pf_slot = (pf_slot + 1 == RING_SLOTS) ? 0u : pf_slot + 1;
      pf_ptr += STAGE_BYTES;
      if (--pf_left == 0) {
        pf_left = STAGES_PER_PASS;
        pf_pass = (pf_pass + 1 == passes) ? 0 : pf_pass + 1;
        pf_ptr = (pf_pass < passes0) ? base[0] : base[1];
      }
    }
  }
  __device__ __forceinline__ void issue() {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece(i);
  }

  // Call once, all waves.  On return stages 0 and 1 are readable.
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
    rd_addr = LDS_RING + lane16;
#pragma unroll
    for (int s = 0; s < RING_SLOTS - 1; ++s) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING_SLOTS - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
  }

  __device__ __forceinline__ uint32_t next_addr() const {
    const uint32_t n = (rd_slot + 1 == RING_SLOTS) ? 0u : rd_slot + 1;
    return LDS_RING + n * STAGE_BYTES + lane16;
  }

  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING_SLOTS - 4)) : "memory");
#ifndef CRNERF_EXP_NOBARRIER  // (timing experiments only)
    __builtin_amdgcn_s_barrier();
#endif
    rd_slot = (rd_slot + 1 == RING_SLOTS) ? 0u : rd_slot + 1;
    rd_addr = LDS_RING + rd_slot * STAGE_BYTES + lane16;
  }

  __device__ __forceinline__ f32x4 read_at(uint32_t addr, int frag) const {
    return *(const __attribute__((address_space(3))) f32x4*)(lds + addr + frag * FRAG_BYTES);
  }

  // first k-group (8 fragments) of stage c into cur -- once, before the first layer
  __device__ __forceinline__ void prime(f32x4 (&cur)[8]) const {
#pragma unroll
    for (int t = 0; t < 8; ++t) cur[t] = read_at(rd_addr, t);
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

// One layer: NT output tiles, k-groups 0..NGA-1 with B operands from srcA then NGB groups from srcB
// (group g of a source = registers 4(g%4)..+3 of its tile g/4).  Software-pipelined by k-group:
// on entry cur[0..NT) holds the fragments of group 0; while group g's MFMAs run, group g+1's
// fragments are loaded (from the next stage when g+1 starts one); after the last group the first
// group of the NEXT layer (NT_NEXT fragments, always the head of the next stage) is left in cur.
template <int NT, int NGA, int NGB, int NT_NEXT, int NA, int NB>
__device__ __forceinline__ void mma_layer(WeightPipe& p, const f32x16 (&srcA)[NA], const f32x16 (&srcB)[NB],
                                          f32x16 (&acc)[NT], f32x4 (&cur)[8]) {
  static_assert((NGA + 3) / 4 <= NA && (NGB + 3) / 4 <= NB, "source too small");
  static_assert(STAGE_FRAGS % NT == 0 && ((NGA + NGB) * NT) % STAGE_FRAGS == 0, "layer must be whole stages");
  constexpr int NG = NGA + NGB;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const bool last = (g == NG - 1);
    const int nphi = ((g + 1) * NT) % STAGE_FRAGS;     // slot of the next group's first fragment in its stage
    const bool new_stage = last || nphi == 0;
    const uint32_t base = new_stage ? p.next_addr() : p.rd_addr;
    const int cnt = last ? NT_NEXT : NT;
    f32x4 nxt[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < cnt) nxt[t] = p.read_at(base, (last ? 0 : nphi) + t);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int slot = (g * NT + t) % STAGE_FRAGS;       // this fragment's position in its stage
      if (slot % 4 == 0) p.issue_piece(slot / 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b = (g < NGA) ? srcA[g >> 2][(g & 3) * 4 + j] : srcB[(g - NGA) >> 2][((g - NGA) & 3) * 4 + j];
        acc[t] = CRNERF_MFMA(cur[t][j], b, acc[t]);
      }
    }
    if (new_stage) p.advance();
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < cnt) cur[t] = nxt[t];
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
// (posenc.h); cur carries the prefetched head fragments between layers/passes.  Returns feat[t][4q+j] = rgb feature 32t+8q+4h+j of point p, and sigma (both halves).
__device__ __forceinline__ void mlp_tile(WeightPipe& p, int model, const f32x16 (&pe)[3], const f32x16 (&dv)[1],
                                         f32x16 (&feat)[2], float& sigma, int h, f32x4 (&cur)[8], PhaseTimer& tm) {
  const lds_float* C = (const lds_float*)(p.lds + (model ? LDS_CONST1 : LDS_CONST0));
  const float NEG_INF = -__builtin_huge_valf();
  f32x16 act[8], acc[8];
  tm.tick(T_PROLOGUE);

  init_acc<8>(acc, C + C_BIAS, h);                       // xyz_encoding_1
  tm.tick(T_EPILOGUE);
  mma_layer<8, G_XYZ, 0, 8>(p, pe, pe, acc, cur);
  tm.tick(T_MMA);
  store_act<8>(acc, act, 0.0f);
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {                          // xyz_encoding_2..4
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    tm.tick(T_EPILOGUE);
    mma_layer<8, G_HID, 0, 8>(p, act, act, acc, cur);
    tm.tick(T_MMA);
    store_act<8>(acc, act, 0.0f);
  }
  init_acc<8>(acc, C + C_BIAS + 4 * W_HIDDEN, h);        // xyz_encoding_5 = Linear(cat[xyz, h])
  tm.tick(T_EPILOGUE);
  mma_layer<8, G_XYZ, G_HID, 8>(p, pe, act, acc, cur);
  tm.tick(T_MMA);
  store_act<8>(acc, act, 0.0f);
#pragma unroll 1
  for (int l = 5; l < 8; ++l) {                          // xyz_encoding_6..8
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    tm.tick(T_EPILOGUE);
    mma_layer<8, G_HID, 0, 8>(p, act, act, acc, cur);
    tm.tick(T_MMA);
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
    tm.tick(T_SIGMA);
  }
  init_acc<8>(acc, C + C_BFIN, h);                       // xyz_encoding_final (no activation)
  tm.tick(T_EPILOGUE);
  mma_layer<8, G_HID, 0, 4>(p, act, act, acc, cur);
  tm.tick(T_MMA);
  store_act<8>(acc, act, NEG_INF);
  {
    f32x16 acc4[4];                                      // dir_encoding = relu(Linear(cat[final, dir]))
    init_acc<4>(acc4, C + C_BDIR, h);
    tm.tick(T_EPILOGUE);
    mma_layer<4, G_HID, G_DIR, 2>(p, act, dv, acc4, cur);
    tm.tick(T_MMA);
    store_act<4>(acc4, act, 0.0f);
  }
  {
    f32x16 acc2[2];                                      // static_rgb = sigmoid(Linear)
    init_acc<2>(acc2, C + C_BRGB, h);
    tm.tick(T_EPILOGUE);
    mma_layer<2, G_HALF, 0, 8>(p, act, act, acc2, cur);
    tm.tick(T_MMA);  // leaves the next pass's layer-1 head in cur
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) feat[t][r] = sigmoid_ref(acc2[t][r]);
    tm.tick(T_EPILOGUE);
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
