// NeRF_sigma forward, "v16" core: one 16-point tile per wavefront on v_mfma_f32_16x16x4_f32, TWO
// wavefronts per SIMD (8 per workgroup, <= 256 registers each).
//
// Why a second core (measured on the 32x32x2 / one-wave-per-SIMD core, profiles/README.md): an
// in-order wave cannot hide its own non-MFMA issue time -- the ~66-cycle issue cost of every
// global_load_lds piece (6.4 %), the accumulator->operand epilogues (3.4 %), posenc / sigma /
// compositing (~3 %) all leave the SIMD's matrix pipe idle.  With two resident waves the partner's
// MFMAs fill those slots.  Same swapped-operand scheme, same fragment stream sizes:
//   lane (p = lane&15, g = lane>>4) owns point p; D[feature][point] leaves it holding features
//   16T + 4g + r (register r of tile T), which is the k-value it supplies as B operand of k-step
//   (T, r) of the next layer.  A fragment (one ds_read_b128) = frag(u, T)[lane=16kq+i][r] =
//   W[16T+i][16u+4kq+r] feeds the four MFMAs r = 0..3 of k-group u into acc[T].
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"
#include "sincos_pow2.h"
#include "mlp_core.h"   // typedefs, LDS map, PhaseTimer, softplus/sigmoid helpers

namespace crnerf {

#define CRNERF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int V16_WAVES = 8;
constexpr int V16_PIECES = STAGE_FRAGS / V16_WAVES;   // LDS-DMA pieces per wave per stage (2)
constexpr int V16_AHEAD = 4;                          // fragments read ahead of use (crosses stage barriers)

// k-groups (16 k-values) and output tiles (16 rows) per layer
constexpr int U_XYZ = XYZ_PAD / 16;   // 6
constexpr int U_HID = W_HIDDEN / 16;  // 16
constexpr int U_DIR = DIR_PAD / 16;   // 2
constexpr int U_HALF = 128 / 16;      // 8

// Same ring protocol as WeightPipe (mlp_core.h), 8 waves x 2 pieces per stage.
struct WeightPipe16 {
  lds_char* lds;
  const char* base[2];   // wave-uniform: packed streams + this wave's 2 KiB column (the lane offset rides in the VGPR-offset operand)
  const char* pf_ptr;
  int pf_left, pf_pass, passes0, passes;
  int stages_per_pass = STAGES_PER_PASS;   // 151 forward stream, 138 transposed (backward-data) stream
  uint32_t pf_slot, rd_slot, rd_addr, lane16, wave2k, lds_ring;
  int stagger;          // 0/1: which of the two candidate slot sets this wave uses for its LDS-DMA issue

  __device__ __forceinline__ void issue_piece(int i) {
    // asm, not the builtin: see glds16 (mlp_core.h) -- with the builtin every fragment prefetch is drained
    // (s_waitcnt lgkmcnt(0)) at the first use after each LDS-DMA instruction
    const uint32_t dst = lds_ring + pf_slot * STAGE_BYTES;
    if (i == 0) glds16(dst, pf_ptr, lane16, 0);
    else glds16(dst, pf_ptr, lane16, FRAG_BYTES);
    if (i == V16_PIECES - 1) {
      pf_slot = (pf_slot + 1 == RING_SLOTS) ? 0u : pf_slot + 1;
      pf_ptr += STAGE_BYTES;
      if (--pf_left == 0) {
        pf_left = stages_per_pass;
        pf_pass = (pf_pass + 1 == passes) ? 0 : pf_pass + 1;
        pf_ptr = (pf_pass < passes0) ? base[0] : base[1];
      }
    }
  }

  __device__ __forceinline__ void start(lds_char* lds_, const char* stream0, const char* stream1, int passes0_, int passes_,
                                        int lane, int wave) {
    static_assert(V16_PIECES == 2, "issue_piece hard-codes the two instruction offsets");
    lds = lds_;
    lane16 = (uint32_t)lane * 16u;
    wave2k = (uint32_t)wave * (V16_PIECES * FRAG_BYTES);
    lds_ring = (uint32_t)(uintptr_t)lds_ + LDS_RING + wave2k;
    stagger = (wave >> 2) & 1;   // waves w and w+4 of a 512-thread workgroup share a SIMD
    base[0] = stream0 + wave2k;
    base[1] = stream1 + wave2k;
    passes0 = passes0_;
    passes = passes_;
    pf_pass = 0;
    pf_left = stages_per_pass;
    pf_ptr = (passes0 > 0) ? base[0] : base[1];
    pf_slot = 0;
    rd_slot = 0;
    rd_addr = LDS_RING + lane16;
#pragma unroll
    for (int s = 0; s < RING_SLOTS - 1; ++s)
#pragma unroll
      for (int i = 0; i < V16_PIECES; ++i) issue_piece(i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V16_PIECES * (RING_SLOTS - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
  }

  __device__ __forceinline__ uint32_t next_addr() const {
    const uint32_t n = (rd_slot + 1 == RING_SLOTS) ? 0u : rd_slot + 1;
    return LDS_RING + n * STAGE_BYTES + lane16;
  }

  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V16_PIECES * (RING_SLOTS - 4)) : "memory");
    __builtin_amdgcn_s_barrier();
    rd_slot = (rd_slot + 1 == RING_SLOTS) ? 0u : rd_slot + 1;
    rd_addr = LDS_RING + rd_slot * STAGE_BYTES + lane16;
  }

  // fragment at stream position (slot-in-stage) s of the current stage; s >= 16 reads ahead into the next stage
  __device__ __forceinline__ f32x4 read_slot(int s) const {
    const uint32_t a = (s < STAGE_FRAGS) ? rd_addr + s * FRAG_BYTES : next_addr() + (s - STAGE_FRAGS) * FRAG_BYTES;
    return *(const __attribute__((address_space(3))) f32x4*)(lds + a);
  }

  __device__ __forceinline__ void prime(f32x4 (&q)[V16_AHEAD]) const {
#pragma unroll
    for (int i = 0; i < V16_AHEAD; ++i) q[i] = read_slot(i);
  }
};

// acc[T][r] = bias[16T + 4g + r]
template <int NT>
__device__ __forceinline__ void init_acc16(f32x4 (&acc)[NT], const lds_float* bias, int g) {
#pragma unroll
  for (int T = 0; T < NT; ++T) acc[T] = *(const __attribute__((address_space(3))) f32x4*)(bias + 16 * T + 4 * g);
}

// One layer: NT output tiles of 16 rows; k-groups 0..NGA-1 take their B operands from srcA[u],
// the following NGB groups from srcB[u-NGA].  Fragments are consumed in stream order (group-major,
// tile-minor); q[] always holds the next V16_AHEAD fragments of the STREAM (it runs on into the next
// layer / pass), so no LDS latency is exposed at stage or layer boundaries.
// Deferred stores of the PREVIOUS layer's output (training twins, mlp_train16.h ActSaver): nothing at inference.
struct NoDeferred {
  static constexpr int pieces = 0;
  __device__ __forceinline__ void piece(int) const {}
};
template <class SAVE, int ND>
struct DeferredActs {   // the ND tiles of `a` go to `row`, one piece per call
  static constexpr int pieces = ND;
  const SAVE& save;
  float* row;
  const f32x4 (&a)[16];
  __device__ __forceinline__ void piece(int T) const { save.piece(row, T, a[T]); }
};

template <int NT, int NGA, int NGB, int NA, int NB, class DEF = NoDeferred>
__device__ __forceinline__ void mma_layer16(WeightPipe16& p, const f32x4 (&srcA)[NA], const f32x4 (&srcB)[NB], f32x4 (&acc)[NT],
                                            f32x4 (&q)[V16_AHEAD], const DEF& def = DEF()) {
  static_assert(NGA <= NA && NGB <= NB, "source too small");
  constexpr int NF = (NGA + NGB) * NT;
  static_assert(NF % STAGE_FRAGS == 0 && NT % 2 == 0, "layer must be whole stages");
  constexpr int STEPS = NF / 2;                                        // fragment-pair steps of this layer (8 MFMAs each)
  constexpr int STRIDE = DEF::pieces > 0 ? (STEPS / DEF::pieces >= 4 ? 4 : (STEPS / DEF::pieces >= 2 ? 2 : 1)) : 1;
  static_assert(DEF::pieces == 0 || STRIDE * DEF::pieces <= STEPS, "not enough steps to carry the deferred stores");
#pragma unroll
  for (int u = 0; u < NGA + NGB; ++u) {
#pragma unroll
    for (int T = 0; T < NT; T += 2) {   // two fragments (tiles T, T+1 of k-group u) per step: two independent MFMA chains
      const int f = u * NT + T;
      const int s = f % STAGE_FRAGS;
      // the two waves sharing a SIMD issue their LDS-DMA pieces at different fragment slots, so one's
      // ~66-cycle issue stall is covered by the other's MFMAs
      if (s % 4 == 0 && ((s >> 2) & 1) == p.stagger) p.issue_piece(s >> 3);
      const f32x4 a0 = q[f % V16_AHEAD], a1 = q[(f + 1) % V16_AHEAD];
      q[f % V16_AHEAD] = p.read_slot(s + V16_AHEAD);
      q[(f + 1) % V16_AHEAD] = p.read_slot(s + 1 + V16_AHEAD);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b = (u < NGA) ? srcA[u < NGA ? u : 0][r] : srcB[u < NGA ? 0 : u - NGA][r];
        acc[T] = CRNERF_MFMA16(a0[r], b, acc[T]);
        acc[T + 1] = CRNERF_MFMA16(a1[r], b, acc[T + 1]);
      }
      if (s == STAGE_FRAGS - 2) p.advance();
      if (DEF::pieces > 0) {          // one deferred store every STRIDE steps (>= 8 x 64 cycles of store data path between a wave's stores)
        const int step = f / 2;
        if (step % STRIDE == 0 && step / STRIDE < DEF::pieces) def.piece(step / STRIDE);
      }
    }
  }
}

template <int NT, bool RELU, int NDST>
__device__ __forceinline__ void store_act16(const f32x4 (&acc)[NT], f32x4 (&act)[NDST]) {
#pragma unroll
  for (int T = 0; T < NT; ++T)
#pragma unroll
    for (int r = 0; r < 4; ++r) act[T][r] = RELU ? fmaxf(acc[T][r], 0.0f) : acc[T][r];
}

// One 16-point tile through one model.  pe[6] / dv[2]: embeddings in B-operand order (posenc16.h).
// feat[T][r] = rgb feature 16T + 4g + r of point p (T = 0..3); sigma valid in all lanes.
struct NoSave {   // inference: nothing is materialised
  template <int NT>
  __device__ __forceinline__ void operator()(int, const f32x4 (&)[NT]) const {}
  template <int NT>
  __device__ __forceinline__ float* begin(int, const f32x4 (&)[NT]) const { return nullptr; }
  __device__ __forceinline__ void piece(float*, int, const f32x4&) const {}
};

// Where the direction embedding comes from when the dir layer needs it (late in the tile): registers
// (module entry: it differs per point) or 8 floats per lane group parked in LDS (fused renderer: it is
// a per-ray constant, and 8 fewer live registers across ten layers is what keeps the kernel spill-free).
struct DirRegs {
  f32x4 v[2];
  __device__ __forceinline__ void get(f32x4 (&o)[2]) const { o[0] = v[0]; o[1] = v[1]; }
};
struct DirLds {
  const lds_float* p;   // this lane group's 8 floats, 16-byte aligned
  __device__ __forceinline__ void get(f32x4 (&o)[2]) const {
    o[0] = *(const __attribute__((address_space(3))) f32x4*)(p);
    o[1] = *(const __attribute__((address_space(3))) f32x4*)(p + 4);
  }
};

template <class DIR, class SAVE = NoSave>
__device__ __forceinline__ void mlp_tile16(WeightPipe16& p, int model, const f32x4 (&pe)[6], const DIR& dir,
                                           f32x4 (&feat)[4], float& sigma, int g, f32x4 (&q)[V16_AHEAD], PhaseTimer& tm,
                                           const SAVE& save = SAVE()) {
  const lds_float* C = (const lds_float*)(p.lds + (model ? LDS_CONST1 : LDS_CONST0));
  f32x4 act[16], acc[16];
  tm.tick(T_PROLOGUE);

  // Training twins: the output of layer l is stored BEHIND the MFMAs of layer l + 1 (DeferredActs, see ActSaver in mlp_train16.h);
  // `row` is the row pointer save.begin() returned for the slot whose pieces are pending.
  typedef DeferredActs<SAVE, 16> Def16;
  init_acc16<16>(acc, C + C_BIAS, g);                    // xyz_encoding_1
  mma_layer16<16, U_XYZ, 0>(p, pe, pe, acc, q);
  tm.tick(T_MMA);
  store_act16<16, true>(acc, act);
  float* row = save.begin(0, act);
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {                          // xyz_encoding_2..4
    init_acc16<16>(acc, C + C_BIAS + l * W_HIDDEN, g);
    tm.tick(T_EPILOGUE);
    mma_layer16<16, U_HID, 0>(p, act, act, acc, q, Def16{save, row, act});
    tm.tick(T_MMA);
    store_act16<16, true>(acc, act);
    row = save.begin(l, act);
  }
  init_acc16<16>(acc, C + C_BIAS + 4 * W_HIDDEN, g);     // xyz_encoding_5 = Linear(cat[xyz, h])
  tm.tick(T_EPILOGUE);
  mma_layer16<16, U_XYZ, U_HID>(p, pe, act, acc, q, Def16{save, row, act});
  tm.tick(T_MMA);
  store_act16<16, true>(acc, act);
  row = save.begin(4, act);
#pragma unroll 1
  for (int l = 5; l < 8; ++l) {                          // xyz_encoding_6..8
    init_acc16<16>(acc, C + C_BIAS + l * W_HIDDEN, g);
    tm.tick(T_EPILOGUE);
    mma_layer16<16, U_HID, 0>(p, act, act, acc, q, Def16{save, row, act});
    tm.tick(T_MMA);
    store_act16<16, true>(acc, act);
    row = save.begin(l, act);
  }
  {                                                      // static_sigma: 256 -> 1 on the VALU
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int T = 0; T < 16; T += 2) {
      const f32x4 w0 = *(const __attribute__((address_space(3))) f32x4*)(C + C_WSIG + 16 * T + 4 * g);
      const f32x4 w1 = *(const __attribute__((address_space(3))) f32x4*)(C + C_WSIG + 16 * (T + 1) + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0 = fmaf(w0[r], act[T][r], s0);
        s1 = fmaf(w1[r], act[T + 1][r], s1);
      }
    }
    float s = s0 + s1;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    sigma = softplus_ref(s + C[C_BSIG]);
  }
  tm.tick(T_SIGMA);
  init_acc16<16>(acc, C + C_BFIN, g);                    // xyz_encoding_final (no activation)
  mma_layer16<16, U_HID, 0>(p, act, act, acc, q, Def16{save, row, act});
  tm.tick(T_MMA);
  store_act16<16, false>(acc, act);
  row = save.begin(8, act);
  {
    f32x4 acc8[8], dv[2];                                // dir_encoding = relu(Linear(cat[final, dir]))
    dir.get(dv);
    init_acc16<8>(acc8, C + C_BDIR, g);
    tm.tick(T_EPILOGUE);
    mma_layer16<8, U_HID, U_DIR>(p, act, dv, acc8, q, Def16{save, row, act});
    tm.tick(T_MMA);
    store_act16<8, true>(acc8, act);
    row = save.begin(9, act);                            // only tiles 0..7 (128 features) are meaningful
  }
  {
    f32x4 acc4[4];                                       // static_rgb = sigmoid(Linear)
    init_acc16<4>(acc4, C + C_BRGB, g);
    tm.tick(T_EPILOGUE);
    mma_layer16<4, U_HALF, 0>(p, act, act, acc4, q, DeferredActs<SAVE, 8>{save, row, act});
    tm.tick(T_MMA);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) feat[T][r] = sigmoid_ref(acc4[T][r]);
  }
  tm.tick(T_EPILOGUE);
}

// Embedding of (x,y,z) straight into B-operand registers, v16 slot order (layout.h): lane group g
// computes argument a = 8v + 2g + pr for v = 0..NV-1, pr = 0,1 and keeps (sin, cos) in registers
// (2pr, 2pr+1) of k-group v.
template <int F, int NV>
__device__ __forceinline__ void posenc_regs16(float x, float y, float z, int g, f32x4 (&out)[NV]) {
  const Rev2Pi rx = to_rev2pi(x), ry = to_rev2pi(y), rz = to_rev2pi(z);   // sincos_pow2.h
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int a_lo = 8 * v + pr;                         // argument for g = 0; lane's is a_lo + 2g
      const int a = a_lo + 2 * g;
      float e0, e1;
      if (a_lo + 6 < 3 * F) {                              // every lane group has a trig argument here
        const int f = a / 3, d = a - 3 * f;
        const Rev2Pi val = {d == 0 ? rx.p : (d == 1 ? ry.p : rz.p), d == 0 ? rx.e : (d == 1 ? ry.e : rz.e)};
        float s, c;
        sincos_rev2pi(val, f, s, c);
        e0 = s; e1 = c;
      } else {                                             // mixed: some groups hold (x,y) / (z,0) / pad
        const bool trig = a < 3 * F;
        const int ac = trig ? a : 0;
        const int f = ac / 3, d = ac - 3 * f;
        const Rev2Pi val = {d == 0 ? rx.p : (d == 1 ? ry.p : rz.p), d == 0 ? rx.e : (d == 1 ? ry.e : rz.e)};
        float s, c;
        sincos_rev2pi(val, f, s, c);
        e0 = trig ? s : (a == 3 * F ? x : (a == 3 * F + 1 ? z : 0.0f));
        e1 = trig ? c : (a == 3 * F ? y : 0.0f);
      }
      out[v][2 * pr + 0] = e0;
      out[v][2 * pr + 1] = e1;
    }
}

}  // namespace crnerf
