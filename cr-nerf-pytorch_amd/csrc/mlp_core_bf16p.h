// NeRF_sigma forward on the bf16 matrix cores, "pair" core: one 32-point tile per wavefront, EIGHT wavefronts per
// workgroup = two per SIMD (<= 256 registers each).  Same arithmetic, same packed fragment stream and same k order as
// mlp_core_bf16.h (bit-identical MLP outputs); what changes is the work decomposition.
//
// Why a second bf16 core (round-2 counters of the one-wave-per-SIMD core): the matrix pipe was busy 65 % --
// an in-order wave cannot hide its own non-MFMA issue time: the per-ray phases (embeddings, compositing, sample_pdf,
// merge: 12 % of the cycles) and the 36.2-instead-of-32 cycles per MFMA inside the MLP (every fragment read, LDS-DMA
// piece, bias read and epilogue instruction is an issue slot of the ONLY wave on the SIMD).  With two resident waves
// each MFMA of one wave leaves 64 cycles of issue slots to its own fillers, the partner's MFMAs fill the slots a wave
// loses to an LDS-DMA issue or a barrier, and the per-ray phases issue from two waves at once.
//
//   * lane (p = lane&31, h = lane>>5) holds point p; the 32x32 C/D layout leaves it with features
//     32T + 8(r>>2) + 4h + (r&3) of that point in accumulator register r, and relu + v_cvt_pk_bf16_f32 of registers
//     8j..8j+7 IS the B operand of k-step 2T+j of the next layer (layout.h "fragB").  256 activations of 32 points
//     = 64 VGPRs, ping-pong 128.
//   * tile-outer / k-inner: one accumulator tile (16 registers) is live plus the previous tile's, whose epilogue
//     (8 quarters of 2 VALU) runs behind this tile's k-steps 1..8.  The bias is read from LDS straight into the
//     accumulator of the next tile once its epilogue has drained it.
//   * every A fragment feeds ONE MFMA per wave: 8 waves x 1 KiB per 64 matrix-pipe cycles per SIMD pair
//     = 128 B/clk/CU of ds_read_b128 (half the LDS rate of the guide's table).
//   * weight ring: the 4-slot x 16 KiB static ring of WeightPipeB; a wave contributes TWO 1 KiB LDS-DMA pieces per
//     stage.
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182 (mixed precision as stated in include/crnerf.h).
#pragma once
#include <hip/hip_runtime.h>
#include "mlp_core_bf16.h"

namespace crnerf {

constexpr int P_WAVES = 8;
constexpr int P_PIECES = STAGE_FRAGS / P_WAVES;             // LDS-DMA pieces per wave and stage: 2
constexpr int LDS_SCRATCH_P = LDS_RING + B_RING * STAGE_BYTES;   // the 4-slot ring ends here (88,064)
static_assert(P_PIECES == 2, "issue_piece hard-codes two pieces");

// ---- compile-time schedule over the pass-relative fragment index i -------------------------------------------------
// Barrier slots as in mlp_core_bf16.h (B_ADV_SLOT = 11, tail 3: in front of the first look-ahead read of the next stage).
// Piece slots: stage slots 1 and 5 (tail stage: 0 and 2); the alternative set 3 and 7 (1 and 3) is kept for experiments
// (-DCRNERF_P_STAGGER=1).  Every piece of stage c + 3 is issued before the barrier in stage c, so the barrier waits
// vmcnt(4): stage c + 2 and stage c + 3 of this wave may be in flight, stage c + 1 has landed.
#ifndef CRNERF_P_STAGGER
#define CRNERF_P_STAGGER 0
#endif
constexpr int P_STAGGER = CRNERF_P_STAGGER;   // compile-time: a per-wave (runtime) choice would put a branch in every k-step and keep hipcc
                                              // from unrolling the tile loop, on which every ring constant depends
constexpr int p_piece_at(int i, int stagger) {
  const int sl = i % STAGE_FRAGS;
  if (!b_tail(i)) return sl == 1 + 2 * stagger ? 0 : (sl == 5 + 2 * stagger ? 1 : -1);
  return sl == stagger ? 0 : (sl == 2 + stagger ? 1 : -1);
}
constexpr bool p_schedule_ok() {
  for (int st = 0; st < 2; ++st) {
    int pieces = 0, advances = 0, cursors = 0;
    for (int i = 0; i < STREAMB_USED; ++i) {
      if (b_cur_stage(i) != advances) return false;
      const int rs = b_pos(i + B_AHEAD) / STAGE_FRAGS;
      if (rs > advances || rs < i / STAGE_FRAGS) return false;   // EVERY read targets a stage whose barrier has been passed
      if (p_piece_at(i, st) >= 0) {
        if (p_piece_at(i, st) != pieces % P_PIECES || cursors != pieces / P_PIECES) return false;
        if (b_cur_stage(i) != i / STAGE_FRAGS) return false;      // pieces of stage c + 3 are issued before stage c's barrier
        ++pieces;
      }
      if (b_advance_at(i)) {
        ++advances;
        if (pieces != P_PIECES * advances) return false;          // vmcnt(4) at the barrier counts on both pieces being out
      }
      if (b_cursor_at(i)) {
        ++cursors;
        if (pieces != P_PIECES * cursors) return false;
      }
    }
    if (advances != STAGESB_PER_PASS || cursors != STAGESB_PER_PASS) return false;
  }
  return true;
}
static_assert(p_schedule_ok(), "pair-core fragment schedule violates the ring protocol");

struct WeightPipeP {
  const char* base[2];   // packed streams + this wave's 2 KiB column (scalar)
  const char* cur;       // stream of the tile being multiplied
  const char* nxt;       // stream of the next tile
  uint32_t m0s[B_RING];  // scalar: LDS byte address of slot k + this wave's 2 KiB column
  uint32_t voff;         // per-lane: lane16 + STAGE_BYTES * (stage being fetched, tile-relative)
  uint32_t rd_base;      // per-lane: LDS_RING + lane16
  uint32_t lane16;
  lds_char* lds;

  __device__ __forceinline__ void issue_piece(int i, int F) {   // both constant after unrolling
    const char* src = F < STAGESB_PER_PASS ? cur : nxt;
#ifndef CRNERF_EXP_NOGLDS   // (energy / timing experiments only: results are garbage without the loads; tools/bf16_energy_probe.py)
    if (i == 0) glds16(m0s[F % B_RING], src, voff, 0);            // writes M0; piece 1 reuses it with its instruction offset
    else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(src), "n"(FRAG_BYTES) : "memory");
#endif
  }
  __device__ __forceinline__ void cursor_update(int F) {
    voff = (F + 1 == STAGESB_PER_PASS) ? lane16 : voff + STAGE_BYTES;
    asm volatile("" : "+v"(voff));
  }
  __device__ __forceinline__ void start(lds_char* lds_, const char* stream0, const char* stream1, int first_model, int lane, int wave) {
    lds = lds_;
    lane16 = (uint32_t)lane * 16u;
    rd_base = LDS_RING + lane16;
    const uint32_t col = (uint32_t)wave * (P_PIECES * FRAG_BYTES);
#pragma unroll
    for (int k = 0; k < B_RING; ++k) m0s[k] = (uint32_t)(uintptr_t)lds_ + LDS_RING + k * STAGE_BYTES + col;
    base[0] = stream0 + col;
    base[1] = stream1 + col;
    cur = nxt = base[first_model];
    voff = lane16;
#pragma unroll
    for (int F = 0; F < B_RING - 1; ++F) {
      issue_piece(0, F);
      issue_piece(1, F);
      cursor_update(F);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  __device__ __forceinline__ void begin_tile(int next_model) {
    cur = nxt;
    nxt = next_model ? base[1] : base[0];
    asm volatile("" : "+s"(cur), "+s"(nxt));
    // launder the read base once per tile: as a loop invariant hipcc materialises base + <slot, fragment> for all 64 ring
    // positions in 64 VGPRs outside the tile loop instead of using the ds_read offset field
    asm volatile("" : "+v"(rd_base));
  }
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  __device__ __forceinline__ u32x4 read(int pos) const {
    return *(const __attribute__((address_space(3))) u32x4*)(lds + rd_base + ((pos / STAGE_FRAGS) % B_RING) * STAGE_BYTES +
                                                              (pos % STAGE_FRAGS) * FRAG_BYTES);
  }
  __device__ __forceinline__ void prime(u32x4 (&q)[B_AHEAD]) const {
#pragma unroll
    for (int j = 0; j < B_AHEAD; ++j) q[j] = read(j);
  }
};

// ---- epilogues, one point group: quarter qc (0..7) = accumulator registers 2qc, 2qc+1 -> dword qc&3 of k-step
// 2T + (qc>>2) of the next layer's B operand.  The accumulators live in ARCHITECTURAL registers here (gfx950 MFMA takes
// VGPRs for C/D): an "a" constraint anywhere in the kernel makes hipcc split the 256-register budget of a two-waves-per-SIMD
// kernel 128 VGPR + 128 AGPR (SIRegisterInfo: usesAGPRs => MaxNumVGPRs /= 2), which does not hold the two 64-register
// activation buffers -- and a VGPR accumulator needs no v_accvgpr_read: a quarter is TWO VALU instructions.  The asm
// statements pin each quarter to the k-step it was written in (see PackEpi in mlp_core_bf16.h).
struct NoEpiP {
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void finish(int, int, const f32x16&) {}
};

template <bool RELU>
struct PackEpiP {
  u32x4 (&dst)[KS_HID];
  __device__ __forceinline__ explicit PackEpiP(u32x4 (&d)[KS_HID]) : dst(d) {}
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void finish(int T, int qc, const f32x16& acc) {
    uint32_t pk;
    if (RELU)
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(pk) : "v"(acc[2 * qc]), "v"(acc[2 * qc + 1]));
    else
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(acc[2 * qc]), "v"(acc[2 * qc + 1]));
    dst[2 * T + (qc >> 2)][qc & 3] = pk;
  }
};

// xyz_encoding_8: as PackEpiP<true>, plus static_sigma (256 -> 1) in fp32 on the un-rounded activations
struct SigmaEpiP {
  u32x4 (&dst)[KS_HID];
  float& sg;
  const lds_float* wsig;
  int h;
  f32x4 wv[4];
  __device__ __forceinline__ SigmaEpiP(u32x4 (&d)[KS_HID], float& s, const lds_float* w, int h_) : dst(d), sg(s), wsig(w), h(h_) {}
  __device__ __forceinline__ void prefetch(int T) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wv[c] = lds_f4(wsig + 32 * T + 8 * c + 4 * h);
  }
  __device__ __forceinline__ void finish(int T, int qc, const f32x16& acc) {
    const int e = 2 * (qc & 1);
    float x0 = acc[2 * qc], x1 = acc[2 * qc + 1];
    asm volatile("" : "+v"(x0), "+v"(x1));
    dst[2 * T + (qc >> 2)][qc & 3] = pack_pair<true>(x0, x1);
    sg = fmaf(wv[qc >> 1][e], fmaxf(x0, 0.0f), sg);
    sg = fmaf(wv[qc >> 1][e + 1], fmaxf(x1, 0.0f), sg);
    asm volatile("" : "+v"(sg));
  }
};

struct RgbEpiP {   // static_rgb: sigmoid, fp32 out
  f32x16 (&feat)[2];
  __device__ __forceinline__ explicit RgbEpiP(f32x16 (&f)[2]) : feat(f) {}
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void finish(int T, int qc, const f32x16& acc) {
    float x0 = acc[2 * qc], x1 = acc[2 * qc + 1];
    asm volatile("" : "+v"(x0), "+v"(x1));
    feat[T][2 * qc] = sigmoid_fast(x0);
    feat[T][2 * qc + 1] = sigmoid_fast(x1);
  }
};

// acc-layout bias of output tile T, half `half` (registers 8 half .. 8 half + 7): element 4c + j = bias[32T + 8c + 4h + j]
__device__ __forceinline__ void load_bias_into(f32x16& acc, const lds_float* bias, int T, int h, int half) {
#pragma unroll
  for (int c = 2 * half; c < 2 * half + 2; ++c) {
    const f32x4 b = lds_f4(bias + 32 * T + 8 * c + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 * c + j] = b[j];
  }
}

// One layer.  NT output tiles; per tile NSA k-steps with B operands srcA[s] then NSB from srcB.  Template parameters as
// mma_layer_b (FBASE: pass-relative first fragment, G0: index of the layer's first tile in the pass -- tile G accumulates
// in accs[G & 1] --, PT: the previous layer's last tile, whose epilogue `prev` runs behind this layer's first tile).  On entry
// accs[G0 & 1] holds the bias of tile 0; the bias of each following tile (and of the NEXT layer's tile 0, from next_bias) is
// read into the other accumulator in the tile's last two k-steps, after the epilogue quarters have drained it.
template <int NT, int NSA, int NSB, int FBASE, int G0, int PT, int NA, int NB, class PREV, class EPI>
__device__ __forceinline__ void mma_layer_p(WeightPipeP& p, const u32x4 (&srcA)[NA], const u32x4 (&srcB)[NB], u32x4 (&q)[B_AHEAD],
                                            f32x16 (&accs)[2], const lds_float* bias, const lds_float* next_bias, int h, PREV& prev,
                                            EPI& epi) {
  static_assert(NSA <= NA && NSB <= NB, "source too small");
  constexpr int NS = NSA + NSB;
  static_assert(NS >= 16 || NS == 6 || NS == 8, "epilogue quarters must finish before the last source tile is read");
  constexpr bool LONG = NS >= 16;
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    const int cur = (G0 + T) & 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int i = FBASE + T * NS + s;
      // quarters of the previous tile's epilogue carried by this k-step: long layers one per k-step in k-steps 1..8 (k-step 14
      // of a layer's first tile is the first reader of the result), short layers (K = 96, 128) two per k-step in k-steps 1..4
      const int first = LONG ? s - 1 : 2 * (s - 1);
      const int count = s < 1 ? 0 : (LONG ? (s <= 8 ? 1 : 0) : (s <= 4 ? 2 : 0));
      const f32x16& pa = accs[cur ^ 1];

      u32x4 af = q[i % B_AHEAD];
      asm volatile("" : "+v"(af));   // ties this k-step's MFMA into the side-effect chain (see the epilogue notes in mlp_core_bf16.h)
      const u32x4 b = s < NSA ? srcA[s < NSA ? s : 0] : srcB[s < NSA ? 0 : s - NSA];
      accs[cur] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, b), accs[cur], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // MFMA first, its fillers behind it
      if (p_piece_at(i, P_STAGGER) >= 0) p.issue_piece(p_piece_at(i, P_STAGGER), i / STAGE_FRAGS + B_RING - 1);
#ifdef CRNERF_EXP_NOLDSREAD   // (experiments only) keep the fragments of the first four k-steps forever
      asm volatile("" : "+v"(q[i % B_AHEAD]));
#elif defined(CRNERF_EXP_HALFLDSREAD)   // (experiments only) every other fragment read
      if (i & 1) q[i % B_AHEAD] = p.read(b_pos(i + B_AHEAD)); else asm volatile("" : "+v"(q[i % B_AHEAD]));
#else
      q[i % B_AHEAD] = p.read(b_pos(i + B_AHEAD));
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifndef CRNERF_EXP_NOEPI   // (experiments only)
#pragma unroll
      for (int u = 0; u < count; ++u) {
        if (T == 0) prev.finish(PT, first + u, pa);
        else epi.finish(T - 1, first + u, pa);
      }
#endif
      if (s == 0) {
        if (T == 0) prev.prefetch(PT);
        else epi.prefetch(T - 1);
      }
      if (s == NS - 2 || s == NS - 1) {   // the next tile's bias, straight into its accumulator (drained by k-step 8 / 4)
        const lds_float* nb = (T + 1 < NT) ? bias : next_bias;
        load_bias_into(accs[cur ^ 1], nb, (T + 1 < NT) ? T + 1 : 0, h, s - (NS - 2));
      }
      if (b_advance_at(i)) p.advance();
      if (b_cursor_at(i)) p.cursor_update(i / STAGE_FRAGS + B_RING - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// One 32-point tile through one model.  pe[s]: the xyz embedding as B operands (posenc_b); dirsrc: this lane half's 2 x 16 bytes
// of the ray's direction embedding, parked in LDS.  Returns feat[t][r] = rgb feature 32t + 8(r>>2) + 4h + (r&3) of point p, and
// sigma (valid in both lane halves).
__device__ __forceinline__ void mlp_tile_p(WeightPipeP& p, int model, int next_model, const u32x4 (&pe)[KS_XYZ], const lds_char* dirsrc,
                                           f32x16 (&feat)[2], float& sigma, int h, u32x4 (&q)[B_AHEAD], PhaseTimer& tm) {
  p.begin_tile(next_model);
  uint32_t c_off = model ? LDS_CONST1 : LDS_CONST0;
  asm volatile("" : "+s"(c_off));     // loop-invariant LDS: launder the address once per tile (LICM would hoist ~1,300 reads)
  const lds_float* C = (const lds_float*)(p.lds + c_off);
  const lds_float* B1 = C + C_BIAS;
  u32x4 actA[KS_HID], actB[KS_HID];
  f32x16 accs[2];
  float sg = 0.0f;
  load_bias_into(accs[0], B1, 0, h, 0);
  load_bias_into(accs[0], B1, 0, h, 1);
  tm.tick(T_PROLOGUE);

  NoEpiP none;
  PackEpiP<true> eA(actA), eB(actB);
  PackEpiP<false> efin(actA);
  SigmaEpiP e8(actB, sg, C + C_WSIG, h);
  RgbEpiP ergb(feat);
  mma_layer_p<8, KS_XYZ, 0, OFFB_L1, 0, 0>(p, pe, pe, q, accs, B1, B1 + 1 * W_HIDDEN, h, none, eA);                                // xyz_encoding_1
  mma_layer_p<8, KS_HID, 0, OFFB_L2, 8, 7>(p, actA, actA, q, accs, B1 + 1 * W_HIDDEN, B1 + 2 * W_HIDDEN, h, eA, eB);               // 2
  mma_layer_p<8, KS_HID, 0, OFFB_L2 + FB_HID, 16, 7>(p, actB, actB, q, accs, B1 + 2 * W_HIDDEN, B1 + 3 * W_HIDDEN, h, eB, eA);     // 3
  mma_layer_p<8, KS_HID, 0, OFFB_L2 + 2 * FB_HID, 24, 7>(p, actA, actA, q, accs, B1 + 3 * W_HIDDEN, B1 + 4 * W_HIDDEN, h, eA, eB); // 4
  mma_layer_p<8, KS_XYZ, KS_HID, OFFB_L5, 32, 7>(p, pe, actB, q, accs, B1 + 4 * W_HIDDEN, B1 + 5 * W_HIDDEN, h, eB, eA);           // 5 = Linear(cat[xyz, h])
  mma_layer_p<8, KS_HID, 0, OFFB_L6, 40, 7>(p, actA, actA, q, accs, B1 + 5 * W_HIDDEN, B1 + 6 * W_HIDDEN, h, eA, eB);              // 6
  mma_layer_p<8, KS_HID, 0, OFFB_L6 + FB_HID, 48, 7>(p, actB, actB, q, accs, B1 + 6 * W_HIDDEN, B1 + 7 * W_HIDDEN, h, eB, eA);     // 7
  mma_layer_p<8, KS_HID, 0, OFFB_L6 + 2 * FB_HID, 56, 7>(p, actA, actA, q, accs, B1 + 7 * W_HIDDEN, C + C_BFIN, h, eA, e8);        // 8 (+ static_sigma)
  mma_layer_p<8, KS_HID, 0, OFFB_FIN, 64, 7>(p, actB, actB, q, accs, C + C_BFIN, C + C_BDIR, h, e8, efin);                         // xyz_encoding_final
  tm.tick(T_MMA);
  sg += __shfl_xor(sg, 32);
  sigma = softplus_fast(sg + C[C_BSIG]);
  u32x4 dv[KS_DIR];
#pragma unroll
  for (int s = 0; s < KS_DIR; ++s) dv[s] = *(const __attribute__((address_space(3))) u32x4*)(dirsrc + 32 * s);
  tm.tick(T_SIGMA);
  mma_layer_p<4, KS_HID, KS_DIR, OFFB_DIR, 72, 7>(p, actA, dv, q, accs, C + C_BDIR, C + C_BRGB, h, efin, eB);                      // dir_encoding
  mma_layer_p<2, KS_HALF, 0, OFFB_RGB, 76, 3>(p, actB, actB, q, accs, C + C_BRGB, C + C_BRGB, h, eB, ergb);                         // static_rgb
  tm.tick(T_MMA);
#pragma unroll
  for (int qc = 0; qc < 8; ++qc) {   // rgb's last tile: nothing left to hide it behind
    ergb.finish(1, qc, accs[(76 + 1) & 1]);
  }
  tm.tick(T_EPILOGUE);
}

}  // namespace crnerf
