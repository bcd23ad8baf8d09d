// NeRF_sigma forward, fp32-ACCURATE on the fp16 matrix cores ("h2" core, round 4): one 32-point tile per wavefront, four wavefronts per
// workgroup (one per SIMD, 512 registers).
//
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182, in fp32.  Every fp32 operand of the eleven nn.Linear is split into TWO fp16
// pieces, x = h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) (11 + 11 mantissa bits and the sign of h2: x to 2^-24, one fp32 rounding), and a
// product is the sum of the three leading piece products w2 a1 + w1 a2 + w1 a1 (small terms first; each exact in fp32, accumulated in the MFMA's
// fp32 accumulator; the dropped w2 a2 is <= 2^-24 of the product).  Weights are split at pack time and stored scaled by 2^8 (layout.h "fragH";
// so are the biases in the pack's consts block), so an accumulator holds 2^8 x the layer's pre-activation.  Range: |activation| < 65,504 (tracked,
// see `amax`) and |weight| < 255 (checked by the pack).
//
// Round-4 structure ("lazy epilogue"; the round-3 h2 core was mlp_core_x3.h's code with two pieces: matrix pipe busy 53-56 %, 4 VALU instructions
// per MFMA, every layer boundary ~0.5 k VALU instructions with no MFMA in flight):
//   * K-outer as the x3 core -- all output tiles of a layer accumulate (128 registers in the AGPR half of the file) while the contraction is
//     walked once, stream "fragH" unchanged -- but a layer's output is NEVER finished between the layers: in a layer's last k-step the third MFMA
//     of every tile writes its result not back to the accumulator but to a 128-register RAW set in architectural VGPRs (vDst != srcC), and that
//     set is the SOURCE of the next layer: the eight values of k-step s + 1 are finished while the 3 NT MFMAs of k-step s run -- relu (fp32), range
//     max, then the split straight into packed operand halves by v_fma_mixlo/mixhi_f16 (h1 = fp16(x * 2^-8), h2 = fp16(x * 2^-8 - h1): scale,
//     subtraction and conversion in ONE instruction each) -- 4 VALU per value, one value per tile in the MFMAs' shadow, no v_accvgpr_read anywhere.
//     The operands of a layer's k-step 0 are made in the LAST k-step of the layer before it.
//   * an accumulator tile its last MFMA has left takes the next layer's bias straight from LDS: no accumulator initialisation between layers.
//   * the sigma head (256 -> 1, fp32 VALU) rides in xyz_encoding_final's walk of h8: one fma per finished value.
//   * issue order is pinned with sched_barrier(0), a tile PAIR at a time (consecutive MFMAs on different accumulators): MFMA | finish half | MFMA |
//     finish half | MFMA | split half | MFMA | split half | MFMA | queue refills, LDS-DMA piece | MFMA | the same -- hipcc's own schedule of the
//     round-3 core put most fragment reads directly in front of the MFMA that consumes them.
// Weights: the seven-slot 16 KiB-stage LDS ring of xcore_pipe.h; a k-step of an eight-tile layer is exactly one stage (8 tiles x 2 pieces).
#pragma once
#ifndef CRNERF_X_NP
#define CRNERF_X_NP 2
#endif
#include "xcore_pipe.h"
#if CRNERF_X_NP != 2
#error "mlp_core_h2.h is the two-piece fp16 core; the three-piece bf16 core is mlp_core_x3.h"
#endif

namespace crnerf {
inline namespace xcore_h2 {

constexpr float H2_ACT_LIMIT = 65504.0f;   // largest finite fp16
constexpr float H2_INV = 1.0f / H2_WSCALE;

#define CRNERF_MFMA_H(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(xbf16x8, (a)), __builtin_bit_cast(xbf16x8, (b)), (c), 0, 0, 0)
#define CRNERF_H2_PIN() __builtin_amdgcn_sched_barrier(0)

struct BOpH { xu32x4 b1, b2; };   // the two piece operands of one k-step: dword d = k-values 2d (low half), 2d + 1 (high half)

// value x (fp32) -> half e&1 of dword e/2 of the two operand vectors: h1 = fp16(x * sc), then h2 = fp16(x * sc - h1).  x * sc is exact (sc a power of
// two), so is the difference: each result is rounded once, to nearest even, fp16 subnormals kept -- the arithmetic of the round-3 split, in two
// instructions.  (v_fma_mix*: op_sel_hi picks fp32 (0) or fp16 (1) per source, op_sel the fp16 half; mixlo / mixhi keep the
// other half of the destination.)  The two steps are separate calls so that the layer code can put an MFMA between the write of h1 and its use.
// HAZARD, and how it is closed (round 5; tools/isa_audit.py "valu->mfma", DESIGN Appendix A): an MFMA must not read a register for 2 wait states after
// a VALU wrote it.  hipcc pads that between ITS OWN instructions; an asm statement is not a VALU write to its hazard recogniser (it gets the fixed
// one-state pad of any asm output), so whether the MFMA that takes b1 / b2 as its B operand is far enough behind the last split is a property of one
// build -- the round-4 training form had `v_fma_mixhi v15 ; v_cndmask ; v_mfma ... v[14:17]` in one place and single tiles came out wrong.  Closed
// by construction in both consumer loops: mma_layer_h2 (below) pins its stream with sched_barrier(0) between every MFMA and every chunk of splits, and
// the last split of a k-step's operands is followed by two MFMAs and two tail gaps of the SAME k-step before the hand-over; mma_layer_h2t
// (mlp_core_h2t.h), which leaves the schedule to hipcc, ends every hand-over with h2_operands_ready(b): a two-state nop tied to the eight operand
// registers, behind every split that writes them and in front of every MFMA that reads them.  tests/test_host.py audits the ISA of every build.
__device__ __forceinline__ void h2_split_first(xu32x4& b1, int e, float x, float sc) {
  uint32_t h1 = b1[e >> 1];
  if (e & 1) asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h1) : "v"(x), "v"(sc));
  else asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h1) : "v"(x), "v"(sc));
  b1[e >> 1] = h1;
}
__device__ __forceinline__ void h2_split_second(const xu32x4& b1, xu32x4& b2, int e, float x, float sc) {
  uint32_t h2 = b2[e >> 1];
  if (e & 1) asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h2) : "v"(x), "v"(sc), "v"(b1[e >> 1]));
  else asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(h2) : "v"(x), "v"(sc), "v"(b1[e >> 1]));
  b2[e >> 1] = h2;
}
// the operands written by h2_split_* may be read by MFMAs from here on (see HAZARD above): s_nop 1 = the two wait states VALU write -> MFMA read needs
__device__ __forceinline__ void h2_operands_ready(xu32x4& b1, xu32x4& b2) { asm volatile("s_nop 1" : "+v"(b1), "+v"(b2)); }
__device__ __forceinline__ void h2_split_into(xu32x4& b1, xu32x4& b2, int e, float x, float sc) {
  h2_split_first(b1, e, x, sc);
  h2_split_second(b1, b2, e, x, sc);
}

// acc-layout bias of output tile T straight from LDS: element 4q + j = bias[32T + 8q + 4h + j]
__device__ __forceinline__ void init_acc_tile(f32x16& acc, const lds_float* bias, int T, int h) {
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias + 32 * T + 8 * qq + 4 * h);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[4 * qq + k] = bv[k];
  }
}

// what a tile carries through its layers beside the accumulators
struct H2Carry {
  float amax;        // running max |raw activation| of this lane, scaled by 2^8 like the accumulators (range guard, see mlp_tile_x3)
  float emax;        // max |embedding value| (unscaled)
  float xprev;       // the even value of the pair being finished
  float sg;          // static_sigma partial sum (scaled by 2^8)
  f32x4 sw0, sw1;    // static_sigma weights of the group being finished
};

// One layer on NT output tiles.  acc[0..NT): the accumulators, holding the layer's bias on entry.  Its k-steps take their B operands from two parts, in
// the order PF says: NSP k-steps of ready-made pieces pcs[.] (an embedding) and NSR k-steps of the RAW outputs raw[.] of the layer before (group g =
// registers 8(g%2) .. +7 of tile g/2), finished on the way: FIN 1 = relu, 2 = linear (xyz_encoding_final's output).  In the LAST k-step the third
// MFMA of every tile writes its result to raw[T] (by then the old raw values are all consumed) -- the raw set lives in architectural VGPRs, read by
// the finishing VALU code directly, the accumulators in the AGPR half -- and acc[T] takes the NEXT layer's bias (NTN tiles from nbias) straight
// from LDS.  b: in = the operands of k-step 0, out = those of the next layer's k-step 0 -- made here in the last k-step from the new raw tile 0
// (NEXT_FIN 1 / 2), copied from next0 (NEXT_FIN 0), or left alone (-1).  SIG / NEXT_SIG: the finished values also feed static_sigma.
template <int NT, bool PF, int NSP, int NSR, int FIN, int NTN, int NEXT_FIN, bool SIG, bool NEXT_SIG, int NPC>
__device__ __forceinline__ void mma_layer_h2(WeightPipeX& p, xu32x4 (&q)[X_AHEAD], BOpH& b, const xu32x4 (&pcs)[NPC][2], f32x16 (&raw)[8], f32x16 (&acc)[8],
                                             const lds_float* nbias, const xu32x4 (&next0)[2], const lds_float* wsig, H2Carry& c, int h) {
  constexpr int NS = NSP + NSR;
  static_assert(NSP <= NPC && NSR <= 16 && (NS * NT * 2) % X_STAGE_FRAGS == 0 && (NS * NT * 2) % X_AHEAD == 0, "layer: whole stages, whole queue turns");
  static_assert(NT == 8 || NT == 4 || NT == 2, "tiles");
  static_assert(NT >= 4 || (NEXT_FIN <= 0 && !NEXT_SIG), "a two-tile layer has no room for the next layer's first operands");
  static_assert(NTN <= NT, "the next layer's bias goes into this layer's accumulators");
  const float inv = H2_INV;
  // queue slot of fragment f refilled with fragment f + X_AHEAD; ring kept going (call in stream order, after the fragment's last use)
  auto refill = [&](int f) {
    const int slot = f % X_STAGE_FRAGS;
    if (slot % (X_STAGE_FRAGS / X_PIECES) == 0) p.issue_piece(slot / (X_STAGE_FRAGS / X_PIECES));
    q[f % X_AHEAD] = p.read_slot(slot + X_AHEAD);
    if (slot == X_STAGE_FRAGS - 1) p.advance();
  };
  // one value of the operands being made: relu + range max (+ sigma) + first piece -- first half; the second piece -- second half
  auto finish_a = [&](BOpH& o, float v, int fin, bool sig, int e, float& xr) {
    xr = fin == 1 ? fmaxf(v, 0.0f) : v;
    // (volatile asm: nothing reads amax / sg before the end of the tile, and hipcc otherwise SINKS these updates there -- with every finished value
    // of the tile spilled to scratch on the way)
    if (e & 1) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(c.amax) : "v"(c.xprev), "v"(xr));
    else c.xprev = xr;
    if (sig) {
      const float w = e < 4 ? c.sw0[e & 3] : c.sw1[e & 3];
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(c.sg) : "v"(w), "v"(xr));
    }
    h2_split_first(o.b1, e, xr, inv);
  };
  auto finish_b = [&](BOpH& o, int e, float xr) {
    h2_split_second(o.b1, o.b2, e, xr, inv);
  };
  BOpH nb = b;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const bool last = s == NS - 1;
    // what this k-step's chunks make: the operands of k-step s + 1 (raw group ng, or ready pieces) or of the next layer's k-step 0
    const bool nraw = !last && (PF ? s + 1 >= NSP : s + 1 < NSR);
    const int ng = PF ? s + 1 - NSP : s + 1;
    const int cfin = last ? (NEXT_FIN > 0 ? NEXT_FIN : 0) : (nraw ? FIN : 0);
    const bool csig = last ? NEXT_SIG : (nraw && SIG);
    if (!last && !nraw) {
      nb.b1 = pcs[PF ? s + 1 : s + 1 - NSR][0];
      nb.b2 = pcs[PF ? s + 1 : s + 1 - NSR][1];
    }
    if (last && NEXT_FIN == 0) {
      nb.b1 = next0[0];
      nb.b2 = next0[1];
    }
    // which tile carries value e's chunk: in the last k-step the chunks read the new raw tile 0, whose MFMA is issued with the first tile PAIR
    auto chunk_tile = [&](int e) { return last ? 2 + e * (NT - 2) / 8 : e * NT / 8; };
    auto chunks_a = [&](int T, float (&xr)[8]) {
      if (cfin) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (chunk_tile(e) == T) finish_a(nb, last ? raw[0][e] : raw[ng >> 1][8 * (ng & 1) + e], cfin, csig, e, xr[e]);
      }
    };
    auto chunks_b = [&](int T, const float (&xr)[8]) {
      if (cfin) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (chunk_tile(e) == T) finish_b(nb, e, xr[e]);
      }
    };
    // behind a tile's last MFMA: its two queue slots refilled (+ the ring's work); last k-step: the next layer's bias into the accumulator the MFMA
    // has just left; last tile: the static_sigma weights of the group the NEXT k-step's chunks finish (features 32 (g/2) + 16 (g%2) + 8 (e>>2) + 4h + (e&3))
    auto tail_gap = [&](int T) {
      const int f = (s * NT + T) * 2;
      refill(f);
      refill(f + 1);
      if (last && T < NTN) init_acc_tile(acc[T], nbias, T, h);
      if (T == NT - 1) {
        const int t = s + 1;   // the k-step whose chunks will use the weights
        const bool traw = t < NS - 1 && (PF ? t + 1 >= NSP : t + 1 < NSR);
        const bool need = t < NS - 1 ? (SIG && traw) : NEXT_SIG;
        const int g = t < NS - 1 ? (PF ? t + 1 - NSP : t + 1) : (t == NS - 1 ? 0 : 1);
        if (need) {
          c.sw0 = *(const __attribute__((address_space(3))) f32x4*)(wsig + 32 * (g >> 1) + 16 * (g & 1) + 4 * h);
          c.sw1 = *(const __attribute__((address_space(3))) f32x4*)(wsig + 32 * (g >> 1) + 16 * (g & 1) + 8 + 4 * h);
        }
      }
    };
#pragma unroll
    for (int T = 0; T < NT; T += 2) {   // two tiles at a time: consecutive MFMAs belong to different accumulators
      const int f = (s * NT + T) * 2;
      const xu32x4 u1 = q[f % X_AHEAD], u2 = q[(f + 1) % X_AHEAD], w1 = q[(f + 2) % X_AHEAD], w2 = q[(f + 3) % X_AHEAD];
      float xa[8], xb[8];
      acc[T] = CRNERF_MFMA_H(u2, b.b1, acc[T]);               // small terms first
      CRNERF_H2_PIN();
      chunks_a(T, xa);
      CRNERF_H2_PIN();
      acc[T + 1] = CRNERF_MFMA_H(w2, b.b1, acc[T + 1]);
      CRNERF_H2_PIN();
      chunks_a(T + 1, xb);
      CRNERF_H2_PIN();
      acc[T] = CRNERF_MFMA_H(u1, b.b2, acc[T]);
      CRNERF_H2_PIN();
      chunks_b(T, xa);
      CRNERF_H2_PIN();
      acc[T + 1] = CRNERF_MFMA_H(w1, b.b2, acc[T + 1]);
      CRNERF_H2_PIN();
      chunks_b(T + 1, xb);
      CRNERF_H2_PIN();
      if (last) {                                             // the layer's output leaves the accumulator file
        raw[T] = CRNERF_MFMA_H(u1, b.b1, acc[T]);
        asm volatile("" : "+v"(raw[T]));                      // (into architectural VGPRs: hipcc's own choice is the AGPR half + one v_accvgpr_read per value)
      } else acc[T] = CRNERF_MFMA_H(u1, b.b1, acc[T]);
      CRNERF_H2_PIN();
      tail_gap(T);
      CRNERF_H2_PIN();
      if (last) {
        raw[T + 1] = CRNERF_MFMA_H(w1, b.b1, acc[T + 1]);
        asm volatile("" : "+v"(raw[T + 1]));
      } else acc[T + 1] = CRNERF_MFMA_H(w1, b.b1, acc[T + 1]);
      CRNERF_H2_PIN();
      tail_gap(T + 1);
      CRNERF_H2_PIN();
    }
    b = nb;     // (no h2_operands_ready here: the pins above already order every split of nb in front of two MFMAs + two tail gaps of THIS k-step)
  }
}

// One 32-point tile through one model.  pe / dv: the positional embeddings in the register order of posenc_regs (posenc.h), exactly as the x3 core
// takes them.  Returns feat[t][4q+j] = rgb feature 32t+8q+4h+j of point p, and sigma (both lane halves).
// Range guard: an activation beyond fp16's range would split into (inf, -inf) pieces, turn into NaN in the next layer and be clamped to 0 by its
// relu -- a finite, wrong result -- so the tile tracks max |operand| and POISONS the 65 outputs of such a point with NaN.
// (The kernels call mlp_tile_x3 -- mlp_core_h2t.h -- which is this function for inference and the training form mlp_tile_h2t when rows are saved.)
__device__ __forceinline__ void mlp_tile_h2i(WeightPipeX& p, int model, const f32x16 (&pe)[3], const f32x16 (&dv)[1], f32x16 (&feat)[2], float& sigma,
                                             int h, xu32x4 (&q)[X_AHEAD], PhaseTimer& tm) {
  const lds_float* C = (const lds_float*)(p.lds + (model ? LDS_CONST1 : LDS_CONST0));
  const lds_float* B1 = C + C_BIAS;
  const lds_float* WS = C + C_WSIG;
  f32x16 X[8], R[8];   // X: the accumulators (AGPR half of the register file); R: the raw output of the layer before (architectural VGPRs)
  H2Carry c;
  c.amax = 0.0f;
  c.emax = 0.0f;
  c.xprev = 0.0f;
  c.sg = 0.0f;
  // the embeddings as ready-made piece operands (k-step s = registers 8s .. 8s + 7 of the flat posenc_regs order), unscaled
  xu32x4 pep[KS_XYZ][2], dvp[KS_DIR][2];
#pragma unroll
  for (int s = 0; s < KS_XYZ; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = pe[(8 * s + e) / 16][(8 * s + e) % 16];
      c.emax = fmaxf(c.emax, fabsf(v));
      h2_split_into(pep[s][0], pep[s][1], e, v, 1.0f);
    }
#pragma unroll
  for (int s = 0; s < KS_DIR; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = dv[0][8 * s + e];
      c.emax = fmaxf(c.emax, fabsf(v));
      h2_split_into(dvp[s][0], dvp[s][1], e, v, 1.0f);
    }
  // the embedding operands are read by MFMAs only: they live in the AGPR half, next to the accumulators, and leave the VGPRs to the raw set
#pragma unroll
  for (int s = 0; s < KS_XYZ; ++s) asm volatile("" : "+a"(pep[s][0]), "+a"(pep[s][1]));
#pragma unroll
  for (int s = 0; s < KS_DIR; ++s) asm volatile("" : "+a"(dvp[s][0]), "+a"(dvp[s][1]));
  init_acc<8>(X, B1, h);                                   // xyz_encoding_1's bias (the pack's consts carry the 2^8)
  BOpH b{pep[0][0], pep[0][1]};
  tm.tick(T_PROLOGUE);

  // layer     NT  PF    NSP     NSR     FIN NTN NEXT_FIN SIG    NEXT_SIG                       next bias          next0   wsig
  mma_layer_h2<8, true, KS_XYZ, 0, 0, 8, 1, false, false>(p, q, b, pep, R, X, B1 + 1 * W_HIDDEN, pep[0], WS, c, h);          // xyz_encoding_1
  tm.tick(T_X0);
  mma_layer_h2<8, true, 0, KS_HID, 1, 8, 1, false, false>(p, q, b, pep, R, X, B1 + 2 * W_HIDDEN, pep[0], WS, c, h);          // 2
  mma_layer_h2<8, true, 0, KS_HID, 1, 8, 1, false, false>(p, q, b, pep, R, X, B1 + 3 * W_HIDDEN, pep[0], WS, c, h);          // 3
  mma_layer_h2<8, true, 0, KS_HID, 1, 8, 0, false, false>(p, q, b, pep, R, X, B1 + 4 * W_HIDDEN, pep[0], WS, c, h);          // 4 (5 starts on the embedding)
  tm.tick(T_X1);
  mma_layer_h2<8, true, KS_XYZ, KS_HID, 1, 8, 1, false, false>(p, q, b, pep, R, X, B1 + 5 * W_HIDDEN, pep[0], WS, c, h);     // 5 = Linear(cat[xyz, h])
  tm.tick(T_X2);
  mma_layer_h2<8, true, 0, KS_HID, 1, 8, 1, false, false>(p, q, b, pep, R, X, B1 + 6 * W_HIDDEN, pep[0], WS, c, h);          // 6
  mma_layer_h2<8, true, 0, KS_HID, 1, 8, 1, false, false>(p, q, b, pep, R, X, B1 + 7 * W_HIDDEN, pep[0], WS, c, h);          // 7
  mma_layer_h2<8, true, 0, KS_HID, 1, 8, 1, false, true>(p, q, b, pep, R, X, C + C_BFIN, pep[0], WS, c, h);                  // 8 (its tail starts static_sigma)
  tm.tick(T_X3);
  mma_layer_h2<8, true, 0, KS_HID, 1, 4, 2, true, false>(p, q, b, pep, R, X, C + C_BDIR, pep[0], WS, c, h);                  // xyz_encoding_final (+ static_sigma on h8)
  tm.tick(T_X4);
  {
    const float s = c.sg + __shfl_xor(c.sg, 32);
    sigma = softplus_ref(s * H2_INV + C[C_BSIG]);
  }
  tm.tick(T_SIGMA);
  mma_layer_h2<4, false, KS_DIR, KS_HID, 2, 2, 1, false, false>(p, q, b, dvp, R, X, C + C_BRGB, pep[0], WS, c, h);           // dir_encoding = relu(Linear(cat[final, dir]))
  tm.tick(T_X5);
  mma_layer_h2<2, true, 0, KS_HALF, 1, 0, -1, false, false>(p, q, b, pep, R, X, C + C_BRGB, pep[0], WS, c, h);               // static_rgb
  tm.tick(T_X6);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) feat[t][r] = sigmoid_ref(R[t][r] * H2_INV);
  {   // an operand left fp16's range somewhere in this point's MLP: NaN out, not a finite wrong answer (amax of the raw values is scaled by 2^8)
    float am = fmaxf(c.amax, c.emax * H2_WSCALE);
    am = fmaxf(am, __shfl_xor(am, 32));
    if (!(am < H2_ACT_LIMIT * H2_WSCALE)) {
      const float poison = __builtin_nanf("");
      sigma = poison;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) feat[t][r] = poison;
    }
  }
  tm.tick(T_EPILOGUE);
}

}  // inline namespace xcore_h2
}  // namespace crnerf
