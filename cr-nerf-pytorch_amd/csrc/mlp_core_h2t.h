// The h2 core's TRAINING form ("h2t"): NeRF_sigma forward with the layer activations kept for the backward, and the generic layer the backward-data
// kernel (mlp_backward_h2.hip) is built from -- fp32-ACCURATE on the fp16 matrix cores, two fp16 pieces per operand, three MFMAs per product
// (the arithmetic of mlp_core_h2.h: x = h1 + h2, h1 = fp16(x), h2 = fp16(x - h1); w a ~= w2 a1 + w1 a2 + w1 a1, small terms first; weights and
// biases carry 2^8 from the pack, layout.h "fragH" / "fragHT").
//
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182, in fp32; what is saved is what the fp32 / f32x3 training twins save
// (xcore_pipe.h ActSaveX: fp32 activation rows in reference feature order + the relu-activity bits), so the backward twins read it unchanged.
//
// Why a second form: the inference core (mlp_core_h2.h) never finishes a layer's output between the layers -- raw accumulators are the next layer's
// source, finished value by value in the MFMAs' shadow -- which leaves no place where a whole fp32 activation row exists.  The training form has the
// structure of the x3 core (mlp_core_x3.h): K-outer, all output tiles of a layer accumulate while the contraction is walked once; a layer's
// finished output (scale 2^-8, relu, range max) sits in 128 registers and is split, eight values per k-step, when the NEXT layer walks it -- which
// is also when its two 16-byte row pieces leave for HBM.  The stores, not the epilogue, are what this form's time goes to (10.5 KB per point).
//
// Operand scale: mma_layer_h2t splits x * sc with a per-LANE power of two sc.  Forward: sc = 1.  Backward: the deltas of a point span whatever the loss hands down (1e-9 is ordinary), far below fp16's normal range;
// every point's delta vector is therefore scaled so that its largest entry sits in [2^7, 2^8) -- the products are linear in it, the accumulator
// is scaled back by the exact inverse -- and a point's small entries degrade to an absolute error of 2^-25 x 2^-7 of its largest one, invisible
// in a dot product the large entries dominate.  So the backward has no range failure of its own.
#pragma once
#include "mlp_core_h2.h"

namespace crnerf {
inline namespace xcore_h2 {

constexpr int H2T_ROW_BURST = 4;   // k-steps whose row pieces leave together (mma_layer_h2t)

// One layer: NT output tiles; k-steps 0..NSA-1 take their B operands from srcA (registers 8(s%2) .. +7 of tile s/2), the following NSB from srcB.
// q always holds the next X_AHEAD fragments of the STREAM.  SAVEA / SAVEB: the source is saved while it is walked (rowA / rowB = the slot's rows,
// voff = this lane's byte offset; unconditional raw-buffer stores, see ActSaveX).  sc: the lane's operand scale (see above).
template <int NT, int NSA, int NSB, bool SAVEA = false, bool SAVEB = false, int NA, int NB>
__device__ __forceinline__ void mma_layer_h2t(WeightPipeX& p, const f32x16 (&srcA)[NA], const f32x16 (&srcB)[NB], f32x16 (&acc)[NT], xu32x4 (&q)[X_AHEAD],
                                              float sc, SaveRowX rowA = SaveRowX{}, SaveRowX rowB = SaveRowX{}, uint32_t voff = 0) {
  static_assert((NSA + 1) / 2 <= NA && (NSB + 1) / 2 <= NB, "source too small");
  constexpr int NS = NSA + NSB;
  static_assert(((NS * NT * 2) % X_STAGE_FRAGS) == 0 && ((NS * NT * 2) % X_AHEAD) == 0 && NT % 2 == 0, "layer: whole stages, whole queue turns, tile pairs");
  // Row stores leave in BURSTS of G = 4 k-steps' pieces: 8 stores = 256 contiguous bytes per point, issued by the prepare() of the group's first
  // k-step (the layer's whole source sits in registers, so WHEN a piece leaves is free to choose).  Round 5, profiles/r5/row_store_experiments.txt:
  // the kernel is bound by how fast the L2 can drain these rows to HBM -- with every store landing in an L2-resident window the twin runs at its
  // no-store speed (4.27 ms per 2^20 points against 6.35), half the store INSTRUCTIONS change nothing, non-temporal stores double the time (the L2's
  // merging of the 32-byte pieces is essential) -- and that drain is faster when a line's pieces arrive together: two stores per k-step (rounds 3-4)
  // 6.28 ms, G = 2: 5.85, G = 4: 5.68, G = 8: 5.76, G = 16: 5.88; one contiguous 8 KB block per wave and burst (tile-blocked rows): no better.
  constexpr int G = H2T_ROW_BURST;
  // consume one fragment of the stream: returns it, refills the queue, keeps the ring going (f = the fragment's index in the layer)
  auto take = [&](int f) {
    const int slot = f % X_STAGE_FRAGS;
    const xu32x4 w = q[f % X_AHEAD];
    if (slot % (X_STAGE_FRAGS / X_PIECES) == 0) p.issue_piece(slot / (X_STAGE_FRAGS / X_PIECES));
    q[f % X_AHEAD] = p.read_slot(slot + X_AHEAD);
    if (slot == X_STAGE_FRAGS - 1) p.advance();
    return w;
  };
  // the operands of k-step s: written BEFORE the MFMAs of k-step s - 1, so the split and the two row stores sit in their shadow
  auto prepare = [&](int s, BOpH& o) {
    float v[8];
    const int ss = s < NSA ? s : s - NSA;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = s < NSA ? srcA[s < NSA ? ss >> 1 : 0][8 * (ss & 1) + e] : srcB[s < NSA ? 0 : ss >> 1][8 * (ss & 1) + e];
    if ((s < NSA ? SAVEA : SAVEB) && ss % G == 0) {
      const SaveRowX& row = s < NSA ? rowA : rowB;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (ss + g < (s < NSA ? NSA : NSB)) {
          const int t = ss + g;
          float w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = s < NSA ? srcA[s < NSA ? t >> 1 : 0][8 * (t & 1) + e] : srcB[s < NSA ? 0 : t >> 1][8 * (t & 1) + e];
          const xu32x4 lo = {__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3])};
          const xu32x4 hi = {__float_as_uint(w[4]), __float_as_uint(w[5]), __float_as_uint(w[6]), __float_as_uint(w[7])};
          __builtin_amdgcn_raw_buffer_store_b128(lo, row.rs, (int)(voff + 64u * t), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(hi, row.rs, (int)(voff + 64u * t + 32u), 0, 0);
        }
      }
    }
    // The split: the inference core's two-instruction v_fma_mix form (2 VALU per value instead of ~5 as plain conversions).  Round 4 had dropped it
    // here because single tiles came out wrong; round 5 found why with tools/isa_audit.py: in that build ONE builtin MFMA read its B operand one wait
    // state after the asm v_fma_mixhi that completed it (VALU write -> MFMA read needs two, and hipcc does not pad an asm statement's writes) --
    // mlp_core_h2.h "HAZARD".  h2_operands_ready() below closes it by construction.
#pragma unroll
    for (int e = 0; e < 8; ++e) h2_split_first(o.b1, e, v[e], sc);
#pragma unroll
    for (int e = 0; e < 8; ++e) h2_split_second(o.b1, o.b2, e, v[e], sc);
  };
  BOpH b;
  prepare(0, b);
  h2_operands_ready(b.b1, b.b2);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    BOpH nb = b;
    if (s + 1 < NS) prepare(s + 1, nb);
#pragma unroll
    for (int T = 0; T < NT; T += 2) {   // two tiles at a time: consecutive MFMAs belong to different accumulators
      const int f = (s * NT + T) * 2;
      const xu32x4 u1 = take(f), u2 = take(f + 1);
      const xu32x4 w1 = take(f + 2), w2 = take(f + 3);
      acc[T] = CRNERF_MFMA_H(u2, b.b1, acc[T]);               // small terms first
      acc[T + 1] = CRNERF_MFMA_H(w2, b.b1, acc[T + 1]);
      acc[T] = CRNERF_MFMA_H(u1, b.b2, acc[T]);
      acc[T + 1] = CRNERF_MFMA_H(w1, b.b2, acc[T + 1]);
      acc[T] = CRNERF_MFMA_H(u1, b.b1, acc[T]);
      acc[T + 1] = CRNERF_MFMA_H(w1, b.b1, acc[T + 1]);
    }
    b = nb;
    h2_operands_ready(b.b1, b.b2);
  }
}

// act = max(acc * 2^-8, floor) -- the layer's finished output -- and the range guard's running max |activation| (floor = -inf: a linear layer)
template <int NT, int NDST>
__device__ __forceinline__ void finish_act_h2t(const f32x16 (&acc)[NT], f32x16 (&act)[NDST], float floor_, float& amax) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      typedef float pkf2 __attribute__((ext_vector_type(2)));
      pkf2 v = {acc[t][r], acc[t][r + 1]};
      v = v * pkf2{H2_INV, H2_INV};                 // one v_pk_mul_f32 for the pair (hipcc leaves the scalar form alone; same bits, 1.5 % of the twin)
      act[t][r] = fmaxf(v[0], floor_);
      act[t][r + 1] = fmaxf(v[1], floor_);
      amax = fmaxf(fmaxf(amax, fabsf(act[t][r])), fabsf(act[t][r + 1]));
    }
}

// One 32-point tile through one model, keeping what the backward needs (SV = ActSaveX).  Arguments and results as mlp_core_h2.h's mlp_tile_x3
// (which dispatches here when SV::on); a point whose operands left fp16's range comes out as NaN (65 outputs), its saved rows are then garbage:
// crnerf_render_rays_train_f32x3 in repair mode renders such rays again, rows included.
template <class SV>
__device__ __forceinline__ void mlp_tile_h2t(WeightPipeX& p, int model, const f32x16 (&pe)[3], const f32x16 (&dv)[1], f32x16 (&feat)[2], float& sigma,
                                             int h, xu32x4 (&q)[X_AHEAD], PhaseTimer& tm, const SV& sv) {
  const lds_float* C = (const lds_float*)(p.lds + (model ? LDS_CONST1 : LDS_CONST0));
  const float NEG_INF = -__builtin_huge_valf();
  f32x16 act[8], acc[8];
  constexpr bool SAVE = SV::on;
  const uint32_t vo = sv.offset();
  float amax = 0.0f;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(pe[t][r]));
#pragma unroll
  for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(dv[0][r]));
  tm.tick(T_PROLOGUE);

  init_acc<8>(acc, C + C_BIAS, h);                         // xyz_encoding_1 (the pack's consts carry the 2^8)
  mma_layer_h2t<8, KS_XYZ, 0>(p, pe, pe, acc, q, 1.0f);
  finish_act_h2t<8>(acc, act, 0.0f, amax);
  sv.template masks<8>(0, act);
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {                            // xyz_encoding_2..4 (their input h_l is saved in slot l - 1 on the way)
    tm.tick(T_X7);
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    tm.tick(T_X0);
    mma_layer_h2t<8, KS_HID, 0, SAVE, false>(p, act, act, acc, q, 1.0f, sv.row(l - 1), SaveRowX{}, vo);
    tm.tick(T_X1);
    finish_act_h2t<8>(acc, act, 0.0f, amax);
    tm.tick(T_X2);
    sv.template masks<8>(l, act);
    tm.tick(T_X3);
  }
  init_acc<8>(acc, C + C_BIAS + 4 * W_HIDDEN, h);          // xyz_encoding_5 = Linear(cat[xyz, h])
  mma_layer_h2t<8, KS_XYZ, KS_HID, false, SAVE>(p, pe, act, acc, q, 1.0f, SaveRowX{}, sv.row(3), vo);
  finish_act_h2t<8>(acc, act, 0.0f, amax);
  sv.template masks<8>(4, act);
#pragma unroll 1
  for (int l = 5; l < 8; ++l) {                            // xyz_encoding_6..8
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_layer_h2t<8, KS_HID, 0, SAVE, false>(p, act, act, acc, q, 1.0f, sv.row(l - 1), SaveRowX{}, vo);
    finish_act_h2t<8>(acc, act, 0.0f, amax);
    sv.template masks<8>(l, act);
  }
  tm.tick(T_MMA);
  {                                                        // static_sigma: 256 -> 1 on the VALU (fp32, unscaled weights)
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 w = *(const __attribute__((address_space(3))) f32x4*)(C + C_WSIG + 32 * t + 8 * qq + 4 * h);
        s = fmaf(w[0], act[t][4 * qq + 0], s);
        s = fmaf(w[1], act[t][4 * qq + 1], s);
        s = fmaf(w[2], act[t][4 * qq + 2], s);
        s = fmaf(w[3], act[t][4 * qq + 3], s);
      }
    s += __shfl_xor(s, 32);
    sigma = softplus_ref(s + C[C_BSIG]);
    tm.tick(T_SIGMA);
  }
  init_acc<8>(acc, C + C_BFIN, h);                         // xyz_encoding_final (no activation); h8 -> slot 7
  mma_layer_h2t<8, KS_HID, 0, SAVE, false>(p, act, act, acc, q, 1.0f, sv.row(7), SaveRowX{}, vo);
  finish_act_h2t<8>(acc, act, NEG_INF, amax);
  {
    f32x16 acc4[4];                                        // dir_encoding = relu(Linear(cat[final, dir])); final -> slot 8
    init_acc<4>(acc4, C + C_BDIR, h);
    mma_layer_h2t<4, KS_HID, KS_DIR, SAVE, false>(p, act, dv, acc4, q, 1.0f, sv.row(8), SaveRowX{}, vo);
    finish_act_h2t<4>(acc4, act, 0.0f, amax);
    {
      f32x16 a4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a4[t] = act[t];
      sv.template masks<4>(9, a4);
    }
  }
  {
    f32x16 acc2[2];                                        // static_rgb = sigmoid(Linear); the dir activation -> slot 9
    init_acc<2>(acc2, C + C_BRGB, h);
    mma_layer_h2t<2, KS_HALF, 0, SAVE, false>(p, act, act, acc2, q, 1.0f, sv.row(9), SaveRowX{}, vo);
    tm.tick(T_MMA);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) feat[t][r] = sigmoid_ref(acc2[t][r] * H2_INV);
  }
  sv.note_range(p.lds, amax);   // the pass's largest |operand| so far: what the f16x2 weight gradients scale their activation operand with
  {   // an operand left fp16's range somewhere in this point's MLP: NaN out, not a finite wrong answer
    float am = fmaxf(amax, __shfl_xor(amax, 32));
    if (!(am < H2_ACT_LIMIT)) {
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

// What the kernels built on the h2 core call (the x3 core's name, so that render_fused_x3.hip / mlp_forward_x3.hip compile against either core):
// the lazy-epilogue inference tile, or -- when the caller saves rows -- the training form.
template <class SV = NoSaveX>
__device__ __forceinline__ void mlp_tile_x3(WeightPipeX& p, int model, const f32x16 (&pe)[3], const f32x16 (&dv)[1], f32x16 (&feat)[2], float& sigma,
                                            int h, xu32x4 (&q)[X_AHEAD], PhaseTimer& tm, const SV& sv = SV()) {
  if constexpr (SV::on) mlp_tile_h2t(p, model, pe, dv, feat, sigma, h, q, tm, sv);
  else mlp_tile_h2i(p, model, pe, dv, feat, sigma, h, q, tm);
}

}  // inline namespace xcore_h2
}  // namespace crnerf
