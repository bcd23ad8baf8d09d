// NeRF_sigma forward, fp32-ACCURATE on the bf16 matrix cores ("x3" core): one 32-point tile per wavefront, four wavefronts per
// workgroup (one per SIMD, 512 registers), activations register-resident in fp32.
//
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182, in fp32 -- the arithmetic of mlp_core.h / mlp_core16.h, evaluated
// differently: the fp32 MFMA of this part peaks at 157 TFLOP/s, the bf16 MFMA at 2.5 PFLOP/s, so every product w * a of a Linear layer is
// formed from THREE-PIECE bf16 splits of both fp32 operands,
//     w = w1 + w2 + w3,  w1 = bf16(w), w2 = bf16(w - w1), w3 = bf16(w - w1 - w2)          (8 + 8 + 8 mantissa bits; exact to the last bit or two)
//     w * a ~= w3 a1 + w1 a3 + w2 a2 + w2 a1 + w1 a2 + w1 a1                                (the six leading piece products, small terms first)
// each piece product exact in fp32 and accumulated in the MFMA's fp32 accumulator; the three dropped terms are <= 3 x 2^-24 of the product --
// one fp32 rounding.  Weights are split once, at pack time (layout.h "fragX", three fragments per (k-step, tile)); an activation is split
// once per layer, in registers, when its k-step comes up (44 VALU instructions per 16 k-values, beside 6 x NT MFMAs).  Biases, relu, the
// sigma head (fp32 VALU), softplus / sigmoid and the embeddings (accurate sincosf) are the fp32 code of mlp_core.h.
//
// Structure (that of mlp_core.h, the round-1 fp32 core): lane (p = lane&31, h = lane>>5) owns point p; the 32x32 C/D layout leaves it with
// features 32t + 8q + 4h + j in register 4q+j of tile t, and registers 8(s%2) .. +7 of tile s/2 ARE the lane's eight k-values of k-step s of
// the next layer.  K-outer: the accumulators of all output tiles of a layer are live (128 registers), the contraction is walked once.
// Weights: the 16 KiB-stage LDS ring of xcore_pipe.h (seven slots, dynamic slot counters, LDS-DMA as inline asm); fragments are consumed in
// stream order through an X_AHEAD-deep register queue, so the stage boundaries (16 fragments) need not line up with the k-steps (3 NT fragments).
//
// This is the scale-free fp32-accurate core (bf16 carries fp32's exponent): the fallback of the faster, range-limited h2 core (mlp_core_h2.h).
// What was tried on it and dropped is in DESIGN.md Appendix A (L2 touches ahead of the LDS-DMA, 14 x 8 KiB stages, a deeper fragment queue).
#pragma once
#include "xcore_pipe.h"
#if CRNERF_X_NP != 3
#error "mlp_core_x3.h is the three-piece bf16 core; the two-piece fp16 core is mlp_core_h2.h"
#endif

namespace crnerf {
inline namespace xcore_x3 {

// eight fp32 k-values of a lane -> the three piece operands (dword d = values 2d, 2d + 1; v_cvt_pk_bf16_f32 rounds to nearest even)
__device__ __forceinline__ uint32_t x3_pk(float a, float b) {
  typedef __bf16 pk2 __attribute__((ext_vector_type(2)));
  const pk2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void x3_split(const float (&v)[8], xbf16x8& b1, xbf16x8& b2, xbf16x8& b3) {
  xu32x4 w1, w2, w3;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float x0 = v[2 * d], x1 = v[2 * d + 1];
    w1[d] = x3_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(w1[d] << 16), r1 = x1 - __uint_as_float(w1[d] & 0xffff0000u);
    w2[d] = x3_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(w2[d] << 16), s1 = r1 - __uint_as_float(w2[d] & 0xffff0000u);
    w3[d] = x3_pk(s0, s1);
  }
  b1 = __builtin_bit_cast(xbf16x8, w1);
  b2 = __builtin_bit_cast(xbf16x8, w2);
  b3 = __builtin_bit_cast(xbf16x8, w3);
}

#define CRNERF_MFMA_X(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8, (a)), (b), (c), 0, 0, 0)

// One layer: NT output tiles; k-steps 0..NSA-1 take their B operands from srcA (registers 8(s%2).. of tile s/2), the following NSB from
// srcB.  PAD: stage-padding fragments behind the layer (dir_encoding), skipped through the queue without being multiplied.  q always holds
// the next X_AHEAD fragments of the STREAM.  SAVEA / SAVEB (training twin): source A / B is saved while it is walked -- rowA / rowB = the
// slot's rows, voff = this lane's byte offset.
template <int NT, int NSA, int NSB, int PAD, bool SAVEA = false, bool SAVEB = false, int NA, int NB>
__device__ __forceinline__ void mma_layer_x3(WeightPipeX& p, const f32x16 (&srcA)[NA], const f32x16 (&srcB)[NB], f32x16 (&acc)[NT],
                                             xu32x4 (&q)[X_AHEAD], SaveRowX rowA = SaveRowX{}, SaveRowX rowB = SaveRowX{}, uint32_t voff = 0) {
  static_assert((NSA + 1) / 2 <= NA && (NSB + 1) / 2 <= NB, "source too small");
  constexpr int NS = NSA + NSB;
  static_assert(((NS * NT * XNP + PAD) % STAGE_FRAGS) == 0 && ((NS * NT * XNP + PAD) % X_AHEAD) == 0 && NT % 2 == 0,
                "layer (+ padding) must be whole stages and whole queue turns; tiles go in pairs");
  // consume one fragment of the stream: returns it, refills the queue, keeps the ring going (f = the fragment's index in the layer)
  auto take = [&](int f) {
    const int slot = f % X_STAGE_FRAGS;
    const xu32x4 w = q[f % X_AHEAD];
    if (slot % (X_STAGE_FRAGS / X_PIECES) == 0) p.issue_piece(slot / (X_STAGE_FRAGS / X_PIECES));
    q[f % X_AHEAD] = p.read_slot(slot + X_AHEAD);
    if (slot == X_STAGE_FRAGS - 1) p.advance();
    return w;
  };
  // the split of k-step s + 1 is written BEFORE the MFMAs of k-step s (it does not depend on them): its ~50 VALU instructions and the two
  // row stores of the training twin sit in the shadow of 6 NT MFMAs instead of in front of them
  auto prepare = [&](int s, xbf16x8& b1, xbf16x8& b2, xbf16x8& b3) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ss = s < NSA ? s : s - NSA;
      v[e] = s < NSA ? srcA[s < NSA ? ss >> 1 : 0][8 * (ss & 1) + e] : srcB[s < NSA ? 0 : ss >> 1][8 * (ss & 1) + e];
    }
    if (s < NSA ? SAVEA : SAVEB) {   // training twin: this k-step's sixteen features of the source leave for HBM (two 16-byte pieces per lane)
      const SaveRowX& row = s < NSA ? rowA : rowB;
      const int ss = s < NSA ? s : s - NSA;
      const xu32x4 lo = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      const xu32x4 hi = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
      __builtin_amdgcn_raw_buffer_store_b128(lo, row.rs, (int)(voff + 64u * ss), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(hi, row.rs, (int)(voff + 64u * ss + 32u), 0, 0);
    }
    x3_split(v, b1, b2, b3);
  };
  xbf16x8 b1, b2, b3;
  prepare(0, b1, b2, b3);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    xbf16x8 n1 = b1, n2 = b2, n3 = b3;
    if (s + 1 < NS) prepare(s + 1, n1, n2, n3);
#pragma unroll
    for (int T = 0; T < NT; T += 2) {   // two tiles at a time: consecutive MFMAs belong to different accumulators
      const int f = (s * NT + T) * 3;
      const xu32x4 u1 = take(f), u2 = take(f + 1), u3 = take(f + 2);
      const xu32x4 w1 = take(f + 3), w2 = take(f + 4), w3 = take(f + 5);
      acc[T] = CRNERF_MFMA_X(u3, b1, acc[T]);          // small terms first
      acc[T + 1] = CRNERF_MFMA_X(w3, b1, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u1, b3, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w1, b3, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u2, b2, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w2, b2, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u2, b1, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w2, b1, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u1, b2, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w1, b2, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u1, b1, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w1, b1, acc[T + 1]);
    }
    b1 = n1; b2 = n2; b3 = n3;
  }
#pragma unroll
  for (int f = NS * NT * XNP; f < NS * NT * XNP + PAD; ++f) (void)take(f);
}

// One 32-point tile through one model.  pe / dv: the positional embeddings in the register order of posenc_regs (posenc.h), exactly as
// mlp_core.h's mlp_tile takes them.  Returns feat[t][4q+j] = rgb feature 32t+8q+4h+j of point p, and sigma (both lane halves).
template <class SV = NoSaveX>
__device__ __forceinline__ void mlp_tile_x3(WeightPipeX& p, int model, const f32x16 (&pe)[3], const f32x16 (&dv)[1], f32x16 (&feat)[2], float& sigma,
                                            int h, xu32x4 (&q)[X_AHEAD], PhaseTimer& tm, const SV& sv = SV()) {
  const lds_float* C = (const lds_float*)(p.lds + (model ? LDS_CONST1 : LDS_CONST0));
  const float NEG_INF = -__builtin_huge_valf();
  f32x16 act[8], acc[8];
  constexpr bool SAVE = SV::on;
  const uint32_t vo = sv.offset();
  tm.tick(T_PROLOGUE);

  init_acc<8>(acc, C + C_BIAS, h);                         // xyz_encoding_1
  mma_layer_x3<8, KS_XYZ, 0, 0>(p, pe, pe, acc, q);
  store_act<8>(acc, act, 0.0f);
  sv.template masks<8>(0, act);
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {                            // xyz_encoding_2..4 (their input h_l is saved in slot l - 1 on the way)
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_layer_x3<8, KS_HID, 0, 0, SAVE, false>(p, act, act, acc, q, sv.row(l - 1), SaveRowX{}, vo);
    store_act<8>(acc, act, 0.0f);
    sv.template masks<8>(l, act);
  }
  init_acc<8>(acc, C + C_BIAS + 4 * W_HIDDEN, h);          // xyz_encoding_5 = Linear(cat[xyz, h])
  mma_layer_x3<8, KS_XYZ, KS_HID, 0, false, SAVE>(p, pe, act, acc, q, SaveRowX{}, sv.row(3), vo);
  store_act<8>(acc, act, 0.0f);
  sv.template masks<8>(4, act);
#pragma unroll 1
  for (int l = 5; l < 8; ++l) {                            // xyz_encoding_6..8
    init_acc<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_layer_x3<8, KS_HID, 0, 0, SAVE, false>(p, act, act, acc, q, sv.row(l - 1), SaveRowX{}, vo);
    store_act<8>(acc, act, 0.0f);
    sv.template masks<8>(l, act);
  }
  tm.tick(T_MMA);
  {                                                        // static_sigma: 256 -> 1 on the VALU (fp32, as mlp_core.h)
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
  mma_layer_x3<8, KS_HID, 0, 0, SAVE, false>(p, act, act, acc, q, sv.row(7), SaveRowX{}, vo);
  store_act<8>(acc, act, NEG_INF);
  {
    f32x16 acc4[4];                                        // dir_encoding = relu(Linear(cat[final, dir])); final -> slot 8
    init_acc<4>(acc4, C + C_BDIR, h);
    mma_layer_x3<4, KS_HID, KS_DIR, XPAD_DIR, SAVE, false>(p, act, dv, acc4, q, sv.row(8), SaveRowX{}, vo);
    store_act<4>(acc4, act, 0.0f);
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
    mma_layer_x3<2, KS_HALF, 0, 0, SAVE, false>(p, act, act, acc2, q, sv.row(9), SaveRowX{}, vo);
    tm.tick(T_MMA);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) feat[t][r] = sigmoid_ref(acc2[t][r]);
    tm.tick(T_EPILOGUE);
  }
}

}  // inline namespace xcore_x3
}  // namespace crnerf
