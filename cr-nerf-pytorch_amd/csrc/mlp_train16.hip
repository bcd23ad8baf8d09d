// Training twins of NeRF_sigma.forward (models/nerf.py:157-182): forward that saves the layer
// activations, backward-data (dgrad) on the transposed weight stream, and weight/bias gradients as
// point-reduction GEMMs.  In the reference all of this is PyTorch autograd over 11 addmm nodes.
//
//   forward-train : mlp_tile16 + stores of h1..h8, final, dir_act           acts[10][P][256] (+ 256 relu bits per point and layer)
//   backward-data : per 16-point tile, delta stays in registers exactly like the forward activations
//                   (D[in][point] = W^T[in][out] . delta[out][point]); relu masks come from the saved bits;
//                   every layer's delta is stored                           deltas[10][P][256], d_rgb[P][64], d_sig[P]
//   wgrad         : dW[m][n] = sum_p delta[p][m] * a[p][n]  (fp32 MFMA 32x32x2, k = points),
//                   point-chunked partials + deterministic reduce, written straight in the reference
//                   [out,in] layout (saved activations are in reference feature order, so no un-pack);
//                   db[m] = sum_p delta[p][m]
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include "kernels.h"
#include "mlp_core16.h"
#include "mlp_train16.h"

namespace crnerf {

__global__ __launch_bounds__(512, 2) void mlp_forward_train16_kernel(const char* __restrict__ packed, const float* __restrict__ x,
                                                                     float* __restrict__ out, float* __restrict__ acts, long P, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  load_consts(lds, packed, packed);
  WeightPipe16 pipe;
  pipe.start(lds, packed + CONST_BYTES, packed + CONST_BYTES, 1, 1, lane, wave);
  f32x4 q[V16_AHEAD];
  pipe.prime(q);
  PhaseTimer tm;
  tm.start(false);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * V16_WAVES + wave;
    const long n = tile * 16 + p;
    const bool valid = n < P;
    const float* row = x + (valid ? n : 0) * IN_DIM;
    f32x4 pe[6], feat[4];
    DirRegs dreg;
    f32x4 (&dv)[2] = dreg.v;
    int gg = g;
    asm volatile("" : "+v"(gg));   // keep the 32 lane-group-dependent column selects inside the loop (hoisted, they are spilled)
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = posenc_slot_to_col16(16 * v + 4 * gg + r, XYZ_FREQS);
        pe[v][r] = (valid && c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
      }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = posenc_slot_to_col16(16 * v + 4 * gg + r, DIR_FREQS);
        dv[v][r] = (valid && c >= 0) ? row[XYZ_DIM + (c < 0 ? 0 : c)] : 0.0f;
      }
    float sigma;
    ActSaver sv{acts, P, n, valid, g};
    mlp_tile16(pipe, 0, pe, dreg, feat, sigma, g, q, tm, sv);
    if (valid) {
      float* o = out + n * OUT_DIM;
#pragma unroll
      for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[16 * T + 4 * g + r] = feat[T][r];
      if (g == 0) o[FEAT_DIM] = sigma;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------- backward-data
template <int NT>
__device__ __forceinline__ void zero_acc16(f32x4 (&acc)[NT]) {
#pragma unroll
  for (int T = 0; T < NT; ++T) acc[T] = f32x4{0, 0, 0, 0};
}

// delta_in[T] = mask * acc[T] (mask: this layer's relu-activity bits, see mask_slot); store to the deltas slot
template <int NT, bool MASK>
__device__ __forceinline__ void finish_delta(const f32x4 (&acc)[NT], f32x4 (&dl)[16], unsigned long long bits, float* delta_row, bool valid) {
  const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    f32x4 d = acc[T];
    if (MASK) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * T + r;
        const uint32_t on = ((k < 32 ? lo >> k : hi >> (k - 32)) & 1u);
        d[r] = on ? d[r] : 0.0f;
      }
    }
    dl[T] = d;
    if (delta_row && valid) *(f32x4*)(delta_row + 16 * T) = d;     // delta_row == null: the caller defers the stores (DeltaSaver)
  }
}

// The deltas of a layer are stored BEHIND the MFMAs of the next (transposed) layer, one 64-byte-per-point piece every four
// fragment steps, for the reason given at ActSaver (mlp_train16.h): a burst of 16 stores per wave behind every layer stalls
// all eight waves of the workgroup on the CU's store path.
struct DeltaSaver {
  bool valid;
  __device__ __forceinline__ void piece(float* row, int T, const f32x4& v) const {
    if (valid) *(f32x4*)(row + 16 * T) = v;
  }
};

__global__ __launch_bounds__(512, 2) void mlp_backward16_kernel(const char* __restrict__ packedT, const float* __restrict__ out,
                                                                const float* __restrict__ d_out, const float* __restrict__ acts,
                                                                float* __restrict__ deltas, float* __restrict__ d_rgb, float* __restrict__ d_sig,
                                                                long P, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  load_consts(lds, packedT, packedT);   // the consts block carries the sigma-head weights
  const lds_float* C = (const lds_float*)(lds + LDS_CONST0);
  WeightPipe16 pipe;
  pipe.stages_per_pass = STAGEST_PER_PASS;
  pipe.start(lds, packedT + CONST_BYTES, packedT + CONST_BYTES, 1, 1, lane, wave);
  f32x4 q[V16_AHEAD];
  pipe.prime(q);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * V16_WAVES + wave;
    const long n = tile * 16 + p;
    const bool valid = n < P;
    const long nn = valid ? n : 0;
    const float* orow = out + nn * OUT_DIM;
    const float* grow = d_out + nn * OUT_DIM;
    unsigned long long bits[ACT_SLOTS];   // all ten layers' relu bits of this lane's 64 features: one load batch per tile
#pragma unroll
    for (int sl = 0; sl < ACT_SLOTS; ++sl) bits[sl] = (valid && sl != 8) ? *mask_slot((float*)acts, P, sl, nn, g) : 0ull;
    auto del_row = [&](int slot) { return deltas + ((long)slot * P + nn) * ACT_W + 4 * g; };

    f32x4 dl[16], acc[16];
    // static_rgb: sigmoid'            nerf.py:154,180
    f32x4 drgb[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float f = orow[16 * T + 4 * g + r], gf = valid ? grow[16 * T + 4 * g + r] : 0.0f;
        drgb[T][r] = gf * f * (1.0f - f);
      }
      if (valid) *(f32x4*)(d_rgb + n * FEAT_DIM + 16 * T + 4 * g) = drgb[T];
    }
    // static_sigma: softplus' = sigmoid(pre) = 1 - exp(-sigma)      nerf.py:146,172
    const float sg = orow[FEAT_DIM];
    const float dsp = valid ? grow[FEAT_DIM] * (1.0f - expf(-sg)) : 0.0f;
    if (valid && g == 0) d_sig[n] = dsp;

    {  // through static_rgb^T -> dir_encoding output (relu)
      f32x4 acc8[8];
      zero_acc16<8>(acc8);
      mma_layer16<8, 4, 0>(pipe, drgb, drgb, acc8, q);
      finish_delta<8, true>(acc8, dl, bits[9], nullptr, valid);
    }
    const DeltaSaver ds{valid};
    float* pending = del_row(9);                       // the row whose pieces ride in the next layer's MFMA loop
    zero_acc16<16>(acc);                               // through dir_encoding^T[:, :256] -> xyz_encoding_final output (linear)
    mma_layer16<16, 8, 0>(pipe, dl, dl, acc, q, DeferredActs<DeltaSaver, 8>{ds, pending, dl});
    finish_delta<16, false>(acc, dl, 0ull, nullptr, valid);
    pending = del_row(8);
    zero_acc16<16>(acc);                               // through xyz_encoding_final^T, + sigma head -> h8 (relu)
    mma_layer16<16, 16, 0>(pipe, dl, dl, acc, q, DeferredActs<DeltaSaver, 16>{ds, pending, dl});
#pragma unroll
    for (int T = 0; T < 16; ++T) {
      const f32x4 w = *(const __attribute__((address_space(3))) f32x4*)(C + C_WSIG + 16 * T + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[T][r] = fmaf(w[r], dsp, acc[T][r]);
    }
    finish_delta<16, true>(acc, dl, bits[7], nullptr, valid);
    pending = del_row(7);
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {                     // through xyz_encoding_{l+1}^T -> h_l (relu); l+1 = 8..2
      zero_acc16<16>(acc);
      mma_layer16<16, 16, 0>(pipe, dl, dl, acc, q, DeferredActs<DeltaSaver, 16>{ds, pending, dl});
      unsigned long long bl = bits[0];   // bits[l - 1] by selects: a dynamic register index would go through M0, which the
#pragma unroll                           // LDS-DMA asm (glds16) rewrites behind the compiler's back
      for (int t = 1; t < 7; ++t) bl = (l - 1 == t) ? bits[t] : bl;
      finish_delta<16, true>(acc, dl, bl, nullptr, valid);
      pending = del_row(l - 1);
    }
#pragma unroll
    for (int T = 0; T < 16; ++T) ds.piece(pending, T, dl[T]);   // the first layer's deltas: nothing left to hide them behind
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------- wgrad: C[m][n] = sum_p D[p][m] * A[p][n]
struct WgradJob {
  const float* D; int ldd; int M;      // deltas  [P, ldd], M columns used
  const float* A; int lda; int N;      // inputs  [P, lda], N columns used
  float* partial;                      // [nchunk][M][N]
  float* bias_partial;                 // [nchunk][M] column sums of D (bias gradient) or null
  long P; int chunk;
  int bf16;                            // != 0: the full 256 x 256 tiles multiply bf16-rounded operands (fp32 accumulate), see wgrad_kernel
  const uint32_t* dmax;                // wgrad_h2_kernel: the bits of max |D| over the whole tensor (written by the h2 data gradient), or null
  const uint32_t* amax;                // wgrad_h2_kernel: the saved state's range word (bits of the h2 forward's largest |operand|, kernels.h), or null
};

__device__ __forceinline__ uint32_t pk_bf16_rne(float a, float b) {   // v_cvt_pk_bf16_f32: a -> low half, b -> high half
  typedef __bf16 pkbf16x2 __attribute__((ext_vector_type(2)));
  const pkbf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}

// packed two-piece fp16 split of a point pair (low half x0, high half x1): w1 = fp16(x s), w2 = fp16(x s - w1) -- the arithmetic of split_stage_h
// below, left to hipcc's scheduler.  An MFMA may read the results only behind pieces_ready() (two wait states after the last VALU write: hipcc does
// not count an asm statement as one -- mlp_core_h2.h "HAZARD").
__device__ __forceinline__ void split_pair_h(float x0, float x1, float s, uint32_t& w1, uint32_t& w2) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(w1) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(w1) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(w2) : "v"(x0), "v"(s), "v"(w1));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(w2) : "v"(x1), "v"(s), "v"(w1));
}
typedef _Float16 wh16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void pieces_ready(wh16x8_t& a, wh16x8_t& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
// the scale of a delta tensor's fp16 pieces from its range word (see the full-tile f16x2 branch of wgrad_body): s = 2^(140 - e), inv = 1 / s
__device__ __forceinline__ void delta_scale_h(const uint32_t* dmax, float& s, float& inv) {
  uint32_t eb = (__builtin_nontemporal_load(dmax) >> 23) & 0xffu;
  eb = __builtin_amdgcn_readfirstlane(eb < 32u ? 32u : (eb > 254u ? 254u : eb));
  s = __uint_as_float((267u - eb) << 23);
  inv = __uint_as_float((eb - 13u) << 23);
}

// the scale of the ACTIVATION operand's fp16 pieces (round 6): s = 2^(14 - e) with e the exponent of the largest |operand| the h2 forward
// multiplied in this pass (amax: the saved state's range word), so that the largest scaled activation sits in [2^14, 2^15) and an activation
// 2^-39 of it still lands on a piece bit.  Up to round 5 activations went in unscaled: below 2^-14 a piece is an fp16 subnormal, so an activation of
// 2^-10 kept 15 bits and a column of uniformly small activations (or a scene in small coordinates) lost what the bound promised.  No word, a
// word of zero (the fp32 / f32x3 twins wrote the rows) or Inf (a point left fp16's range in the forward): scale 1.  Whatever the word says, the
// stream's own range watch decides whether the chunk's sums stand.
__device__ __forceinline__ void act_scale_h(const uint32_t* amax, float& s, float& inv) {
  uint32_t eb = amax ? (__builtin_nontemporal_load(amax) >> 23) & 0xffu : 0u;
  eb = __builtin_amdgcn_readfirstlane(eb);
  if (eb == 0u || eb == 255u) { s = 1.0f; inv = 1.0f; return; }
  eb = eb < 103u ? 103u : (eb > 165u ? 165u : eb);          // 2^-24 <= max < 2^39: s in [2^-24, 2^38]
  s = __uint_as_float((268u - eb) << 23);                   // 2^(141 - eb): max * s in [2^14, 2^15)
  inv = __uint_as_float((eb - 14u) << 23);
}

// MODE: 0 fp32 / single-piece bf16 (j.bf16), 1 bf16x3, 2 f16x2 where the job carries a range word (bf16x3 otherwise and as its fallback)
template <int MT, int NT, int MODE = 0>
__device__ __forceinline__ void wgrad_partial_tiles(const WgradJob& j, f32x16 (&acc)[4][4], int m0, int n0, long p0, long p1, int i, int kk,
                                                    bool bias_wave, int bx, int bz) {
  // k-steps (2 points each) per iteration = prefetch depth: the small blocks are bandwidth-bound, keep more rows in flight
  constexpr int KS = MT * NT >= 8 ? 4 : (MT * NT >= 4 ? 8 : 16);
  int dcol[MT], acol[NT];
  float dmask[MT], amask[NT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { const int c = m0 + 32 * t + i; dcol[t] = c < j.M ? c : j.M - 1; dmask[t] = c < j.M ? 1.0f : 0.0f; }
#pragma unroll
  for (int t = 0; t < NT; ++t) { const int c = n0 + 32 * t + i; acol[t] = c < j.N ? c : j.N - 1; amask[t] = c < j.N ? 1.0f : 0.0f; }
  const long plast = p1 - 1;
  constexpr bool X3 = MODE != 0;
  if (j.bf16 || X3) {
    // opt-in mixed precision (CRNERF_BWD_WGRAD_BF16), narrow blocks: the same dword-per-lane operand loads, eight points per lane
    // and k-step of 16 points, rounded to bf16 in registers and multiplied on v_mfma_f32_32x32x16_bf16 (see the full-tile path in
    // wgrad_kernel); bias sums from the un-rounded deltas
    typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
    float dcu[8][MT], acu[8][NT], dnx[8][MT], anx[8][NT];
    auto fetch16 = [&](long pb, float (&d)[8][MT], float (&a)[8][NT]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const long pt = pb + 8 * kk + e;
        const long pc = pt < plast ? pt : plast;
        const float keep = pt < p1 ? 1.0f : 0.0f;
        const float* dr = j.D + pc * j.ldd;
        const float* ar = j.A + pc * j.lda;
#pragma unroll
        for (int t = 0; t < MT; ++t) d[e][t] = dr[dcol[t]] * (dmask[t] * keep);
#pragma unroll
        for (int t = 0; t < NT; ++t) a[e][t] = ar[acol[t]] * amask[t];
      }
    };
    const bool do_bias16 = j.bias_partial && bz == 0 && bias_wave;
    float bs16[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) bs16[t] = 0.0f;
    bool done_h = false;
    if constexpr (MODE == 2) {
      if (j.dmax) {
        // CRNERF_BWD_WGRAD_F16X2 on the narrow blocks: two-piece fp16 splits, three MFMAs per tile; range, scale and the bf16x3 fallback as in the
        // full-tile branch of wgrad_body
        float sdl, sinv, one, ainv;                       // (`one`: the activation operand's scale -- 1 up to round 5, act_scale_h since)
        delta_scale_h(j.dmax, sdl, sinv);
        act_scale_h(j.amax, one, ainv);
        float amx = 0.0f, dmx = 0.0f;
        fetch16(p0, dcu, acu);
        for (long pb = p0; pb < p1; pb += 16) {
          fetch16(pb + 16, dnx, anx);
          wh16x8_t a1[NT], a2[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            uint32_t w1[4], w2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              split_pair_h(acu[2 * q][t], acu[2 * q + 1][t], one, w1[q], w2[q]);
              amx = fmaxf(fmaxf(amx, fabsf(acu[2 * q][t])), fabsf(acu[2 * q + 1][t]));
            }
            a1[t] = __builtin_bit_cast(wh16x8_t, make_uint4(w1[0], w1[1], w1[2], w1[3]));
            a2[t] = __builtin_bit_cast(wh16x8_t, make_uint4(w2[0], w2[1], w2[2], w2[3]));
            pieces_ready(a1[t], a2[t]);
          }
#pragma unroll
          for (int a = 0; a < MT; ++a) {
            uint32_t w1[4], w2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              split_pair_h(dcu[2 * q][a], dcu[2 * q + 1][a], sdl, w1[q], w2[q]);
              dmx = fmaxf(fmaxf(dmx, fabsf(dcu[2 * q][a])), fabsf(dcu[2 * q + 1][a]));
            }
            wh16x8_t d1 = __builtin_bit_cast(wh16x8_t, make_uint4(w1[0], w1[1], w1[2], w1[3]));
            wh16x8_t d2 = __builtin_bit_cast(wh16x8_t, make_uint4(w2[0], w2[1], w2[2], w2[3]));
            pieces_ready(d1, d2);
            if (do_bias16) {
#pragma unroll
              for (int e = 0; e < 8; ++e) bs16[a] += dcu[e][a];
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
              f32x16 c = acc[a][b];
              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(d2, a1[b], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, a2[b], c, 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, a1[b], c, 0, 0, 0);
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int t = 0; t < MT; ++t) dcu[e][t] = dnx[e][t];
#pragma unroll
            for (int t = 0; t < NT; ++t) acu[e][t] = anx[e][t];
          }
        }
        const bool in_range = amx * one < 65504.0f && dmx * sdl < 65504.0f;
        if (__builtin_amdgcn_ballot_w64(!in_range) == 0ull) {
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = (acc[a][b] * ainv) * sinv;     // two exact power-of-two factors, one after the other (their product may leave fp32's range)
          done_h = true;
        } else {
#pragma unroll
          for (int a = 0; a < MT; ++a) {
            bs16[a] = 0.0f;
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
          }
        }
      }
    }
    if (!done_h) {
    fetch16(p0, dcu, acu);
    for (long pb = p0; pb < p1; pb += 16) {
      fetch16(pb + 16, dnx, anx);
      if constexpr (X3) {   // CRNERF_BWD_WGRAD_BF16X3 on the narrow blocks: three-piece splits, six MFMAs per tile (see the full-block path)
        auto split3 = [&](float x0, float x1, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
          w1 = pk_bf16_rne(x0, x1);
          const float r0 = x0 - __uint_as_float(w1 << 16), r1 = x1 - __uint_as_float(w1 & 0xffff0000u);
          w2 = pk_bf16_rne(r0, r1);
          const float s0 = r0 - __uint_as_float(w2 << 16), s1 = r1 - __uint_as_float(w2 & 0xffff0000u);
          w3 = pk_bf16_rne(s0, s1);
        };
        pbf16x8 a1[NT], a2[NT], a3[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          uint32_t w1[4], w2[4], w3[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) split3(acu[2 * q][t], acu[2 * q + 1][t], w1[q], w2[q], w3[q]);
          a1[t] = __builtin_bit_cast(pbf16x8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
          a2[t] = __builtin_bit_cast(pbf16x8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
          a3[t] = __builtin_bit_cast(pbf16x8, make_uint4(w3[0], w3[1], w3[2], w3[3]));
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          uint32_t w1[4], w2[4], w3[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) split3(dcu[2 * q][a], dcu[2 * q + 1][a], w1[q], w2[q], w3[q]);
          const pbf16x8 d1 = __builtin_bit_cast(pbf16x8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
          const pbf16x8 d2 = __builtin_bit_cast(pbf16x8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
          const pbf16x8 d3 = __builtin_bit_cast(pbf16x8, make_uint4(w3[0], w3[1], w3[2], w3[3]));
          if (do_bias16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs16[a] += dcu[e][a];
          }
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            f32x16 c = acc[a][b];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d3, a1[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, a3[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, a2[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, a1[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, a2[b], c, 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, a1[b], c, 0, 0, 0);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int t = 0; t < MT; ++t) dcu[e][t] = dnx[e][t];
#pragma unroll
          for (int t = 0; t < NT; ++t) acu[e][t] = anx[e][t];
        }
        continue;
      }
      pbf16x8 df[MT], af[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        union { pbf16x2 h[4]; pbf16x8 v8; } u;
#pragma unroll
        for (int q = 0; q < 4; ++q) u.h[q] = pbf16x2{(__bf16)dcu[2 * q][t], (__bf16)dcu[2 * q + 1][t]};
        df[t] = u.v8;
        if (do_bias16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bs16[t] += dcu[e][t];
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        union { pbf16x2 h[4]; pbf16x8 v8; } u;
#pragma unroll
        for (int q = 0; q < 4; ++q) u.h[q] = pbf16x2{(__bf16)acu[2 * q][t], (__bf16)acu[2 * q + 1][t]};
        af[t] = u.v8;
      }
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[a], af[b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int t = 0; t < MT; ++t) dcu[e][t] = dnx[e][t];
#pragma unroll
        for (int t = 0; t < NT; ++t) acu[e][t] = anx[e][t];
      }
    }
    }
    if (do_bias16) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        bs16[t] += __shfl_xor(bs16[t], 32);
        if (kk == 0 && m0 + 32 * t + i < j.M) j.bias_partial[(long)bx * j.M + m0 + 32 * t + i] = bs16[t];
      }
    }
    float* out16 = j.partial + (long)bx * j.M * j.N;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kk, nn = n0 + 32 * b + i;
          if (m < j.M && nn < j.N) out16[(long)m * j.N + nn] = acc[a][b][r];
        }
    return;
  }
  float dc[KS][MT], ac[KS][NT], dn[KS][MT], an[KS][NT];
  auto fetch = [&](long pb, float (&d)[KS][MT], float (&a)[KS][NT]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const long pt = pb + 2 * s + kk;
      const long pc = pt < plast ? pt : plast;
      const float keep = pt < p1 ? 1.0f : 0.0f;
      const float* dr = j.D + pc * j.ldd;
      const float* ar = j.A + pc * j.lda;
#pragma unroll
      for (int t = 0; t < MT; ++t) d[s][t] = dr[dcol[t]] * (dmask[t] * keep);
#pragma unroll
      for (int t = 0; t < NT; ++t) a[s][t] = ar[acol[t]] * amask[t];
    }
  };
  const bool do_bias = j.bias_partial && bz == 0 && bias_wave;
  float bs[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) bs[t] = 0.0f;
  fetch(p0, dc, ac);
  for (long pb = p0; pb < p1; pb += 2 * KS) {
    fetch(pb + 2 * KS, dn, an);               // next iteration's operands fly while this one's MFMAs run
    if (do_bias) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) bs[t] += dc[s][t];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[s][a], ac[s][b], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int t = 0; t < MT; ++t) dc[s][t] = dn[s][t];
#pragma unroll
      for (int t = 0; t < NT; ++t) ac[s][t] = an[s][t];
    }
  }
  if (do_bias) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      bs[t] += __shfl_xor(bs[t], 32);
      if (kk == 0 && m0 + 32 * t + i < j.M) j.bias_partial[(long)bx * j.M + m0 + 32 * t + i] = bs[t];
    }
  }
  float* out = j.partial + (long)bx * j.M * j.N;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kk, nn = n0 + 32 * b + i;
        if (m < j.M && nn < j.N) out[(long)m * j.N + nn] = acc[a][b][r];
      }
}

// Workgroup tile 256(M) x 256(N): 4 waves (one per SIMD, 512-register budget) x 128x128 = 4x4 MFMA tiles each.
// Per k-step (2 points) a wave loads 4 delta fragments + 4 input fragments (one dword per lane: a 32-float
// row segment per half-wave, straight from HBM/L2 in MFMA operand shape) for 16 MFMAs; the next
// iteration's operands are in flight while the current MFMAs run.  Tiles beyond M or N are skipped.
// (bx, by, bz): the workgroup's chunk / row-block / column-block inside job j -- blockIdx for a single-job launch, decoded from a flat
// block index by the batched launch below.

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The three-piece bf16 split of column T of eight fp32 rows, cut into eleven micro-stages of FOUR independent VALU instructions each (one per
// point pair): stage K + 1 reads what stage K wrote, so with an MFMA between two stages no instruction ever waits for its predecessor --
// hipcc's own schedule of the same arithmetic runs each pair's chain (cvt -> shift / and -> subtract -> cvt ...) to its end before the next.
//   w[0] = bf16(x), r = x - w[0], w[1] = bf16(r), r' = r - w[1], w[2] = bf16(r');   dword q of a piece = points 2q (low half), 2q + 1 (high half)
// (inline asm, one statement per instruction: written as C++ the stages do not survive -- instcombine turns `w << 16` into a second conversion
// and sinks the masks into the subtractions' stage, so half the MFMA gaps get eight instructions and the others none)
template <int K, int T>
__device__ __forceinline__ void split_stage(const float __attribute__((ext_vector_type(4))) (&v)[8], uint32_t (&w)[3][4], float (&r)[8], float (&u)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (K == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[0][q]) : "v"(v[2 * q][T]), "v"(v[2 * q + 1][T]));
    else if constexpr (K == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[2 * q]) : "v"(w[0][q]));
    else if constexpr (K == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[2 * q + 1]) : "v"(w[0][q]));
    else if constexpr (K == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[2 * q]) : "v"(v[2 * q][T]), "v"(u[2 * q]));
    else if constexpr (K == 4) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[2 * q + 1]) : "v"(v[2 * q + 1][T]), "v"(u[2 * q + 1]));
    else if constexpr (K == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[1][q]) : "v"(r[2 * q]), "v"(r[2 * q + 1]));
    else if constexpr (K == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[2 * q]) : "v"(w[1][q]));
    else if constexpr (K == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[2 * q + 1]) : "v"(w[1][q]));
    else if constexpr (K == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[2 * q]) : "v"(u[2 * q]));
    else if constexpr (K == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[2 * q + 1]) : "v"(u[2 * q + 1]));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[2][q]) : "v"(r[2 * q]), "v"(r[2 * q + 1]));
  }
}
template <int T>
__device__ __forceinline__ void split_all(const float __attribute__((ext_vector_type(4))) (&v)[8], uint32_t (&w)[3][4], float (&r)[8], float (&u)[8]) {
  static_for<11>([&](auto K) { split_stage<decltype(K)::value, T>(v, w, r, u); });
}

// The two-piece fp16 split of column T of eight fp32 rows (the h2 core's arithmetic, mlp_core_h2.h h2_split_*): h1 = fp16(x * s), h2 = fp16(x * s - h1),
// one v_fma_mix per half -- four micro-stages of four independent instructions (one per point pair; dword q = points 2q low, 2q + 1 high).
template <int K, int T>
__device__ __forceinline__ void split_stage_h(const float __attribute__((ext_vector_type(4))) (&v)[8], uint32_t (&w)[2][4], float s) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (K == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(w[0][q]) : "v"(v[2 * q][T]), "v"(s));
    else if constexpr (K == 1) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(w[0][q]) : "v"(v[2 * q + 1][T]), "v"(s));
    else if constexpr (K == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(w[1][q]) : "v"(v[2 * q][T]), "v"(s), "v"(w[0][q]));
    else asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(w[1][q]) : "v"(v[2 * q + 1][T]), "v"(s), "v"(w[0][q]));
  }
}
template <int T>
__device__ __forceinline__ void split_all_h(const float __attribute__((ext_vector_type(4))) (&v)[8], uint32_t (&w)[2][4], float s) {
  static_for<4>([&](auto K) { split_stage_h<decltype(K)::value, T>(v, w, s); });
}
// m[0..3] = max(m, |eight values of two rows|): four independent v_max3_f32 (one slot of the stream)
__device__ __forceinline__ void max_slot(float (&m)[4], const float __attribute__((ext_vector_type(4)))& r0, const float __attribute__((ext_vector_type(4)))& r1) {
  asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m[0]) : "v"(r0[0]), "v"(r0[1]));
  asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m[1]) : "v"(r0[2]), "v"(r0[3]));
  asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m[2]) : "v"(r1[0]), "v"(r1[1]));
  asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m[3]) : "v"(r1[2]), "v"(r1[3]));
}

// MODE 1: the "bf16x3" weight gradients (CRNERF_BWD_WGRAD_BF16X3); MODE 2: the "f16x2" form of the full tiles with bf16x3 behind it -- see the
// full-tile branch below; MODE 4: MODE 2 for jobs without a full tile, compiled without that branch (wgrad_h2_narrow_kernel: half the registers,
// two workgroups per CU)
template <int MODE = 0>
__device__ __forceinline__ void wgrad_body(const WgradJob& j, const int bx, const int by, const int bz) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the wave's tile origin lives in SGPRs
  const int i = lane & 31, kk = lane >> 5;
  // The workgroup's 256 x 256 block holds tm x tn live 32 x 32 MFMA tiles (<= 8 x 8).  Its four waves are laid out 2 x 2,
  // 1 x 4 or 4 x 1 over them -- whichever keeps the most waves busy and, among those, gives the busiest wave the fewest
  // tiles: 256 x 256 layers -> 2 x 2 of 4 x 4 tiles; 256 x 93 (embedding blocks) -> 4 x 1 of 2 x 3; 128 x 256 -> 1 x 4 of
  // 4 x 2; 128 x 27 -> 4 x 1 of 1 x 1; 64 x 128 and 1 x 256 -> 1 x 4.  (With a fixed 2 x 2 layout the narrow blocks ran
  // on two or one of the four SIMDs.)
  const int tm = (j.M - (int)by * 256 + 31) / 32 < 8 ? (j.M - (int)by * 256 + 31) / 32 : 8;
  const int tn = (j.N - (int)bz * 256 + 31) / 32 < 8 ? (j.N - (int)bz * 256 + 31) / 32 : 8;
  int wm = 2, wn = 2, best = -1;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int cm = c == 0 ? 1 : (c == 1 ? 2 : 4), cn = 4 / cm;   // ties go to 1 x 4 (measured: 128 x 256 runs 669 us as 1 x 4, 809 us as 2 x 2)
    const int pm = (tm + cm - 1) / cm, pn = (tn + cn - 1) / cn;          // tiles per wave
    if (pm > 4 || pn > 4) continue;
    const int active = ((tm + pm - 1) / pm) * ((tn + pn - 1) / pn);
    const int score = active * 64 - pm * pn;
    if (score > best) { best = score; wm = cm; wn = cn; }
  }
  const int wmi = wm == 2 ? (wave & 1) : (wm == 4 ? wave : 0), wni = wm == 2 ? (wave >> 1) : (wn == 4 ? wave : 0);
  const int pm = (tm + wm - 1) / wm, pn = (tn + wn - 1) / wn;
  const int m0 = by * 256 + wmi * pm * 32;
  const int n0 = bz * 256 + wni * pn * 32;
  const long p0 = (long)bx * j.chunk;
  const long p1 = p0 + j.chunk < j.P ? p0 + j.chunk : j.P;
  const int mt = tm - wmi * pm < pm ? tm - wmi * pm : pm;                 // live 32-row tiles of this wave (<= 0: none)
  const int nt = tn - wni * pn < pn ? tn - wni * pn : pn;
  const bool bias_wave = wni == 0;                                        // one wave per row block sums the bias gradient
  if (mt <= 0 || nt <= 0) return;
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  if (MODE != 4 && mt == 4 && nt == 4 && m0 + 128 <= j.M && n0 + 128 <= j.N && (j.ldd & 3) == 0 && (j.lda & 3) == 0) {
    // ---- full 128x128 wave tile (every 256-wide layer): branch-free stream.  Column mapping: MFMA tile t, lane i <->
    // column 4i + t, so a lane's four operands of a point are ONE 16-byte load (512 contiguous bytes per half-wave) and
    // the whole k-step is 2 x global_load_dwordx4 + 16 MFMAs.  (The guarded generic loop below puts a branch around
    // every load and MFMA; hipcc then waits vmcnt(0) at each join, i.e. for the prefetch it has just issued.)
    const float* dbase = j.D + m0 + 4 * i;
    const float* abase = j.A + n0 + 4 * i;
    const long plast = p1 - 1;
    if constexpr (MODE != 0) {
      // ---- opt-in "bf16x3" (CRNERF_BWD_WGRAD_BF16X3): fp32-ACCURATE products on the bf16 matrix cores.  Every fp32 operand is split in
      // registers into three bf16 pieces, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (8 + 8 + 8 mantissa
      // bits: the split is exact up to the last bit or two), and a product d * a is the sum of the SIX leading piece products
      // d1 a1 + (d1 a2 + d2 a1) + (d2 a2 + d1 a3 + d3 a1); the three that are dropped are <= 3 x 2^-24 of it -- the size of one fp32 rounding.
      // Piece products are exact in fp32 (8 x 8 bits) and accumulate in the MFMA's fp32 accumulator like the products of the fp32 MFMA do.
      // Why: the fp32 MFMA peaks at 157 TFLOP/s, the bf16 MFMA at 2.5 PFLOP/s -- six bf16 MFMAs per k-step of 16 points are 96 x 32 cycles
      // against 128 x 64 for the same points in fp32, so the kernel runs at the rate of its operand reads (2 KB per point and layer) like
      // the single-piece path below, with none of its rounding noise.  The small terms are added first.
      typedef __bf16 xbf16x8_t __attribute__((ext_vector_type(8)));
      f32x4 dcur[8], acur[8], dnxt[8], anxt[8];
      auto fetch16 = [&](long pb, f32x4 (&d)[8], f32x4 (&a)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const long pt = pb + 8 * kk + e;
          const long pc = pt < plast ? pt : plast;
          d[e] = *(const f32x4*)(dbase + pc * j.ldd);      // raw: rows past the chunk are zeroed where the registers are handed to the
          a[e] = *(const f32x4*)(abase + pc * j.lda);      // next k-step (take16) -- a multiply here would wait for the load just issued
        }
      };
      auto take16 = [&](long pb) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dcur[e] = dnxt[e] * (pb + 8 * kk + e < p1 ? 1.0f : 0.0f);
          acur[e] = anxt[e];
        }
      };
      // column t of eight points -> the three piece fragments (dword q = points 2q, 2q + 1)
      auto split3 = [&](const f32x4 (&v)[8], int t, xbf16x8_t& f1, xbf16x8_t& f2, xbf16x8_t& f3) {
        uint32_t w1[4], w2[4], w3[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x0 = v[2 * q][t], x1 = v[2 * q + 1][t];
          w1[q] = pk_bf16_rne(x0, x1);
          const float r0 = x0 - __uint_as_float(w1[q] << 16), r1 = x1 - __uint_as_float(w1[q] & 0xffff0000u);
          w2[q] = pk_bf16_rne(r0, r1);
          const float s0 = r0 - __uint_as_float(w2[q] << 16), s1 = r1 - __uint_as_float(w2[q] & 0xffff0000u);
          w3[q] = pk_bf16_rne(s0, s1);
        }
        f1 = __builtin_bit_cast(xbf16x8_t, make_uint4(w1[0], w1[1], w1[2], w1[3]));
        f2 = __builtin_bit_cast(xbf16x8_t, make_uint4(w2[0], w2[1], w2[2], w2[3]));
        f3 = __builtin_bit_cast(xbf16x8_t, make_uint4(w3[0], w3[1], w3[2], w3[3]));
      };
      const bool do_bias3 = j.bias_partial && bz == 0 && bias_wave;
      f32x4 bsum3 = {0.0f, 0.0f, 0.0f, 0.0f};
      bool done_h = false;
      if constexpr (MODE == 2) {
        if (j.dmax && ((p1 - p0) & 31) == 0 && p1 > p0) {
          // ---- "f16x2" (the default behind the h2 data gradient, launch_mlp_backward): the same stream with every operand split into TWO fp16 pieces,
          // x * s = h1 + h2 (11 + 11 mantissa bits, the h2 core's split: one v_fma_mix per half), and a product formed from the THREE leading piece
          // products d2 a1 + d1 a2 + d1 a1 on v_mfma_f32_32x32x16_f16 -- half the MFMAs and a third of the split work of bf16x3 (measured with a
          // tuning build that issued half the MFMAs, profiles/r5/wgrad_half_work.txt: the thirteen jobs of a backward 7.4 -> 6.1-6.3 ms per 2^20
          // points), what is dropped is d2 a2 <= 2^-22 of the product.  fp16 has five exponent bits, so the operands need a range: activations go
          // in under the power of two of the pass's largest |operand| (act_scale_h: the h2 forward's range word; without one as they are, the h2
          // forward's own limit being |a| < 65504), deltas under ONE power of two per tensor, s = 2^(140 - e) with e the exponent
          // of the tensor's largest |delta| (j.dmax, written by the h2 data gradient's workgroups: max s |delta| in [2^13, 2^14)); a delta 2^-38 of
          // that maximum still lands on a piece bit.  The accumulators run in the scaled domain and are scaled back (exactly) before they leave.
          // Nothing is trusted: the stream keeps the largest |a| and |delta| it has seen (v_max3 in its spare slots), and a wave that saw an operand
          // leave fp16's range -- rows of a ray the forward had to repair, a stale or missing range word -- throws its sums away and runs its chunk
          // again on the bf16x3 stream below: same result as wgrad_x3_kernel, bit for bit.
          typedef _Float16 xh16x8_t __attribute__((ext_vector_type(8)));
          float sd, sinv, one, ainv;                                      // 2^(140 - e): the largest scaled delta in [2^13, 2^14); 1 / sd; `one`: the
          delta_scale_h(j.dmax, sd, sinv);                                // activation operand's scale (1 up to round 5, act_scale_h since) and its inverse
          act_scale_h(j.amax, one, ainv);
          const uint32_t rowd = (uint32_t)j.ldd * 4u, rowa = (uint32_t)j.lda * 4u;
          uint32_t vd[8], va[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { vd[e] = (uint32_t)(8 * kk + e) * rowd + 16u * i; va[e] = (uint32_t)(8 * kk + e) * rowa + 16u * i; }
          const char* db = (const char*)(j.D + m0) + p0 * (long)rowd;
          const char* ab = (const char*)(j.A + n0) + p0 * (long)rowa;
          const long sd_b = 16L * rowd, sa_b = 16L * rowa;
          f32x4 draw[2][8], araw[8];
          uint32_t Ah[2][4][2][4], Dh[3][2][4];                          // piece dwords: [buffer][column][piece][dword], [set][piece][dword]
          float amx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dmx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          long left = (p1 - p0) / 16;
#pragma unroll
          for (int e = 0; e < 8; ++e) { draw[0][e] = *(const f32x4*)(db + vd[e]); araw[e] = *(const f32x4*)(ab + va[e]); }
          split_all_h<0>(araw, Ah[0][0], one); split_all_h<1>(araw, Ah[0][1], one); split_all_h<2>(araw, Ah[0][2], one); split_all_h<3>(araw, Ah[0][3], one);
          split_all_h<0>(draw[0], Dh[0], sd);
          max_slot(amx, araw[0], araw[1]); max_slot(amx, araw[2], araw[3]); max_slot(amx, araw[4], araw[5]); max_slot(amx, araw[6], araw[7]);
          auto fragh = [](const uint32_t (&w)[4]) { return __builtin_bit_cast(xh16x8_t, make_uint4(w[0], w[1], w[2], w[3])); };
          // One phase = the 12 MFMAs of delta column A against the four activation columns of buffer C (every accumulator: d2 a1, d1 a2, d1 a1, the
          // four accumulators taking turns); behind MFMA g rides micro-stage g of split X (g < 4), g - 4 of split Y (g < 8), or one slot of the tail
          // (two bias rows, two range slots), and a scheduling fence.
          auto phase = [&](auto A_, auto DS_, auto C_, auto&& x_stage, auto&& y_stage, auto&& tail) {
            constexpr int a = decltype(A_)::value, ds = decltype(DS_)::value, c = decltype(C_)::value;
            static_for<12>([&](auto G) {
              constexpr int g = decltype(G)::value, pp = g / 4, b = g % 4;
              constexpr int dp = pp == 0 ? 1 : 0, ap = pp == 1 ? 1 : 0;
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fragh(Dh[ds][dp]), fragh(Ah[c][b][ap]), acc[a][b], 0, 0, 0);
              if constexpr (g < 4) x_stage(G);
              else if constexpr (g < 8) y_stage(std::integral_constant<int, g - 4>{});
              else tail(std::integral_constant<int, g - 8>{});
              __builtin_amdgcn_sched_barrier(0);
            });
          };
          auto kstep = [&](auto CUR) {
            constexpr int c = decltype(CUR)::value, n = 1 - c;
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>; using IC = std::integral_constant<int, c>;
            left -= 1;
            const long adv = left > 0 ? 1 : 0;
            db += adv * sd_b;
            ab += adv * sa_b;
#pragma unroll
            for (int e = 0; e < 8; ++e) { draw[n][e] = *(const f32x4*)(db + vd[e]); araw[e] = *(const f32x4*)(ab + va[e]); }
            __builtin_amdgcn_sched_barrier(0);
            // the same rotation as the bf16x3 k-step: delta pieces through three sets, the next k-step's activation rows first touched in phase 1
            phase(I0{}, I0{}, IC{}, [&](auto K) { split_stage_h<decltype(K)::value, 1>(draw[c], Dh[1], sd); },
                  [&](auto K) { split_stage_h<decltype(K)::value, 2>(draw[c], Dh[2], sd); },
                  [&](auto R) { constexpr int r = decltype(R)::value; if constexpr (r < 2) bsum3 += draw[c][r]; else max_slot(dmx, draw[c][2 * (r - 2)], draw[c][2 * (r - 2) + 1]); });
            phase(I1{}, I1{}, IC{}, [&](auto K) { split_stage_h<decltype(K)::value, 0>(araw, Ah[n][0], one); },
                  [&](auto K) { split_stage_h<decltype(K)::value, 1>(araw, Ah[n][1], one); },
                  [&](auto R) { constexpr int r = decltype(R)::value; if constexpr (r < 2) bsum3 += draw[c][2 + r]; else max_slot(amx, araw[2 * (r - 2)], araw[2 * (r - 2) + 1]); });
            phase(I2{}, I2{}, IC{}, [&](auto K) { split_stage_h<decltype(K)::value, 3>(draw[c], Dh[1], sd); },
                  [&](auto K) { split_stage_h<decltype(K)::value, 2>(araw, Ah[n][2], one); },
                  [&](auto R) { constexpr int r = decltype(R)::value; if constexpr (r < 2) bsum3 += draw[c][4 + r]; else max_slot(dmx, draw[c][4 + 2 * (r - 2)], draw[c][4 + 2 * (r - 2) + 1]); });
            phase(I3{}, I1{}, IC{}, [&](auto K) { split_stage_h<decltype(K)::value, 3>(araw, Ah[n][3], one); },
                  [&](auto K) { split_stage_h<decltype(K)::value, 0>(draw[n], Dh[0], sd); },
                  [&](auto R) { constexpr int r = decltype(R)::value; if constexpr (r < 2) bsum3 += draw[c][6 + r]; else max_slot(amx, araw[4 + 2 * (r - 2)], araw[4 + 2 * (r - 2) + 1]); });
          };
          while (left > 0) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
          }
          const float am = fmaxf(fmaxf(amx[0], amx[1]), fmaxf(amx[2], amx[3])) * one, dm = fmaxf(fmaxf(dmx[0], dmx[1]), fmaxf(dmx[2], dmx[3])) * sd;
          const bool in_range = am < 65504.0f && dm < 65504.0f;          // (an Inf or a NaN among the operands fails it too)
          if (__builtin_amdgcn_ballot_w64(!in_range) == 0ull) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 4; ++b) acc[a][b] = (acc[a][b] * ainv) * sinv;     // two exact power-of-two factors, one after the other
            done_h = true;
          } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
            bsum3 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
        }
      }
      if (!done_h) {
      if (((p1 - p0) & 31) == 0 && p1 > p0) {
        // ---- the stream for whole pairs of k-steps (every chunk but a ragged last one), placed by hand, because one wave per SIMD hides nothing by
        // itself: hipcc's schedule of the loop below is [~530 VALU: addresses, splits] then [96 MFMAs] in clumps -- 2.8 us per k-step where the
        // MFMAs alone are 1.6 (1.9 GHz).  Here a k-step is four phases of 24 MFMAs (one delta column each) and behind EVERY MFMA ride four
        // independent VALU instructions of a split (split_stage: inline asm, fenced by sched_barrier -- the stream is emitted exactly as
        // written: M vvvv x 96, no s_nop, no moves); the rows of the NEXT k-step are requested at the top of a k-step and first touched in phase 1.
        // Row addresses are a uniform base (SGPR pair, advanced per k-step) plus eight per-lane byte offsets computed once -- no 64-bit
        // multiplies in the loop.  Two k-steps per trip so that the double-buffered pieces are compile-time registers.  Same products, same
        // accumulation order per accumulator as the loop below: same bits.  Measured (2^20 points, 13 jobs, profiles/r4/): 9.1 -> 7.1 ms; a
        // 256 x 256 job 550-645 us = 3.3-3.8 TB/s of operand rows, against 480 us with the rows served from L1 / L2 (a round-4 tuning build that re-read 1,024 rows per chunk: the
        // stream's own time, ~3,600 cycles per k-step for 3,072 of MFMA) -- the kernel now sits where its matrix stream and its HBM rows cost
        // about the same, and what is left is the part of the row latency one k-step of prefetch (all the registers allow) does not cover.
        const uint32_t rowd = (uint32_t)j.ldd * 4u, rowa = (uint32_t)j.lda * 4u;
        uint32_t vd[8], va[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { vd[e] = (uint32_t)(8 * kk + e) * rowd + 16u * i; va[e] = (uint32_t)(8 * kk + e) * rowa + 16u * i; }
        const char* db = (const char*)(j.D + m0) + p0 * (long)rowd;
        const char* ab = (const char*)(j.A + n0) + p0 * (long)rowa;
        const long sd = 16L * rowd, sa = 16L * rowa;
        f32x4 draw[2][8], araw[8];
        uint32_t Aw[2][4][3][4], Dw[3][3][4];                          // piece dwords: [buffer][column][piece][dword], [set][piece][dword]
        float sr[8], su[8];                                            // the split's temporaries
        long left = (p1 - p0) / 16;                                    // k-steps still to multiply (even)
#pragma unroll
        for (int e = 0; e < 8; ++e) { draw[0][e] = *(const f32x4*)(db + vd[e]); araw[e] = *(const f32x4*)(ab + va[e]); }
        split_all<0>(araw, Aw[0][0], sr, su); split_all<1>(araw, Aw[0][1], sr, su); split_all<2>(araw, Aw[0][2], sr, su); split_all<3>(araw, Aw[0][3], sr, su);
        split_all<0>(draw[0], Dw[0], sr, su);
        auto frag3 = [](const uint32_t (&w)[4]) { return __builtin_bit_cast(xbf16x8_t, make_uint4(w[0], w[1], w[2], w[3])); };
        // One phase = the 24 MFMAs of delta column A (pieces in set DS) against the four activation columns of buffer C, in the order that
        // gives every accumulator its six products small terms first (d3 a1, d1 a3, d2 a2, d2 a1, d1 a2, d1 a1) and lets the four accumulators
        // take turns; behind MFMA g rides micro-stage g of split X (g < 11), g - 11 of split Y (g < 22), or one row of the bias sums, and a
        // scheduling fence -- the stream is emitted as written.
        auto phase = [&](auto A_, auto DS_, auto C_, auto&& x_stage, auto&& y_stage, auto&& tail) {
          constexpr int a = decltype(A_)::value, ds = decltype(DS_)::value, c = decltype(C_)::value;
          static_for<24>([&](auto G) {
            constexpr int g = decltype(G)::value, p = g / 4, b = g % 4;
            constexpr int dp = p == 0 ? 2 : (p == 1 || p >= 4 ? 0 : 1), ap = p == 0 ? 0 : (p == 1 ? 2 : (p == 2 || p == 4 ? 1 : 0));
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag3(Dw[ds][dp]), frag3(Aw[c][b][ap]), acc[a][b], 0, 0, 0);
            if constexpr (g < 11) x_stage(G);
            else if constexpr (g < 22) y_stage(std::integral_constant<int, g - 11>{});
            else tail(std::integral_constant<int, g - 22>{});
            __builtin_amdgcn_sched_barrier(0);
          });
        };
        auto kstep = [&](auto CUR) {
          constexpr int c = decltype(CUR)::value, n = 1 - c;
          using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
          using I3 = std::integral_constant<int, 3>; using IC = std::integral_constant<int, c>;
          // the next k-step's rows (the last k-step re-reads its own: harmless, and the loop stays free of branches)
          left -= 1;
          const long adv = left > 0 ? 1 : 0;
          db += adv * sd;
          ab += adv * sa;
#pragma unroll
          for (int e = 0; e < 8; ++e) { draw[n][e] = *(const f32x4*)(db + vd[e]); araw[e] = *(const f32x4*)(ab + va[e]); }
          __builtin_amdgcn_sched_barrier(0);
          // delta pieces rotate through three sets (no split overwrites pieces the running phase still reads); the bias sums take the two spare
          // slots of every phase, one delta row each.  The next k-step's activation rows are first touched in phase 1.
          phase(I0{}, I0{}, IC{}, [&](auto K) { split_stage<decltype(K)::value, 1>(draw[c], Dw[1], sr, su); },
                [&](auto K) { split_stage<decltype(K)::value, 2>(draw[c], Dw[2], sr, su); }, [&](auto R) { bsum3 += draw[c][decltype(R)::value]; });
          phase(I1{}, I1{}, IC{}, [&](auto K) { split_stage<decltype(K)::value, 0>(araw, Aw[n][0], sr, su); },
                [&](auto K) { split_stage<decltype(K)::value, 1>(araw, Aw[n][1], sr, su); }, [&](auto R) { bsum3 += draw[c][2 + decltype(R)::value]; });
          phase(I2{}, I2{}, IC{}, [&](auto K) { split_stage<decltype(K)::value, 3>(draw[c], Dw[1], sr, su); },
                [&](auto K) { split_stage<decltype(K)::value, 2>(araw, Aw[n][2], sr, su); }, [&](auto R) { bsum3 += draw[c][4 + decltype(R)::value]; });
          phase(I3{}, I1{}, IC{}, [&](auto K) { split_stage<decltype(K)::value, 3>(araw, Aw[n][3], sr, su); },
                [&](auto K) { split_stage<decltype(K)::value, 0>(draw[n], Dw[0], sr, su); }, [&](auto R) { bsum3 += draw[c][6 + decltype(R)::value]; });
        };
        while (left > 0) {
          kstep(std::integral_constant<int, 0>{});
          kstep(std::integral_constant<int, 1>{});
        }
      } else {
      fetch16(p0, dnxt, anxt);
      take16(p0);
      for (long pb = p0; pb < p1; pb += 16) {
        fetch16(pb + 16, dnxt, anxt);
        if (do_bias3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum3 += dcur[e];
        }
        xbf16x8_t a1[4], a2[4], a3[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) split3(acur, t, a1[t], a2[t], a3[t]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          xbf16x8_t d1, d2, d3;
          split3(dcur, a, d1, d2, d3);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            f32x16 c = acc[a][b];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d3, a1[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, a3[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, a2[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, a1[b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, a2[b], c, 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, a1[b], c, 0, 0, 0);
          }
        }
        // ask the scheduler to thread the splits' VALU work between the MFMAs (three behind each; measured 18.29 -> 17.85 ms per 2^20-point backward;
#pragma unroll            // without the hint hipcc emits the MFMAs of a column back to back and the splits in clumps)
        for (int g = 0; g < 96; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        take16(pb + 16);
      }
      }
      }
      if (do_bias3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bsum3[t] += __shfl_xor(bsum3[t], 32);
        if (kk == 0) *(f32x4*)(j.bias_partial + (long)bx * j.M + m0 + 4 * i) = bsum3;
      }
      float* outp3 = j.partial + (long)bx * j.M * j.N;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + a;
          const f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
          *(f32x4*)(outp3 + (long)m * j.N + n0 + 4 * i) = v;
        }
      return;
    }
    if (j.bf16) {
      // ---- opt-in mixed precision (crnerf_mlp_backward_ex_f32, CRNERF_BWD_WGRAD_BF16): the SAME fp32 operands from HBM, rounded to
      // bf16 (RNE) in registers and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  k-step = 16 points; lane
      // (i, kk) supplies points 8kk..8kk+7 of the step for its four columns 4i..4i+3 (one 16-byte load per point and operand,
      // exactly the bytes of the fp32 path).  The matrix work shrinks 16x, so the kernel runs at the HBM rate of its operand
      // reads (2 KB per point and layer).  The bias gradient is summed from the un-rounded deltas.
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      f32x4 dcur[8], acur[8], dnxt[8], anxt[8];
      auto fetch16 = [&](long pb, f32x4 (&d)[8], f32x4 (&a)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const long pt = pb + 8 * kk + e;
          const long pc = pt < plast ? pt : plast;
          d[e] = *(const f32x4*)(dbase + pc * j.ldd);      // raw: rows past the chunk are zeroed where the registers are handed to the
          a[e] = *(const f32x4*)(abase + pc * j.lda);      // next k-step (take16) -- a multiply here would wait for the load just issued
        }
      };
      auto take16 = [&](long pb) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dcur[e] = dnxt[e] * (pb + 8 * kk + e < p1 ? 1.0f : 0.0f);
          acur[e] = anxt[e];
        }
      };
      auto frag = [&](const f32x4 (&v)[8], int t) {
        union { bf16x2_t h[4]; bf16x8_t v8; } u;
#pragma unroll
        for (int q = 0; q < 4; ++q) u.h[q] = bf16x2_t{(__bf16)v[2 * q][t], (__bf16)v[2 * q + 1][t]};   // v_cvt_pk_bf16_f32
        return u.v8;
      };
      const bool do_bias16 = j.bias_partial && bz == 0 && bias_wave;
      f32x4 bsum = {0.0f, 0.0f, 0.0f, 0.0f};
      fetch16(p0, dnxt, anxt);
      take16(p0);
      for (long pb = p0; pb < p1; pb += 16) {
        fetch16(pb + 16, dnxt, anxt);
        if (do_bias16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += dcur[e];
        }
        bf16x8_t df[4], af[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { df[t] = frag(dcur, t); af[t] = frag(acur, t); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[a], af[b], acc[a][b], 0, 0, 0);
        take16(pb + 16);
      }
      if (do_bias16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bsum[t] += __shfl_xor(bsum[t], 32);
        if (kk == 0) *(f32x4*)(j.bias_partial + (long)bx * j.M + m0 + 4 * i) = bsum;
      }
      float* outp16 = j.partial + (long)bx * j.M * j.N;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + a;
          const f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
          *(f32x4*)(outp16 + (long)m * j.N + n0 + 4 * i) = v;
        }
      return;
    }
    constexpr int KF = 4;
    f32x4 dc[KF], ac[KF], dnx[KF], anx[KF];
    auto fetch4 = [&](long pb, f32x4 (&d)[KF], f32x4 (&a)[KF]) {
#pragma unroll
      for (int s = 0; s < KF; ++s) {
        const long pt = pb + 2 * s + kk;
        const long pc = pt < plast ? pt : plast;                      // clamped: always a readable row
        d[s] = *(const f32x4*)(dbase + pc * j.ldd);                   // raw: see take4
        a[s] = *(const f32x4*)(abase + pc * j.lda);
      }
    };
    // hand the prefetched rows to the next k-steps; rows past the chunk contribute nothing (zeroed HERE, behind the MFMAs: a multiply inside
    // fetch4 makes hipcc wait for the delta loads right after issuing them, and the k-step then runs loads and MFMAs one after the other)
    auto take4 = [&](long pb) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KF; ++s) {
        dc[s] = dnx[s] * (pb + 2 * s + kk < p1 ? 1.0f : 0.0f);
        ac[s] = anx[s];
      }
    };
    const bool do_bias4 = j.bias_partial && bz == 0 && bias_wave;
    f32x4 bs4 = {0.0f, 0.0f, 0.0f, 0.0f};
    fetch4(p0, dnx, anx);
    take4(p0);
    for (long pb = p0; pb < p1; pb += 2 * KF) {
      fetch4(pb + 2 * KF, dnx, anx);
      if (do_bias4) {
#pragma unroll
        for (int s = 0; s < KF; ++s) bs4 += dc[s];
      }
#pragma unroll
      for (int s = 0; s < KF; ++s)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[s][a], ac[s][b], acc[a][b], 0, 0, 0);
      take4(pb + 2 * KF);
    }
    if (do_bias4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) bs4[t] += __shfl_xor(bs4[t], 32);
      if (kk == 0) *(f32x4*)(j.bias_partial + (long)bx * j.M + m0 + 4 * i) = bs4;
    }
    float* outp = j.partial + (long)bx * j.M * j.N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + a;
        const f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
        *(f32x4*)(outp + (long)m * j.N + n0 + 4 * i) = v;
      }
    return;
  }
  // ---- partial tiles (N = 93 / 27 embedding blocks, M = 128 / 64 / 1 heads): the same stream with MT x NT live MFMA
  // tiles, MT in {1,2,4}, NT in {1,2,3,4} chosen once per wave; loads are unconditional (clamped row / column, zero mask),
  // so no control flow sits between a prefetch and the MFMAs that hide it
  const int MTs = mt <= 1 ? 1 : (mt == 2 ? 2 : 4), NTs = nt <= 1 ? 1 : (nt == 2 ? 2 : (nt == 3 ? 3 : 4));
#define CRNERF_WG(MT, NT) if (MTs == MT && NTs == NT) { wgrad_partial_tiles<MT, NT, MODE == 4 ? 2 : MODE>(j, acc, m0, n0, p0, p1, i, kk, bias_wave, bx, bz); return; }
  CRNERF_WG(1, 1) CRNERF_WG(1, 2) CRNERF_WG(1, 3) CRNERF_WG(1, 4) CRNERF_WG(2, 1) CRNERF_WG(2, 2) CRNERF_WG(2, 3) CRNERF_WG(2, 4)
  CRNERF_WG(4, 1) CRNERF_WG(4, 2) CRNERF_WG(4, 3) CRNERF_WG(4, 4)
#undef CRNERF_WG
}


__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradJob j) { wgrad_body<0>(j, blockIdx.x, blockIdx.y, blockIdx.z); }
__global__ __launch_bounds__(256, 1) void wgrad_x3_kernel(WgradJob j) { wgrad_body<1>(j, blockIdx.x, blockIdx.y, blockIdx.z); }
__global__ __launch_bounds__(256, 1) void wgrad_h2_kernel(WgradJob j) { wgrad_body<2>(j, blockIdx.x, blockIdx.y, blockIdx.z); }
// The smallest jobs (128 x 27, 64 x 128: one or two accumulator tiles per wave) compiled for 256 registers: two workgroups share a CU, each on half a
// chunk -- twice the row bytes in flight per CU, which is what bounds these jobs (wgrad()).
__global__ __launch_bounds__(256, 2) void wgrad_h2_narrow_kernel(WgradJob j) { wgrad_body<4>(j, blockIdx.x, blockIdx.y, blockIdx.z); }

// Every weight gradient of one NeRF_sigma backward in ONE launch: at the reference's 1,024-ray batches a per-layer launch is
// ~256 workgroups of 256 points each -- fourteen ramp-ups and drains per model, and a [256 chunks][256][256] partial-sum slab
// per layer that costs as much HBM traffic as the operands.  Batched, the jobs share the chip (a job's workgroups are laid
// out consecutively, the big jobs first), so a chunk can be ~9x longer and the partial sums ~9x smaller.
constexpr int WG_MAX_JOBS = 16;
constexpr int WG_RANGE_WORDS = 64;   // launch_mlp_backward: one word per delta slot (max |delta| bits, ACT_SLOTS used), a 256-byte line of the scratch
struct WgradBatch {
  WgradJob job[WG_MAX_JOBS];
  int first[WG_MAX_JOBS + 1];   // first flat block of job k; first[njobs] = grid size
  int nchunk[WG_MAX_JOBS];
  int my[WG_MAX_JOBS];          // row blocks (ceil(M / 256))
  int njobs;
};
template <int MODE>
__device__ __forceinline__ void wgrad_batch_body(const WgradBatch& b) {
  int k = 0;
#pragma unroll 1
  while (k + 1 < b.njobs && (int)blockIdx.x >= b.first[k + 1]) ++k;
  k = __builtin_amdgcn_readfirstlane(k);
  const int local = (int)blockIdx.x - b.first[k];
  const int bx = local % b.nchunk[k], rest = local / b.nchunk[k];
  wgrad_body<MODE>(b.job[k], bx, rest % b.my[k], rest / b.my[k]);
}
__global__ __launch_bounds__(256, 1) void wgrad_batch_kernel(WgradBatch b) { wgrad_batch_body<0>(b); }
__global__ __launch_bounds__(256, 1) void wgrad_x3_batch_kernel(WgradBatch b) { wgrad_batch_body<1>(b); }
__global__ __launch_bounds__(256, 1) void wgrad_h2_batch_kernel(WgradBatch b) { wgrad_batch_body<2>(b); }

struct ReduceJob { const float* partial; const float* bias_partial; float* dst; float* db; int nchunk, M, N, ldc; };
struct ReduceBatch { ReduceJob job[WG_MAX_JOBS]; int first[WG_MAX_JOBS + 1]; int njobs; };
// dst[m*ldc + n] = sum_c partial[c][m][n]   (fixed summation tree: deterministic; 8 loads in flight per thread), and in the
// same launch db[m] = sum_c bias_partial[c][m] (threads M*N .. M*N + M - 1) when db != null
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, int nchunk, int M, int N, float* __restrict__ dst, int ldc,
                                                  const float* __restrict__ bias_partial, float* __restrict__ db, int idx) {
  const float* src = partial;
  long stride = (long)M * N;
  float* out;
  if (idx < M * N) out = dst + (long)(idx / N) * ldc + idx % N;
  else {
    idx -= M * N;
    if (!db || idx >= M) return;
    src = bias_partial; stride = M; out = db + idx;
  }
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int c = 0;
  for (; c + 8 <= nchunk; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += src[(c + u) * stride + idx];
  }
  for (; c < nchunk; ++c) a[0] += src[c * stride + idx];
  *out = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nchunk, int M, int N, float* __restrict__ dst, int ldc,
                                                           const float* __restrict__ bias_partial, float* __restrict__ db) {
  wgrad_reduce_body(partial, nchunk, M, N, dst, ldc, bias_partial, db, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceBatch b) {
  int k = 0;
#pragma unroll 1
  while (k + 1 < b.njobs && (int)blockIdx.x >= b.first[k + 1]) ++k;
  const ReduceJob& j = b.job[k];
  wgrad_reduce_body(j.partial, j.nchunk, j.M, j.N, j.dst, j.ldc, j.bias_partial, j.db, ((int)blockIdx.x - b.first[k]) * 256 + (int)threadIdx.x);
}

int launch_wgrad_reduce(const float* partial, int nchunk, int M, int N, float* dst, int ldc, const float* bias_partial, float* db, hipStream_t st) {
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((M * N + (db ? M : 0) + 255) / 256), dim3(256), 0, st, partial, nchunk, M, N, dst, ldc, bias_partial, db);
  return 0;
}

// points per wgrad workgroup: small enough to fill the chip at 1024-ray batches, large enough that the
// partial-sum workspace (nchunk x M x N) stays ~50 MB per layer at 65k-ray batches
static int wg_chunk(long P) {
  long c = (P + 255) / 256;             // ~one 256x256 workgroup per CU
  c = (c + 31) / 32 * 32;               // whole pairs of 16-point k-steps: the bf16x3 stream's fast path
  return (int)(c < 128 ? 128 : c);
}

// dW (and, when db != null, the bias gradient of the same delta) for one (delta, input-block) pair
// the chunk length wgrad() gives a job: f16x2 jobs of at most two accumulator tiles per wave run on wgrad_h2_narrow_kernel, two workgroups per CU on half chunks
static int wgrad_job_chunk(long P, int M, int N, int bf16) {
  int chunk = wg_chunk(P);
  if (bf16 == 3 && M <= 128 && N <= 128) { chunk = (chunk / 2 + 31) / 32 * 32; if (chunk < 128) chunk = 128; }
  return chunk;
}
size_t wgrad_workspace_floats(long P, int M, int N, int bf16) {
  const int chunk = wgrad_job_chunk(P, M, N, bf16);     // (the narrow f16x2 jobs write up to twice the partial-sum slabs of a full chunk)
  const size_t nchunk = (size_t)((P + chunk - 1) / chunk);
  return nchunk * ((size_t)M * N + M);
}

int wgrad(const float* D, int ldd, int M, const float* A, int lda, int N, float* dst, int ldc, float* db, long P, float* ws,
          hipStream_t st, int bf16, const uint32_t* dmax, const uint32_t* amax) {
  // at most two accumulator tiles per wave (dir_encoding's direction block, static_rgb): wgrad_h2_narrow_kernel, two workgroups per CU on half chunks
  // (measured per 2^19 points: 139 -> 93 and 142 -> 119 us; the 93-column embedding blocks gain nothing from it, dir_encoding's 4 x 2 tiles spill in 256 registers)
  const bool narrow = bf16 == 3 && M <= 128 && N <= 128;
  const int chunk = wgrad_job_chunk(P, M, N, bf16);     // (narrow: 2 x nchunk x (M N + M) stays inside the workspace of a full block; wgrad_workspace_floats(.., 3) sizes it)
  const int nchunk = (int)((P + chunk - 1) / chunk);
  float* bws = ws + (size_t)nchunk * M * N;
  WgradJob j{D, ldd, M, A, lda, N, ws, db ? bws : nullptr, P, chunk, bf16 >= 2 ? 0 : bf16, bf16 == 3 ? dmax : nullptr, bf16 == 3 ? amax : nullptr};   // bf16x3 / f16x2: their own kernels (wgrad_body<1>, <2>)
  if (narrow) hipLaunchKernelGGL(wgrad_h2_narrow_kernel, dim3(nchunk, (M + 255) / 256, (N + 255) / 256), dim3(256), 0, st, j);
  else if (bf16 == 3) hipLaunchKernelGGL(wgrad_h2_kernel, dim3(nchunk, (M + 255) / 256, (N + 255) / 256), dim3(256), 0, st, j);
  else if (bf16 == 2) hipLaunchKernelGGL(wgrad_x3_kernel, dim3(nchunk, (M + 255) / 256, (N + 255) / 256), dim3(256), 0, st, j);
  else hipLaunchKernelGGL(wgrad_kernel, dim3(nchunk, (M + 255) / 256, (N + 255) / 256), dim3(256), 0, st, j);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((M * N + (db ? M : 0) + 255) / 256), dim3(256), 0, st, ws, nchunk, M, N, dst, ldc, bws, db);
  return 0;
}

// workgroups of the batched weight-gradient launch (all jobs together): four per CU -- enough to even out the jobs' different
// lengths, few enough that the partial sums stay ~100 MB whatever the batch size
static int wgrad_batch_blocks() { static const int per_cu = [] { const char* e = getenv("CRNERF_WGRAD_BLOCKS_PER_CU"); return e ? atoi(e) : 4; }(); return per_cu * num_cus(); }
static size_t wgrad_batch_workspace_floats() { return (size_t)(wgrad_batch_blocks() + WG_MAX_JOBS) * (256 * 256 + 256); }
size_t mlp_train_acts_bytes(long P) { return acts_rows_bytes(P) + ACTS_RANGE_BYTES; }   // activations + relu bits + the range word's line (kernels.h)
size_t mlp_train_scratch_bytes(long P) {
  const int chunk = wg_chunk(P);
  const size_t nchunk = (size_t)((P + chunk - 1) / chunk);
  size_t wsf = nchunk * (size_t)(256 * 256 + 256);                 // per-layer launches (CRNERF_WGRAD_BATCH=0)
  if (wgrad_batch_workspace_floats() > wsf) wsf = wgrad_batch_workspace_floats();   // the batched launch keeps every job's partial sums
  return (size_t)ACT_SLOTS * P * ACT_W * 4 + (size_t)P * FEAT_DIM * 4 + (size_t)P * 4 + WG_RANGE_WORDS * 4 + wsf * 4;   // deltas | d_rgb | d_sig | range words | partial sums
}

static int launch_core(const void* fn, int grid, size_t shmem) {
  return ensure_dynamic_lds(fn, shmem, "train kernel");
}

int launch_mlp_forward_train(const void* packed, const float* x, float* out, float* acts, long P, hipStream_t stream) {
  if (P <= 0) return 0;
  const long groups = (P + 127) / 128;
  const int cus = num_cus();
  const int grid = (int)(groups < cus ? groups : cus), iters = (int)((groups + grid - 1) / grid);
  if (int rc = launch_core((const void*)mlp_forward_train16_kernel, grid, LDS_SCRATCH)) return rc;
  if (int rc = zero_acts_range(acts, P, stream)) return rc;        // the fp32 twin tracks no range
  hipLaunchKernelGGL(mlp_forward_train16_kernel, dim3(grid), dim3(512), LDS_SCRATCH, stream, (const char*)packed, x, out, acts, P, iters);
  return check_launch("mlp_forward_train16_kernel");
}

// ---- batched weight gradients -------------------------------------------------------------------------------------------
// specs[k].weight: relative cost of one point of job k (1 = a full 256 x 256 block); a job's chunk length is chosen so that
// every workgroup of the launch carries about the same work.  Jobs may have different point counts (specs[k].P).
static void wgrad_batch_plan(const WgradSpec* specs, int n, long* chunk, int* nchunk) {
  double total = 0.0;
  for (int k = 0; k < n; ++k) total += (double)specs[k].weight * (double)specs[k].P;
  const double per = total / (double)wgrad_batch_blocks();
  for (int k = 0; k < n; ++k) {
    long c = (long)(per / specs[k].weight);
    c = (c + 31) / 32 * 32;
    if (c < 128) c = 128;
    chunk[k] = c;
    nchunk[k] = (int)((specs[k].P + c - 1) / c);
  }
}
float wgrad_job_weight(int M, int N) {
  const float w = (float)(((M + 31) / 32) * ((N + 31) / 32)) / 64.0f;      // live 32 x 32 MFMA tiles of the 256 x 256 block(s)
  return w < 0.1f ? 0.1f : w;
}
size_t wgrad_batch_ws_floats(const WgradSpec* specs, int n) {
  long chunk[WG_MAX_JOBS];
  int nchunk[WG_MAX_JOBS];
  if (n > WG_MAX_JOBS) return 0;
  wgrad_batch_plan(specs, n, chunk, nchunk);
  size_t f = 0;
  for (int k = 0; k < n; ++k) f += (size_t)nchunk[k] * ((size_t)specs[k].M * specs[k].N + (specs[k].db ? specs[k].M : 0));
  return f;
}
int wgrad_batch(const WgradSpec* specs, int n, float* ws, size_t ws_floats, hipStream_t st) {
  if (n > WG_MAX_JOBS) return set_error(-3, "wgrad batch: too many jobs");
  if (n <= 0) return 0;
  long chunk[WG_MAX_JOBS];
  int nchunk[WG_MAX_JOBS];
  wgrad_batch_plan(specs, n, chunk, nchunk);
  int order[WG_MAX_JOBS];
  for (int k = 0; k < n; ++k) order[k] = k;
  for (int a = 1; a < n; ++a)                       // stable insertion sort, longest workgroups first
    for (int c = a; c > 0 && specs[order[c]].weight * chunk[order[c]] > specs[order[c - 1]].weight * chunk[order[c - 1]]; --c) {
      const int t = order[c]; order[c] = order[c - 1]; order[c - 1] = t;
    }
  WgradBatch b;
  ReduceBatch r;
  b.njobs = r.njobs = n;
  int blocks = 0, rblocks = 0;
  float* w = ws;
  bool x3 = false, h2 = false;
  for (int q = 0; q < n; ++q) {
    const WgradSpec& sp = specs[order[q]];
    const int nc = nchunk[order[q]];
    float* bws = w + (size_t)nc * sp.M * sp.N;
    x3 = x3 || sp.bf16 == 2;
    h2 = h2 || sp.bf16 == 3;
    b.job[q] = WgradJob{sp.D, sp.ldd, sp.M, sp.A, sp.lda, sp.N, w, sp.db ? bws : nullptr, sp.P, (int)chunk[order[q]], sp.bf16 >= 2 ? 0 : sp.bf16,
                        sp.bf16 == 3 ? sp.dmax : nullptr, sp.bf16 == 3 ? sp.amax : nullptr};
    b.first[q] = blocks;
    b.nchunk[q] = nc;
    b.my[q] = (sp.M + 255) / 256;
    blocks += nc * ((sp.M + 255) / 256) * ((sp.N + 255) / 256);
    r.job[q] = ReduceJob{w, sp.db ? bws : nullptr, sp.dst, sp.db, nc, sp.M, sp.N, sp.ldc};
    r.first[q] = rblocks;
    rblocks += (sp.M * sp.N + (sp.db ? sp.M : 0) + 255) / 256;
    w = bws + (sp.db ? (size_t)nc * sp.M : 0);
  }
  b.first[n] = blocks;
  r.first[n] = rblocks;
  if ((size_t)(w - ws) > ws_floats) return set_error(-3, "wgrad batch: workspace too small");
  if (h2) hipLaunchKernelGGL(wgrad_h2_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
  else if (x3) hipLaunchKernelGGL(wgrad_x3_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
  else hipLaunchKernelGGL(wgrad_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(rblocks), dim3(256), 0, st, r);
  return 0;
}

// The weight / bias gradients of all eleven nn.Linear from the saved activations and the stored deltas (shared by the fp32 twins and
// the mixed-precision twins of mlp_gemm_bf16.hip).  wb != 0: bf16-operand path (CRNERF_BWD_WGRAD_BF16).  Two launches in all (every
// job in one batched launch + one reduction) at small batches.  (An eight-wave variant of the full-block kernel -- two waves per SIMD,
// 64 x 128 per wave -- was 15 % SLOWER at 2^17 and 2^21 points, and a dedicated full-block kernel with the operand rows of 8 or 12
// k-steps in flight instead of 4 changed nothing: the kernel is short of neither waves nor prefetch depth.  Counters
// (profiles/r3/pmc_train_sq.txt): matrix pipe busy 85.6 % of the kernel's cycles at a shader clock of 2.07 GHz -- the 2.4 TB/s of
// operand traffic costs clock, not issue slots; 0.856 x 2.07 / 2.4 = the measured 74 % of the 157.3 TFLOP/s figure.)
int launch_mlp_wgrads(const float* x, const float* acts, const float* deltas, const float* d_rgb, const float* d_sig, float* ws, float* const* grads,
                      long P, hipStream_t stream, int wb, const uint32_t* dmax, const uint32_t* amax) {
  auto A = [&](int slot) { return acts + (size_t)slot * P * ACT_W; };
  auto D = [&](int slot) { return deltas + (size_t)slot * P * ACT_W; };
  auto R = [&](int slot) { return dmax ? dmax + slot : nullptr; };   // the range word of delta slot `slot` (wb == 3: the f16x2 full tiles)
  if (wb == 3 && !dmax) wb = 2;
  // batched below 2^18 points (the reference's 1,024-ray batches: 8.24 -> 8.02 ms per step, 56 launches -> 4); beyond that a
  // per-layer launch of 256 equal workgroups is already one even wave over the chip and the batched launch's mixed job lengths
  // only add a tail (16,384-ray step: 103 ms per layer, 108 ms batched).  CRNERF_WGRAD_BATCH=0 / 1 forces either.
  static const int batch_mode = [] { const char* e = getenv("CRNERF_WGRAD_BATCH"); return e ? atoi(e) : -1; }();
  const bool batched = batch_mode < 0 ? P <= (1L << 18) : batch_mode != 0;
  if (batched) {
    WgradSpec sp[WG_MAX_JOBS];
    int n = 0;
    // Per-point cost of the narrow jobs relative to a full 256 x 256 block: it sets every job's chunk length so that all workgroups of the launch
    // carry about the same work.  The fp32 / bf16x3 blocks are bound by their matrix work (profiles/r3/train_1024_before_ordered.txt: the first
    // row); the f16x2 blocks sit at the HBM read rate, where a narrow job costs what its rows cost, not what its tiles cost -- with the first row
    // the narrow jobs' workgroups ran long after the full blocks had finished: wgrad_h2_batch_kernel 846 us per launch at 1,024 rays, 643 us with
    // the second (profiles/r6/wgrad_batch_job_weights.txt; CRNERF_WGB_WEIGHTS="emb,dir,dire,rgb,sig" overrides both for measurements).
    struct JobW { float emb, dir, dire, rgb, sig; };
    static const JobW jw_matrix = {0.52f, 0.66f, 0.19f, 0.26f, 0.26f}, jw_rows = {1.1f, 1.1f, 0.7f, 0.7f, 0.8f};
    static const JobW jw_env = [] {
      JobW w = {0, 0, 0, 0, 0};
      const char* e = getenv("CRNERF_WGB_WEIGHTS");
      const bool ok = e && sscanf(e, "%f,%f,%f,%f,%f", &w.emb, &w.dir, &w.dire, &w.rgb, &w.sig) == 5 && w.emb >= 0.05f && w.dir >= 0.05f && w.dire >= 0.05f && w.rgb >= 0.05f && w.sig >= 0.05f;
      return ok ? w : JobW{0, 0, 0, 0, 0};       // anything else than five weights >= 0.05: ignored
    }();
    const JobW& jw = jw_env.emb > 0 ? jw_env : (wb == 3 ? jw_rows : jw_matrix);
    const float W_FULL = 1.0f, W_EMB = jw.emb, W_DIR = jw.dir, W_DIRE = jw.dire, W_RGB = jw.rgb, W_SIG = jw.sig;
    sp[n++] = WgradSpec{D(0), ACT_W, 256, x, IN_DIM, XYZ_DIM, grads[0], XYZ_DIM, grads[1], W_EMB, wb, P, R(0), amax};                         // xyz_encoding_1
    for (int l = 1; l < 8; ++l) {
      if (l == 4) {                                                                                                                  // xyz_encoding_5: cat([xyz, h4]), nerf.py:168-169
        sp[n++] = WgradSpec{D(4), ACT_W, 256, x, IN_DIM, XYZ_DIM, grads[8], XYZ_DIM + 256, grads[9], W_EMB, wb, P, R(4), amax};
        sp[n++] = WgradSpec{D(4), ACT_W, 256, A(3), ACT_W, 256, grads[8] + XYZ_DIM, XYZ_DIM + 256, nullptr, W_FULL, wb, P, R(4), amax};
      } else {
        sp[n++] = WgradSpec{D(l), ACT_W, 256, A(l - 1), ACT_W, 256, grads[2 * l], 256, grads[2 * l + 1], W_FULL, wb, P, R(l), amax};
      }
    }
    sp[n++] = WgradSpec{D(8), ACT_W, 256, A(7), ACT_W, 256, grads[16], 256, grads[17], W_FULL, wb, P, R(8), amax};                            // xyz_encoding_final
    sp[n++] = WgradSpec{d_sig, 1, 1, A(7), ACT_W, 256, grads[18], 256, grads[19], W_SIG, 0, P};                                         // static_sigma
    sp[n++] = WgradSpec{D(9), ACT_W, 128, A(8), ACT_W, 256, grads[20], 256 + DIR_DIM, grads[21], W_DIR, wb, P, R(9), amax};                   // dir_encoding: cat([final, dir])
    sp[n++] = WgradSpec{D(9), ACT_W, 128, x + XYZ_DIM, IN_DIM, DIR_DIM, grads[20] + 256, 256 + DIR_DIM, nullptr, W_DIRE, wb, P, R(9), amax};
    sp[n++] = WgradSpec{d_rgb, FEAT_DIM, FEAT_DIM, A(9), ACT_W, 128, grads[22], 128, grads[23], W_RGB, wb, P, R(ACT_SLOTS), amax};            // static_rgb
    if (int rc = wgrad_batch(sp, n, ws, wgrad_batch_workspace_floats(), stream)) return rc;
    return check_launch("mlp_backward wgrad (batched)");
  }
  // xyz_encoding_1: input x[:, :93]
  wgrad(D(0), ACT_W, 256, x, IN_DIM, XYZ_DIM, grads[0], XYZ_DIM, grads[1], P, ws, stream, wb, R(0), amax);
  for (int l = 1; l < 8; ++l) {
    if (l == 4) {  // xyz_encoding_5: cat([xyz, h4])            nerf.py:168-169
      wgrad(D(4), ACT_W, 256, x, IN_DIM, XYZ_DIM, grads[8], XYZ_DIM + 256, grads[9], P, ws, stream, wb, R(4), amax);
      wgrad(D(4), ACT_W, 256, A(3), ACT_W, 256, grads[8] + XYZ_DIM, XYZ_DIM + 256, nullptr, P, ws, stream, wb, R(4), amax);
    } else {
      wgrad(D(l), ACT_W, 256, A(l - 1), ACT_W, 256, grads[2 * l], 256, grads[2 * l + 1], P, ws, stream, wb, R(l), amax);
    }
  }
  wgrad(D(8), ACT_W, 256, A(7), ACT_W, 256, grads[16], 256, grads[17], P, ws, stream, wb, R(8), amax);      // xyz_encoding_final
  wgrad(d_sig, 1, 1, A(7), ACT_W, 256, grads[18], 256, grads[19], P, ws, stream);                 // static_sigma
  wgrad(D(9), ACT_W, 128, A(8), ACT_W, 256, grads[20], 256 + DIR_DIM, grads[21], P, ws, stream, wb, R(9), amax);  // dir_encoding: cat([final, dir])
  wgrad(D(9), ACT_W, 128, x + XYZ_DIM, IN_DIM, DIR_DIM, grads[20] + 256, 256 + DIR_DIM, nullptr, P, ws, stream, wb, R(9), amax);
  wgrad(d_rgb, FEAT_DIM, FEAT_DIM, A(9), ACT_W, 128, grads[22], 128, grads[23], P, ws, stream, wb, R(ACT_SLOTS), amax);   // static_rgb
  return check_launch("mlp_backward wgrad");
}

// grads: 24 device pointers in crnerf.h tensor order, each overwritten with the gradient of sum(out * d_out)
int launch_mlp_backward(const void* packedT, const float* x, const float* out, const float* d_out, const float* acts, void* scratch,
                        float* const* grads, long P, hipStream_t stream, int flags, const void* packedT_x3, const void* packedT_h2) {
  int wb = (flags & 4) ? 3 : ((flags & 2) ? 2 : (flags & 1));   // CRNERF_BWD_WGRAD_F16X2 (h2 data gradient only, checked by the caller) / _BF16X3 / _BF16
  // CRNERF_BWD_PHASE_DGRAD / _WGRAD (include/crnerf.h): one half of the backward per call, so that the caller can put the halves on two streams
  const bool do_dgrad = (flags & 24) != 16, do_wgrad = (flags & 24) != 8;
  if (P <= 0) return 0;
  float* deltas = (float*)scratch;
  float* d_rgb = deltas + (size_t)ACT_SLOTS * P * ACT_W;
  float* d_sig = d_rgb + (size_t)P * FEAT_DIM;
  uint32_t* dmax = (uint32_t*)(d_sig + P);          // the delta tensors' range words (h2 data gradient -> f16x2 weight gradients)
  float* ws = d_sig + P + WG_RANGE_WORDS;
  const long groups = (P + 127) / 128;
  const int cus = num_cus();
  const int grid = (int)(groups < cus ? groups : cus), iters = (int)((groups + grid - 1) / grid);
  if (!do_dgrad) {
  } else if (packedT_h2) {   // crnerf_mlp_backward_h2_f32: the data gradient on the h2 core (mlp_backward_h2.hip), same scratch layout
    // CRNERF_BWD_WGRAD_F16X2: the full tiles of the weight gradients on the two-piece fp16 form (wgrad_h2_kernel), ranged by the
    // largest |delta| of every tensor, which the data gradient's workgroups leave in dmax (zero = nothing known: the kernel falls back by itself)
    const bool h2w = wb == 3;
    if (h2w && hipMemsetAsync(dmax, 0, WG_RANGE_WORDS * 4, stream) != hipSuccess) return set_error(-1, "mlp_backward: hipMemsetAsync(range words) failed");
    if (int rc = launch_mlp_dgrad_h2(packedT_h2, out, d_out, acts, deltas, d_rgb, d_sig, P, stream, h2w ? dmax : nullptr)) return rc;
    if (packedT_x3)   // its safety net: the same deltas on the scale-free core, only when the h2 pack carries the range flag (device-side test)
      if (int rc = launch_mlp_dgrad_x3(packedT_x3, out, d_out, acts, deltas, d_rgb, d_sig, P, stream, (const int*)packedT_h2 + H2_FLAG_WORD)) return rc;
  } else if (packedT_x3) {   // crnerf_mlp_backward_x3_f32: the data gradient on the x3 core (mlp_backward_x3.hip), same scratch layout
    if (int rc = launch_mlp_dgrad_x3(packedT_x3, out, d_out, acts, deltas, d_rgb, d_sig, P, stream)) return rc;
  } else {
    if (int rc = launch_core((const void*)mlp_backward16_kernel, grid, LDS_SCRATCH)) return rc;
    hipLaunchKernelGGL(mlp_backward16_kernel, dim3(grid), dim3(512), LDS_SCRATCH, stream, (const char*)packedT, out, d_out, acts, deltas, d_rgb, d_sig, P, iters);
    if (int rc = check_launch("mlp_backward16_kernel")) return rc;
  }
  if (!do_wgrad) return 0;
  if (wb == 3 && !packedT_h2) wb = 2;                 // (a PHASE_WGRAD call of the h2 entry arrives with a non-null marker: abi.hip)
  return launch_mlp_wgrads(x, acts, deltas, d_rgb, d_sig, ws, grads, P, stream, wb, wb == 3 ? dmax : nullptr, wb == 3 ? acts_range_word(acts, P) : nullptr);
}

}  // namespace crnerf
