// What the bf16 matrix-core renderer (mlp_core_bf16p.h, the pair core) keeps from the round-1/2 bf16 core that used to live here
// (one 64-point tile per wavefront, one wave per SIMD; CRNERF_BF16_CORE=64 -- removed in round 5): the compile-time fragment
// schedule of the static 4-slot weight ring and its checker, the bf16 packing / activation helpers, and the positional embedding
// straight into bf16 B-operand registers.
//
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182, evaluated in mixed precision: the
// operands of every Linear except static_sigma (weights and input activations, incl. the two positional
// embeddings) are rounded to bf16 (RNE); products accumulate in fp32; biases, relu, softplus, sigmoid and the
// sigma head (fp32 weights on the un-rounded fp32 output of xyz_encoding_8) stay fp32.  oracle/cpu_ref.py
// `mlp_forward_bf16` restates exactly this.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"
#include "mlp_core.h"

namespace crnerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef CRNERF_B_AHEAD
#define CRNERF_B_AHEAD 4
#endif
constexpr int B_AHEAD = CRNERF_B_AHEAD;   // fragments read ahead of the one being multiplied (tuning: -DCRNERF_B_AHEAD=n)
constexpr int B_TAIL = STREAMB_USED - (STAGESB_PER_PASS - 1) * STAGE_FRAGS;  // fragments in the (short) last stage: 8
static_assert(B_TAIL == 8 && STREAMB_USED % B_AHEAD == 0, "tail stage must fit the piece schedule");

// ---- compile-time schedule over the pass-relative fragment index i (0 <= i < STREAMB_USED) -------------
// position in the padded stream (the look-ahead past the last fragment lands in the next pass)
constexpr int b_pos(int i) { return i < STREAMB_USED ? i : i + (STREAMB_FRAGS - STREAMB_USED); }
// The barrier that opens stage c + 1 sits in the LAST k-step of stage c whose look-ahead read still targets stage c (slot
// 15 - B_AHEAD = 11; in the 8-fragment tail stage slot 3): every read of stage c + 1 is issued behind it.  (Rounds 1-2 had it in
// slot 14, i.e. BEHIND the first three look-ahead reads of stage c + 1, which were therefore covered by nothing but the ~1 us that
// had passed since their LDS-DMA was issued: with the weight stream evicted from L2 by other kernels -- the encoder / decoder between
// two renders -- a late piece was read before it landed and single ray quads came out wrong, about once per ten cold full-image
// renders.  tools/render_cold2.py is the reproducer; tests/test_gpu_bf16.py::test_render_cold_l2_is_deterministic the guard.)
constexpr int B_ADV_SLOT = STAGE_FRAGS - 1 - B_AHEAD;
constexpr int B_ADV_SLOT_TAIL = B_TAIL - 1 - B_AHEAD;
static_assert(B_ADV_SLOT >= 10 && B_ADV_SLOT_TAIL >= 3, "the barrier must come after the stage's third LDS-DMA piece (slots 9 / 3)");
constexpr bool b_tail(int i) { return i / STAGE_FRAGS == STAGESB_PER_PASS - 1; }
constexpr bool b_advance_at(int i) { return i % STAGE_FRAGS == (b_tail(i) ? B_ADV_SLOT_TAIL : B_ADV_SLOT); }
// stage advances executed before iteration i's read is issued (= the newest stage that read may touch)
constexpr int b_cur_stage(int i) { return i / STAGE_FRAGS + (i % STAGE_FRAGS > (b_tail(i) ? B_ADV_SLOT_TAIL : B_ADV_SLOT) ? 1 : 0); }
// LDS-DMA piece (0..3) issued at iteration i, or -1: slots 1, 5, 9, 13 of a stage (k-steps that carry one epilogue
// quarter in the 16-k-step layers, so piece + quarter + fragment read stay within 4 fillers per MFMA gap)
constexpr int b_piece_at(int i) {
  const int sl = i % STAGE_FRAGS;
  if (!b_tail(i)) return sl % 4 == 1 ? sl / 4 : -1;
  return sl == 0 ? 0 : (sl == 1 ? 1 : (sl == 3 ? 2 : (sl == 5 ? 3 : -1)));   // tail stage (B_TAIL = 8 fragments)
}
// prefetch-cursor bookkeeping (SALU only) sits in the stage's last k-step, which carries no epilogue work
constexpr bool b_cursor_at(int i) { return b_tail(i) ? i == STREAMB_USED - 1 : i % STAGE_FRAGS == STAGE_FRAGS - 1; }
constexpr bool b_schedule_ok() {
  int pieces = 0, advances = 0, cursors = 0;
  for (int i = 0; i < STREAMB_USED; ++i) {
    if (b_cur_stage(i) != advances) return false;
    const int rs = b_pos(i + B_AHEAD) / STAGE_FRAGS;
    if (rs > advances || rs < i / STAGE_FRAGS) return false;   // EVERY read targets a stage whose barrier has been passed
    if (b_piece_at(i) >= 0) {
      if (b_piece_at(i) != pieces % 4 || cursors != pieces / 4) return false;   // in order, cursor moved before piece 0
      ++pieces;
    }
    if (b_cursor_at(i)) {
      ++cursors;
      if (pieces != 4 * cursors) return false;
    }
    if (b_advance_at(i)) {
      ++advances;
      if (pieces != 4 * advances - 1) return false;   // the barrier follows the stage's third piece: vmcnt(7), see WeightPipeB
    }
  }
  return advances == STAGESB_PER_PASS && cursors == STAGESB_PER_PASS && b_cur_stage(STREAMB_USED - 1) == STAGESB_PER_PASS;
}
static_assert(b_schedule_ok(), "bf16 fragment schedule violates the ring protocol");

// Weight ring of the bf16 core: B_RING = 4 slots x 16 KiB, and because a pass is 76 = 19 x 4 stages the ring is in the SAME
// phase at the start of every tile -- so, with the tile code fully unrolled, every ring quantity is a compile-time
// constant: stage c of a tile lives in slot c % 4, fragment reads are `ds_read_b128 v, rd_base offset:<slot, fragment>`
// (4 x 16 KiB = the 16-bit offset range), the LDS-DMA destination of stage c + 3 is one of four precomputed M0 values.
// What is left per stage is ONE VALU add (the fetch offset, riding in the LDS-DMA instruction's VGPR operand) and the
// M0 write.  Scalar bookkeeping is NOT free beside the MFMA stream (tools/ubench: ~4 cycles per SALU instruction
// wherever it is placed; the dynamic 6-slot ring's 11 instructions per stage cost 2.4 cycles per MFMA).
// Protocol (as mlp_core.h WeightPipe, with distances for 4 slots): while stage c is multiplied, stage c + 3 is fetched
// into the slot stage c - 1 has left (its last fragment read was issued before the barrier that opened stage c); the
// barrier in stage c's k-step 11 -- behind the stage's third piece, in front of the first look-ahead read of stage c + 1 --
// waits vmcnt(7): stage c + 2 and three pieces of stage c + 3 may be in flight, stage c + 1 (issued >= 30 k-steps ago) has
// landed.  Stages 76, 77, 78 are the NEXT tile's first three: they are fetched from `nxt`, the stream of the model the
// next tile runs (begin_tile).
constexpr int B_RING = 4;
static_assert(STAGESB_PER_PASS % B_RING == 0, "the static ring needs a whole number of ring turns per pass");
static_assert(B_RING * STAGE_BYTES <= 65536 && B_RING <= RING_SLOTS, "ds_read offsets are 16 bits; the LDS ring area is shared with the fp32 core");

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};   // v_cvt_pk_bf16_f32 (RNE); element 0 = low half
  return __builtin_bit_cast(uint32_t, v);
}
// ---- epilogues -------------------------------------------------------------------------------------------
// Issue budget (tools/ubench/gen_mfma_stream.py, one wave per SIMD): beside a v_mfma_f32_32x32x16_bf16 up to FOUR
// other instructions per MFMA gap are free (32.7 cycles/MFMA with 8 fillers per k-step split 4 + 4); 10 per k-step cost
// 35.9, 12 cost 38.0, and 6 all in ONE gap 36.7.  So the epilogue of a tile is cut into 16 quarters (quarter qc =
// accumulator registers 2(qc>>1), +1 of point group qc&1) of 4 VALU each -- two accumulator reads in the first gap of
// a k-step, convert + relu in the second -- that run behind the MFMAs of the NEXT tile, one quarter per k-step.  The
// bias costs nothing: it is the C operand of a tile's first two MFMAs (one f32x16 read from LDS, shared by both groups).
// An epilogue object provides
//     prefetch(T)               LDS reads for tile T's epilogue (only the sigma head has any)
//     load(slot, T, qc, acc)    first half of quarter qc: fetch its two accumulator registers
//     finish(slot, T, qc)       second half: convert / activate / store
// The asm statements pin each piece of work to the gap it was written in: instruction selection may emit pure
// arithmetic anywhere between its operands' definitions and its first use, and without them hipcc parks whole
// layers' epilogues (256 live accumulators) behind the layer's last MFMA.
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 lds_f4(const lds_float* p) { return *(const __attribute__((address_space(3))) f32x4*)p; }

// bf16 pair -> relu.  relu commutes with round-to-nearest-even, so it is applied to the packed pair: as int16 a negative
// bf16 (incl. -0) is negative, and max(x, 0) per 16-bit lane is ONE v_pk_max_i16 for two values.
template <bool RELU>
__device__ __forceinline__ uint32_t pack_pair(float v0, float v1) {
  uint32_t pk = pk_bf16(v0, v1);
  if (RELU) {
    asm volatile("" : "+v"(pk));   // keep the pair packed: hipcc otherwise converts the halves separately and re-packs
    const s16x2 z = {0, 0};
    pk = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), z));
  }
  asm volatile("" : "+v"(pk));
  return pk;
}

// 1 / (1 + 2^(-x log2 e)) on the hardware exp2 / rcp (1 ulp each): 4 instructions instead of ~22 for expf + IEEE
// division, 64 of them per lane per tile; the relative error (~2e-7) is far below this path's bf16 noise
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }

// nn.Softplus(beta=1, threshold=20) on the hardware exp2 / log2: ln(1 + e^x) = log2(1 + 2^(x log2 e)) ln 2.  Absolute error
// ~1e-7 (relative ~1e-3 only where sigma < 1e-4, i.e. where the sample is transparent anyway).
__device__ __forceinline__ float softplus_fast(float x) {
  return x > 20.0f ? x : __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x * 1.44269504f)) * 0.69314718f;
}

// ---- positional embedding straight into bf16 B-operand registers ---------------------------------------
// Reference: PosEmbedding.forward, models/nerf.py:17-30.  Lane half h, dword pp of k-step s holds
// (sin, cos)(2^f x_d) of argument a = 8s + 4h + pp (f = a/3, d = a%3); a == 3F -> (x, y); a == 3F+1 -> (z, 0).
// The result is rounded to bf16 (2^-9 relative), so the hardware v_sin_f32 / v_cos_f32 (inputs in revolutions)
// are accurate enough -- PROVIDED the range reduction is done right: 2^14 |x| / 2pi reaches ~1.3e4 revolutions,
// so x/(2pi) is formed as an unevaluated two-float sum (p + e); scaling by 2^f is exact, fract() of the high part
// is exact, and the reduced argument carries ~1e-7 absolute error at the top frequency.
struct Revolutions { float p, e; };
__device__ __forceinline__ Revolutions to_revolutions(float x) {
  const float C_HI = 0.15915494f;          // fp32(1/(2 pi))
  const float C_LO = 6.4206382e-9f;        // 1/(2 pi) - C_HI  (1/(2 pi) = 0.15915494309189535, C_HI = 0.15915493667125702)
  Revolutions r;
  r.p = x * C_HI;
  r.e = fmaf(x, C_HI, -r.p) + x * C_LO;
  return r;
}

template <int F, int NS>
__device__ __forceinline__ void posenc_b(float x, float y, float z, int h, u32x4 (&out)[NS]) {
  const Revolutions rx = to_revolutions(x), ry = to_revolutions(y), rz = to_revolutions(z);
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int a0 = 8 * s + pp, a1 = a0 + 4;          // argument for h = 0 / h = 1
      const bool trig0 = a0 < 3 * F, trig1 = a1 < 3 * F;
      float sn = 0.0f, cs = 0.0f;
      if (trig0 || trig1) {
        const int d0 = trig0 ? a0 % 3 : 0, d1 = trig1 ? a1 % 3 : 0;
        const int f0 = trig0 ? a0 / 3 : 0, f1 = trig1 ? a1 / 3 : 0;
        const Revolutions r0 = d0 == 0 ? rx : (d0 == 1 ? ry : rz);
        const Revolutions r1 = d1 == 0 ? rx : (d1 == 1 ? ry : rz);
        const float P = h ? r1.p : r0.p, E = h ? r1.e : r0.e;
        const int fe = h ? f1 : f0;
        const float t = __builtin_amdgcn_fractf(ldexpf(P, fe)) + ldexpf(E, fe);
        sn = __builtin_amdgcn_sinf(t);
        cs = __builtin_amdgcn_cosf(t);
      }
      const float e0_0 = trig0 ? sn : (a0 == 3 * F ? x : (a0 == 3 * F + 1 ? z : 0.0f));
      const float e0_1 = trig0 ? cs : (a0 == 3 * F ? y : 0.0f);
      const float e1_0 = trig1 ? sn : (a1 == 3 * F ? x : (a1 == 3 * F + 1 ? z : 0.0f));
      const float e1_1 = trig1 ? cs : (a1 == 3 * F ? y : 0.0f);
      out[s][pp] = pk_bf16(h ? e1_0 : e0_0, h ? e1_1 : e0_1);
    }
}

}  // namespace crnerf
