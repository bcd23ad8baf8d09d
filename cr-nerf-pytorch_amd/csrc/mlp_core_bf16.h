// NeRF_sigma forward for one 64-point tile per wavefront on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate), activations register-resident as packed bf16.
//
// Reference semantics: NeRF_sigma.forward, models/nerf.py:157-182, evaluated in mixed precision: the
// operands of every Linear except static_sigma (weights and input activations, incl. the two positional
// embeddings) are rounded to bf16 (RNE); products accumulate in fp32; biases, relu, softplus, sigmoid and the
// sigma head (fp32 weights on the un-rounded fp32 output of xyz_encoding_8) stay fp32.  oracle/cpu_ref.py
// `mlp_forward_bf16` restates exactly this.
//
// MI355X design:
//   * swapped-operand GEMM D[feature][point] = W[feature][k] * act[k][point].  A wave owns 64 points as two
//     32-point groups; lane (p = lane&31, h = lane>>5) holds points p and 32+p.  The 32x32 C/D layout leaves the
//     lane with features 32T + 8(r>>2) + 4h + (r&3) in accumulator register r; relu + v_cvt_pk_bf16_f32 turn
//     registers 8j..8j+7 into the 8-element B operand of k-step 2T+j of the next layer -- no cross-lane move,
//     no LDS round trip, no HBM (the weight pack permutes W's columns to match, layout.h "fragB").
//   * tile-outer / k-inner order: one output tile's two accumulators (32 VGPRs) are live at a time, its
//     epilogue overlaps the next tile's MFMAs, and the 256-feature activations of the 64 points cost
//     2 x 128 VGPRs (ping-pong) instead of 512 as fp32.
//   * every A fragment (1 KiB, one ds_read_b128 per lane) feeds TWO MFMAs (the two point groups): 4 waves x 1 KiB
//     per 64 MFMA cycles = 64 B/clk/CU, half the LDS bandwidth.  Fragments are read B_AHEAD ahead of use.
//   * weights stream L2 -> LDS through a 4-slot x 16 KiB ring with the barrier protocol of the fp32 core (mlp_core.h
//     WeightPipe): one s_barrier per 16 fragments = 32 MFMAs = 1024 MFMA cycles; all ring arithmetic is compile-time
//     (WeightPipeB below).
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

struct WeightPipeB {
  const char* base[2];   // packed streams + this wave's 4 KiB column (scalar)
  const char* cur;       // stream of the tile being multiplied
  const char* nxt;       // stream of the next tile
  uint32_t m0s[B_RING];  // scalar: LDS byte address of slot k + this wave's 4 KiB column
  uint32_t voff;         // per-lane: lane16 + STAGE_BYTES * (stage being fetched, tile-relative)
  uint32_t rd_base;      // per-lane: LDS_RING + lane16
  uint32_t lane16;
  lds_char* lds;

  // piece `i` (0..3) of tile-relative stage F (76..78 = the next tile's 0..2).  Piece 0 writes M0; pieces 1..3 reuse it
  // with their instruction offset (which applies to both sides).  Nothing else in these kernels touches M0
  // (tests/test_host.py checks the ISA).
  __device__ __forceinline__ void issue_piece(int i, int F) {   // both constant after unrolling
    const char* src = F < STAGESB_PER_PASS ? cur : nxt;
    switch (i) {   // the instruction offset must be an immediate
      case 0: glds16(m0s[F % B_RING], src, voff, 0); break;
      case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(src), "n"(1 * FRAG_BYTES) : "memory"); break;
      case 2: asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(src), "n"(2 * FRAG_BYTES) : "memory"); break;
      default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(src), "n"(3 * FRAG_BYTES) : "memory"); break;
    }
  }
  // after the 4th piece of stage F: one VALU instruction
  __device__ __forceinline__ void cursor_update(int F) {
    voff = (F + 1 == STAGESB_PER_PASS) ? lane16 : voff + STAGE_BYTES;
    asm volatile("" : "+v"(voff));   // keep it ONE register (else hipcc keeps a handful of pre-added offsets alive)
  }
  __device__ __forceinline__ void start(lds_char* lds_, const char* stream0, const char* stream1, int first_model, int lane, int wave) {
    lds = lds_;
    lane16 = (uint32_t)lane * 16u;
    rd_base = LDS_RING + lane16;
#pragma unroll
    for (int k = 0; k < B_RING; ++k) m0s[k] = (uint32_t)(uintptr_t)lds_ + LDS_RING + k * STAGE_BYTES + (uint32_t)wave * 4096u;
    base[0] = stream0 + wave * 4096;
    base[1] = stream1 + wave * 4096;
    cur = nxt = base[first_model];
    voff = lane16;
#pragma unroll
    for (int F = 0; F < B_RING - 1; ++F) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue_piece(i, F);
      cursor_update(F);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // once per tile, before its first k-step: `next_model` = the model of the tile after this one (any valid model for
  // the very last tile: its three stages are fetched and never read)
  __device__ __forceinline__ void begin_tile(int next_model) {
    cur = nxt;
    nxt = next_model ? base[1] : base[0];
    asm volatile("" : "+s"(cur), "+s"(nxt));
  }
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // fragment at padded stream position `pos` (tile-relative; 1216.. = the next tile's stage 0)
  __device__ __forceinline__ u32x4 read(int pos) const {
    return *(const __attribute__((address_space(3))) u32x4*)(lds + rd_base + ((pos / STAGE_FRAGS) % B_RING) * STAGE_BYTES +
                                                              (pos % STAGE_FRAGS) * FRAG_BYTES);
  }
  __device__ __forceinline__ void prime(u32x4 (&q)[B_AHEAD]) const {
#pragma unroll
    for (int j = 0; j < B_AHEAD; ++j) q[j] = read(j);
  }
};

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

struct EpiTemps {
  float t[4][2];
  __device__ __forceinline__ void fetch(int slot, int qc, const f32x16& acc) {
    float v0 = acc[2 * (qc >> 1)], v1 = acc[2 * (qc >> 1) + 1];
    asm volatile("" : "+v"(v0), "+v"(v1));
    t[slot][0] = v0;
    t[slot][1] = v1;
  }
};

struct NoEpi {
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void load(int, int, int, const f32x16&) {}
  __device__ __forceinline__ void finish(int, int, int) {}
};

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

// hidden layers: accumulator registers 2hc, 2hc+1 become dword hc&3 of k-step 2T + (hc>>2) of the next layer's B operand.
// Both halves are real asm: an empty `asm volatile("" : "+v"(x))` pin right after a VALU instruction costs an s_nop
// (hipcc's inline-asm hazard rule), i.e. two more issue slots per quarter in a gap that has none to spare.
template <bool RELU>
struct PackEpi : EpiTemps {
  u32x4 (&dst)[KS_HID][2];
  __device__ __forceinline__ explicit PackEpi(u32x4 (&d)[KS_HID][2]) : dst(d) {}
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void load(int slot, int, int qc, const f32x16& acc) {
    asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3"
                 : "=v"(t[slot][0]), "=v"(t[slot][1])
                 : "a"(acc[2 * (qc >> 1)]), "a"(acc[2 * (qc >> 1) + 1]));
  }
  __device__ __forceinline__ void finish(int slot, int T, int qc) {
    const int hc = qc >> 1, g = qc & 1;
    uint32_t pk;
    if (RELU)
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(pk) : "v"(t[slot][0]), "v"(t[slot][1]));
    else
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(t[slot][0]), "v"(t[slot][1]));
    dst[2 * T + (hc >> 2)][g][hc & 3] = pk;
  }
};

// xyz_encoding_8: as PackEpi<true>, plus static_sigma (256 -> 1) in fp32 on the un-rounded activations
struct SigmaEpi : EpiTemps {
  u32x4 (&dst)[KS_HID][2];
  float (&sg)[2];
  const lds_float* wsig;
  int h;
  f32x4 wv[4];
  __device__ __forceinline__ SigmaEpi(u32x4 (&d)[KS_HID][2], float (&s)[2], const lds_float* w, int h_) : dst(d), sg(s), wsig(w), h(h_) {}
  __device__ __forceinline__ void prefetch(int T) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wv[c] = lds_f4(wsig + 32 * T + 8 * c + 4 * h);
  }
  __device__ __forceinline__ void load(int slot, int, int qc, const f32x16& acc) { fetch(slot, qc, acc); }
  __device__ __forceinline__ void finish(int slot, int T, int qc) {
    const int hc = qc >> 1, g = qc & 1, e = 2 * (hc & 1);
    const float x0 = t[slot][0], x1 = t[slot][1];
    dst[2 * T + (hc >> 2)][g][hc & 3] = pack_pair<true>(x0, x1);
    sg[g] = fmaf(wv[hc >> 1][e], fmaxf(x0, 0.0f), sg[g]);
    sg[g] = fmaf(wv[hc >> 1][e + 1], fmaxf(x1, 0.0f), sg[g]);
    asm volatile("" : "+v"(sg[g]));
  }
};

// 1 / (1 + 2^(-x log2 e)) on the hardware exp2 / rcp (1 ulp each): 4 instructions instead of ~22 for expf + IEEE
// division, 64 of them per lane per tile; the relative error (~2e-7) is far below this path's bf16 noise
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }

// nn.Softplus(beta=1, threshold=20) on the hardware exp2 / log2: ln(1 + e^x) = log2(1 + 2^(x log2 e)) ln 2.  Absolute error
// ~1e-7 (relative ~1e-3 only where sigma < 1e-4, i.e. where the sample is transparent anyway).
__device__ __forceinline__ float softplus_fast(float x) {
  return x > 20.0f ? x : __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x * 1.44269504f)) * 0.69314718f;
}

struct RgbEpi : EpiTemps {   // static_rgb: sigmoid, fp32 out
  f32x16 (&feat)[2][2];
  __device__ __forceinline__ explicit RgbEpi(f32x16 (&f)[2][2]) : feat(f) {}
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void load(int slot, int, int qc, const f32x16& acc) { fetch(slot, qc, acc); }
  __device__ __forceinline__ void finish(int slot, int T, int qc) {
    const int hc = qc >> 1, g = qc & 1;
    feat[g][T][2 * hc] = sigmoid_fast(t[slot][0]);
    feat[g][T][2 * hc + 1] = sigmoid_fast(t[slot][1]);
  }
};

// acc-layout bias of output tile T: element 4c + j = bias[32T + 8c + 4h + j]
__device__ __forceinline__ void load_bias_half(f32x16& bv, const lds_float* bias, int T, int h, int half) {
#pragma unroll
  for (int c = 2 * half; c < 2 * half + 2; ++c) {
    const f32x4 b = lds_f4(bias + 32 * T + 8 * c + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[4 * c + j] = b[j];
  }
}

// One layer.  NT output tiles; per tile NSA k-steps with B operands srcA[s][g] then NSB from srcB (g = point group).
// FBASE: pass-relative index of the layer's first fragment; G0: index of its first tile in the pass (tile G
// accumulates in accs[G & 1]); PT: the previous layer's last tile, whose epilogue `prev` is still pending and runs
// behind this layer's first tile -- legal because k-step s of any layer reads source tile s/2, so the last source
// tile is only needed by the last two k-steps.  On return this layer's tile NT-1 is pending in the same way.
// biasv: on entry the bias of tile 0 (acc layout); reloaded for each following tile -- and from next_bias for the next
// layer's tile 0 -- a few k-steps before it is needed.  (next_bias must be a readable address even for the last layer:
// the consts block sits at LDS address 0, so a null sentinel would alias the first layer's bias.)
template <int NT, int NSA, int NSB, int FBASE, int G0, int PT, int NA, int NB, class PREV, class EPI>
__device__ __forceinline__ void mma_layer_b(WeightPipeB& p, const u32x4 (&srcA)[NA][2], const u32x4 (&srcB)[NB][2],
                                            u32x4 (&q)[B_AHEAD], f32x16 (&accs)[2][2], f32x16& biasv, const lds_float* bias,
                                            const lds_float* next_bias, int h, PREV& prev, EPI& epi) {
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
      // quarters of the previous tile's epilogue carried by this k-step.  Long layers: one per k-step from k-step 1,
      // a second one every fourth k-step, done after k-step 13 (k-step 14 of a layer's first tile reads the result);
      // the short layers (K = 96, 128) take four per k-step in k-steps 1..4.  (Spreading them over k-steps 0..15 with
      // a single double in a layer's first tile measured the same cycles and cost 17 more spilled registers.)
      const int first = LONG ? (s - 1) + (s - 1) / 4 : 4 * (s - 1);
      const int count = s < 1 ? 0 : (LONG ? (s <= 13 ? (s % 4 == 0 ? 2 : 1) : 0) : (s <= 4 ? 4 : 0));
      const f32x16& pa0 = accs[cur ^ 1][0];
      const f32x16& pa1 = accs[cur ^ 1][1];

      u32x4 af = q[i % B_AHEAD];
      asm volatile("" : "+v"(af));   // ties this k-step's MFMAs into the side-effect chain (see the epilogue notes)
      const bf16x8 a = __builtin_bit_cast(bf16x8, af);
      const u32x4 b0 = s < NSA ? srcA[s < NSA ? s : 0][0] : srcB[s < NSA ? 0 : s - NSA][0];
      const u32x4 b1 = s < NSA ? srcA[s < NSA ? s : 0][1] : srcB[s < NSA ? 0 : s - NSA][1];
      // ---- gap 1
      accs[cur][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b0), s == 0 ? biasv : accs[cur][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // MFMA first, its fillers behind it (hipcc would hoist the fillers above the MFMA)
#pragma unroll
      for (int u = 0; u < count; ++u) {
        const int qc = first + u;
        if (T == 0) prev.load(u, PT, qc, (qc & 1) ? pa1 : pa0);
        else epi.load(u, T - 1, qc, (qc & 1) ? pa1 : pa0);
      }
      if (b_piece_at(i) >= 0) p.issue_piece(b_piece_at(i), i / STAGE_FRAGS + B_RING - 1);
      __builtin_amdgcn_sched_barrier(0);   // pin the hand-made pipeline (hipcc would bunch the fillers)
      // ---- gap 2
      accs[cur][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b1), s == 0 ? biasv : accs[cur][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the LDS read goes FIRST behind the MFMA (tools/ubench: 32.5 cycles/MFMA; with the VALU pair first and the read last
      // the s_nop hipcc puts before the next k-step's MFMA becomes visible: 34.5)
      q[i % B_AHEAD] = p.read(b_pos(i + B_AHEAD));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < count; ++u) {
        const int qc = first + u;
        if (T == 0) prev.finish(u, PT, qc);
        else epi.finish(u, T - 1, qc);
      }
      if (s == 0) {
        if (T == 0) prev.prefetch(PT);
        else epi.prefetch(T - 1);
      }
      if (s == NS - 4 || s == NS - 3) {   // the next tile's bias, half per k-step (biasv was consumed at s == 0); early enough that
                                          // the wait for it leaves the two newest fragment reads in flight
        const lds_float* nb = (T + 1 < NT) ? bias : next_bias;
        load_bias_half(biasv, nb, (T + 1 < NT) ? T + 1 : 0, h, s - (NS - 4));
      }
      if (b_advance_at(i)) p.advance();
      if (b_cursor_at(i)) p.cursor_update(i / STAGE_FRAGS + B_RING - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// One 64-point tile through one model.  pe[s][g] / dv[s][g]: the embeddings as B operands (posenc_b below);
// q carries the look-ahead fragments between layers, tiles and passes.  Returns feat[g][t][r] = rgb feature
// 32t + 8(r>>2) + 4h + (r&3) of point 32g + p, and sigma[g] (valid in both lane halves).
__device__ __forceinline__ void mlp_tile_b(WeightPipeB& p, int model, int next_model, const u32x4 (&pe)[KS_XYZ][2], const u32x4 (&dv)[KS_DIR][2],
                                           f32x16 (&feat)[2][2], float (&sigma)[2], int h, u32x4 (&q)[B_AHEAD], PhaseTimer& tm) {
  // the consts block is loop-invariant LDS: launder its address once per tile, or LICM hoists all ~1,300 bias /
  // sigma-weight reads out of the caller's tile loop and spills them to scratch
  p.begin_tile(next_model);
  uint32_t c_off = model ? LDS_CONST1 : LDS_CONST0;
  asm volatile("" : "+s"(c_off));
  const lds_float* C = (const lds_float*)(p.lds + c_off);
  const lds_float* B1 = C + C_BIAS;
  u32x4 actA[KS_HID][2], actB[KS_HID][2];
  f32x16 accs[2][2], biasv;
  float sg[2] = {0.0f, 0.0f};
  load_bias_half(biasv, B1, 0, h, 0);
  load_bias_half(biasv, B1, 0, h, 1);
  tm.tick(T_PROLOGUE);

  // tile parity: every layer before dir has 8 tiles, so accs[G & 1] with G0 = 8 * layer; dir starts at 72, rgb at 76
  NoEpi none;
  PackEpi<true> eA(actA), eB(actB);
  PackEpi<false> efin(actA);
  SigmaEpi e8(actB, sg, C + C_WSIG, h);
  RgbEpi ergb(feat);
  mma_layer_b<8, KS_XYZ, 0, OFFB_L1, 0, 0>(p, pe, pe, q, accs, biasv, B1, B1 + 1 * W_HIDDEN, h, none, eA);                           // xyz_encoding_1
  tm.tick(T_X0);
  mma_layer_b<8, KS_HID, 0, OFFB_L2, 8, 7>(p, actA, actA, q, accs, biasv, B1 + 1 * W_HIDDEN, B1 + 2 * W_HIDDEN, h, eA, eB);          // 2
  tm.tick(T_X1);
  mma_layer_b<8, KS_HID, 0, OFFB_L2 + FB_HID, 16, 7>(p, actB, actB, q, accs, biasv, B1 + 2 * W_HIDDEN, B1 + 3 * W_HIDDEN, h, eB, eA);      // 3
  mma_layer_b<8, KS_HID, 0, OFFB_L2 + 2 * FB_HID, 24, 7>(p, actA, actA, q, accs, biasv, B1 + 3 * W_HIDDEN, B1 + 4 * W_HIDDEN, h, eA, eB);  // 4
  tm.tick(T_X2);
  mma_layer_b<8, KS_XYZ, KS_HID, OFFB_L5, 32, 7>(p, pe, actB, q, accs, biasv, B1 + 4 * W_HIDDEN, B1 + 5 * W_HIDDEN, h, eB, eA);      // 5 = Linear(cat[xyz, h])
  tm.tick(T_X3);
  mma_layer_b<8, KS_HID, 0, OFFB_L6, 40, 7>(p, actA, actA, q, accs, biasv, B1 + 5 * W_HIDDEN, B1 + 6 * W_HIDDEN, h, eA, eB);         // 6
  mma_layer_b<8, KS_HID, 0, OFFB_L6 + FB_HID, 48, 7>(p, actB, actB, q, accs, biasv, B1 + 6 * W_HIDDEN, B1 + 7 * W_HIDDEN, h, eB, eA);      // 7
  mma_layer_b<8, KS_HID, 0, OFFB_L6 + 2 * FB_HID, 56, 7>(p, actA, actA, q, accs, biasv, B1 + 7 * W_HIDDEN, C + C_BFIN, h, eA, e8);   // 8 (+ static_sigma)
  mma_layer_b<8, KS_HID, 0, OFFB_FIN, 64, 7>(p, actB, actB, q, accs, biasv, C + C_BFIN, C + C_BDIR, h, e8, efin);                    // xyz_encoding_final (no activation)
  tm.tick(T_MMA);
  sg[0] += __shfl_xor(sg[0], 32);
  sg[1] += __shfl_xor(sg[1], 32);
  sigma[0] = softplus_fast(sg[0] + C[C_BSIG]);
  sigma[1] = softplus_fast(sg[1] + C[C_BSIG]);
  tm.tick(T_SIGMA);
  mma_layer_b<4, KS_HID, KS_DIR, OFFB_DIR, 72, 7>(p, actA, dv, q, accs, biasv, C + C_BDIR, C + C_BRGB, h, efin, eB);                 // dir_encoding = relu(Linear(cat[final, dir]))
  tm.tick(T_X4);
  mma_layer_b<2, KS_HALF, 0, OFFB_RGB, 76, 3>(p, actB, actB, q, accs, biasv, C + C_BRGB, C + C_BRGB, h, eB, ergb);                      // static_rgb = sigmoid(Linear)
  tm.tick(T_MMA);
#pragma unroll
  for (int qc = 0; qc < 16; ++qc) {   // rgb's last tile: nothing left to hide it behind
    ergb.load(0, 1, qc, accs[(76 + 1) & 1][qc & 1]);
    ergb.finish(0, 1, qc);
  }
  tm.tick(T_EPILOGUE);
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
