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
// Weights: the 16 KiB-stage LDS ring of mlp_core.h, X_RING = 7 slots here (dynamic slot counters; LDS-DMA as inline asm, see glds16), four pieces per wave and
// stage; fragments are consumed in stream order through an X_AHEAD-deep register queue, so the stage boundaries (16 fragments) need not
// line up with the k-steps (3 NT fragments).
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"
#include "mlp_core.h"

// CRNERF_X_NP: pieces per operand.  3 (default) = the bf16 x3 core described above.  2 = the "h2" core (mlp_forward_h2.hip, render_fused_h2.hip):
// TWO fp16 pieces per operand, x = h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) (11 + 11 mantissa bits and the sign of h2 -- again one fp32
// rounding, as long as h2 is a normal fp16 number; fp16 subnormals are honoured by the matrix cores, so below that the ABSOLUTE error is <= 2^-25),
// and a product is the THREE leading piece products w2 a1 + w1 a2 + w1 a1 (the dropped w2 a2 is <= 2^-24 of it): half the MFMAs of the x3 core
// and two thirds of its weight stream (layout.h "fragH": weights AND biases scaled by 2^8 at pack time so that the weights' second pieces stay
// normal; the layer output is scaled down in registers -- exact).  Range: |activation| < 65,504 and |weight| < 255, else inf / nan come out.
// Everything else -- ring, queue, layer walk -- is the same code; the variant lives in its own inline namespace.
#ifndef CRNERF_X_NP
#define CRNERF_X_NP 3
#endif

namespace crnerf {
#if CRNERF_X_NP == 2
inline namespace xcore_h2 {
#else
inline namespace xcore_x3 {
#endif
constexpr int XNP = CRNERF_X_NP;
static_assert(XNP == 2 || XNP == 3, "x core: two fp16 pieces or three bf16 pieces");
constexpr int XSTREAM_FRAGS = XNP == 2 ? STREAMH_FRAGS : STREAMX_FRAGS;       // forward stream of a model
constexpr int XPAD_DIR = XNP == 2 ? 0 : FX_DIR - FX_DIR_USED;                 // stage / queue padding behind dir_encoding
constexpr float XWSCALE = XNP == 2 ? H2_WSCALE : 1.0f;                        // the packed weights carry this factor

#if CRNERF_X_NP == 2
typedef _Float16 xbf16x8 __attribute__((ext_vector_type(8)));    // (the operand vector of the variant's MFMA)
#else
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
#endif
typedef uint32_t xu32x4 __attribute__((ext_vector_type(4)));

// The x3 ring: mlp_core.h's protocol with SEVEN 16-KiB slots.  The 3.75 MB stream of a model does not stay in an XCD's 4 MB L2, so a piece's
// latency is the memory side's, not L2's, and what hides it is the number of stages in flight (the protocol keeps four out of flight: c - 1
// being refilled, c and c + 1 readable, c + 2 certified).  Measured per 1,024 rays: 6 x 16 KiB 1.38 ms, 7 x 16 KiB 1.34 ms, 14 x 8 KiB
// (-DCRNERF_X_RING=14 -DCRNERF_X_STAGE_FRAGS=8: ten stages in flight in the same LDS, twice the barriers) 1.36 ms; without the DMA at all 1.12 ms.
#ifndef CRNERF_X_RING
#define CRNERF_X_RING 7
#endif
#ifndef CRNERF_X_STAGE_FRAGS
#define CRNERF_X_STAGE_FRAGS 16
#endif
constexpr int X_RING = CRNERF_X_RING;
constexpr int X_STAGE_FRAGS = CRNERF_X_STAGE_FRAGS;
constexpr int X_STAGE_BYTES = X_STAGE_FRAGS * FRAG_BYTES;
constexpr int X_PIECES = X_STAGE_FRAGS / 4;        // 1 KiB pieces per wave and stage (four waves)
static_assert(STAGE_FRAGS % X_STAGE_FRAGS == 0 && X_PIECES >= 1 && X_PIECES <= 4, "x3 stages divide the stream's 16-fragment alignment");
constexpr int LDS_TOUCH_X = LDS_RING + X_RING * X_STAGE_BYTES;   // 4 x 256 B: where the L2-prefetch touches land (never read)
constexpr int LDS_SCRATCH_X = LDS_TOUCH_X + 1024;
static_assert(X_RING >= 5 && (X_PIECES + 1) * (X_RING - 3) + 8 <= 63 && LDS_SCRATCH_X + 4 * 5120 <= 160 * 1024, "x3 ring: protocol depth and LDS budget (ring + four waves of ray scratch)");
#ifndef CRNERF_X_TOUCH
#define CRNERF_X_TOUCH 0
#endif
constexpr int X_TOUCH = CRNERF_X_TOUCH;   // stages between an L2-prefetch touch of a stage and its LDS-DMA; 0 = no touches (the default: they did not pay)
#ifndef CRNERF_X_AHEAD_PAIRS
#define CRNERF_X_AHEAD_PAIRS 1
#endif
// fragments read ahead of the one being multiplied: the piece groups of one tile PAIR (12 MFMAs = 384 matrix-pipe cycles on the x3 core, 6 = 192 on
// the h2 core).  Two pairs on the h2 core (-DCRNERF_X_AHEAD_PAIRS=2) measured the same -- 0.824 vs 0.82 ms per 1,024 rays -- at twice the spills:
// the ds_read_b128 latency is not what that core waits for
constexpr int X_AHEAD = 2 * XNP * CRNERF_X_AHEAD_PAIRS;
static_assert(X_AHEAD % XNP == 0 && X_AHEAD <= X_STAGE_FRAGS, "the queue holds whole piece groups and never reaches past the next stage");

// WeightPipe of mlp_core.h for the x3 stream: STAGESX_PER_PASS stages per pass, LDS-DMA as asm.  Protocol as there: stages c and c + 1 may be
// read; advance() -- after the last read of stage c has been issued -- waits until this wave's pieces of stage c + 2 have landed (counted
// vmcnt) and barriers; the slot of stage c - 1 is then refilled with stage c + X_RING - 1 by four issue_piece() calls during stage c + 1.
struct WeightPipeX {
  lds_char* lds;
  const char* base[2];   // scalar: packed streams + this wave's 4 KiB column
  const char* pf_ptr;
  int pf_left, pf_pass, passes0, passes;
  int stages_per_pass = XSTREAM_FRAGS / X_STAGE_FRAGS;   // stages per pass: forward stream (set_stream_frags() for another)
  uint32_t pf_slot, rd_slot, rd_addr, lane16, lds_ring;
  __device__ __forceinline__ void set_stream_frags(int frags) { stages_per_pass = frags / X_STAGE_FRAGS; }   // before start()
  // L2 prefetch (experiment, off): a model's 3.75 MB stream is the size of an XCD's L2 and does not stay there, so whichever CU of the XCD reaches a stage first
  // pays the memory side's latency inside its LDS-DMA -- 19 % of the kernel (with the DMA removed: 1.12 vs 1.38 ms per 1,024 rays).  Each wave
  // therefore TOUCHES the stage X_TOUCH stages further down the stream once per stage: one global_load_lds_dword whose 64 lanes hit 64 cache
  // lines (8 KiB; even waves the first half of the stage, odd waves the second) and whose data lands in a dummy LDS row nobody reads -- an
  // LDS-DMA because a load into a VGPR would write that register at an unknown later time.  Touches are unconditional, one per stage and wave,
  // so the vmcnt bookkeeping of start() / advance() counts them exactly.  MEASURED (-DCRNERF_X_TOUCH=6): SLOWER, 1.34 -> 1.45 ms per 1,024 rays and
  // 5.21 -> 5.77 ms per 2^20 points -- VMEM operations retire in order, so a touch that misses to memory holds back the count of every piece
  // issued behind it.  Off by default; kept for the record.
  const char* tbase[2];  // scalar: packed streams + this wave's half stage
  const char* tp_ptr;
  int tp_left, tp_pass;
  uint32_t touch_lds;
  __device__ __forceinline__ void touch() {
    if (X_TOUCH > 0) {
#ifndef CRNERF_EXP_NOGLDS
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:0" ::"s"(touch_lds), "v"(lane16 << 3), "s"(tp_ptr) : "memory");
#endif
      tp_ptr += X_STAGE_BYTES;
      if (--tp_left == 0) {
        tp_left = stages_per_pass;
        tp_pass = (tp_pass + 1 == passes) ? 0 : tp_pass + 1;
        tp_ptr = (tp_pass < passes0) ? tbase[0] : tbase[1];
      }
    }
  }

  // piece i (0 .. X_PIECES - 1) of the stage being fetched: this wave's fragments X_PIECES * wave + i
  __device__ __forceinline__ void issue_piece(int i) {
    const uint32_t dst = lds_ring + pf_slot * X_STAGE_BYTES;
    if (i == 0) touch();
#ifndef CRNERF_EXP_NOGLDS   // (timing experiments only: results are garbage without the loads)
    switch (i) {   // the instruction offset must be an immediate
      case 0: glds16(dst, pf_ptr, lane16, 0); break;
      case 1: glds16(dst, pf_ptr, lane16, FRAG_BYTES); break;
      case 2: glds16(dst, pf_ptr, lane16, 2 * FRAG_BYTES); break;
      default: glds16(dst, pf_ptr, lane16, 3 * FRAG_BYTES); break;
    }
#endif
    if (i == X_PIECES - 1) {
      pf_slot = (pf_slot + 1 == X_RING) ? 0u : pf_slot + 1;
      pf_ptr += X_STAGE_BYTES;
      if (--pf_left == 0) {
        pf_left = stages_per_pass;
        pf_pass = (pf_pass + 1 == passes) ? 0 : pf_pass + 1;
        pf_ptr = (pf_pass < passes0) ? base[0] : base[1];
      }
    }
  }
  // Call once, all waves.  On return stages 0 and 1 are readable.
  __device__ __forceinline__ void start(lds_char* lds_, const char* stream0, const char* stream1, int passes0_, int passes_, int lane, int wave) {
    lds = lds_;
    lane16 = (uint32_t)lane * 16u;
    const uint32_t wave4k = (uint32_t)wave * (X_PIECES * FRAG_BYTES);   // this wave's column of a stage
    lds_ring = (uint32_t)(uintptr_t)lds_ + LDS_RING + wave4k;
    base[0] = stream0 + wave4k;
    base[1] = stream1 + wave4k;
    passes0 = passes0_;
    passes = passes_;
    pf_pass = 0;
    pf_left = stages_per_pass;
    pf_ptr = (passes0 > 0) ? base[0] : base[1];
    pf_slot = 0;
    rd_slot = 0;
    rd_addr = LDS_RING + lane16;
    tbase[0] = stream0;   // (touch experiment: one 8 KiB stage per instruction)
    tbase[1] = stream1;
    touch_lds = (uint32_t)(uintptr_t)lds_ + LDS_TOUCH_X + (uint32_t)wave * 256u;
    tp_pass = 0;
    tp_left = stages_per_pass - X_TOUCH;            // the touch cursor starts X_TOUCH stages into the first pass (a pass is >= 200 stages)
    tp_ptr = ((passes0 > 0) ? tbase[0] : tbase[1]) + X_TOUCH * X_STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < X_RING - 1; ++s)
#pragma unroll
      for (int i = 0; i < X_PIECES; ++i) issue_piece(i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((X_PIECES + (X_TOUCH > 0 ? 1 : 0)) * (X_RING - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  __device__ __forceinline__ uint32_t next_addr() const {
    const uint32_t n = (rd_slot + 1 == X_RING) ? 0u : rd_slot + 1;
    return LDS_RING + n * X_STAGE_BYTES + lane16;
  }
  // stores: store instructions this wave has issued since its pieces of stage c + 2 (training twin; they share vmcnt with the LDS-DMA and retire in
  // order, so they may stay in flight on top of the two stages of pieces) -- a compile-time lower bound, see mma_layer_x3
  __device__ __forceinline__ void advance(int stores = 0) {
    constexpr int PER_STAGE = X_PIECES + (X_TOUCH > 0 ? 1 : 0);   // VMEM operations of the ring per stage and wave: the pieces (+ one touch, issued before piece 0)
    switch (stores) {
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (X_RING - 4) + 2) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (X_RING - 4) + 4) : "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (X_RING - 4) + 6) : "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (X_RING - 4) + 8) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (X_RING - 4)) : "memory"); break;
    }
#ifndef CRNERF_EXP_NOBARRIER   // (timing experiments only: racy without it) what the per-stage rendezvous of the four waves costs
    __builtin_amdgcn_s_barrier();
#endif
    rd_slot = (rd_slot + 1 == X_RING) ? 0u : rd_slot + 1;
    rd_addr = LDS_RING + rd_slot * X_STAGE_BYTES + lane16;
  }
  // fragment at slot s of the stage being consumed; s >= 16 reads ahead into the next stage
  __device__ __forceinline__ xu32x4 read_slot(int s) const {
    const uint32_t a = (s < X_STAGE_FRAGS) ? rd_addr + s * FRAG_BYTES : next_addr() + (s - X_STAGE_FRAGS) * FRAG_BYTES;
    return *(const __attribute__((address_space(3))) xu32x4*)(lds + a);
  }
  __device__ __forceinline__ void prime(xu32x4 (&q)[X_AHEAD]) const {
#pragma unroll
    for (int i = 0; i < X_AHEAD; ++i) q[i] = read_slot(i);
  }
};

// eight fp32 k-values of a lane -> the three piece operands (dword d = values 2d, 2d + 1; v_cvt_pk_bf16_f32 rounds to nearest even)
__device__ __forceinline__ uint32_t x3_pk(float a, float b) {
  typedef __bf16 pk2 __attribute__((ext_vector_type(2)));
  const pk2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void x3_split(const float (&v)[8], xbf16x8& b1, xbf16x8& b2, xbf16x8& b3) {
#if CRNERF_X_NP == 2
  // two fp16 pieces (v_cvt_f16_f32 rounds to nearest even and keeps subnormals); b3 is not used by this variant
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h1 = (_Float16)v[e];
    b1[e] = h1;
    b2[e] = (_Float16)(v[e] - (float)h1);
  }
  b3 = b2;
  return;
#else
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
#endif
}

// ---- training twin (crnerf_render_rays_train_f32x3): the saved state of the fp32 training twins (mlp_train16.h: acts[10][P][256] fp32 in reference
// feature order, then the relu-activity bits masks[10][P][4] x 64 bit, bit 4T + r <-> feature 16T + 4g + r), written from this core's registers so
// that the fp32 backward twins (mlp_backward16_kernel, the weight-gradient kernels) read it unchanged.  A layer's output is stored when the NEXT
// layer walks it as its B operand -- the eight values of k-step s are two 16-byte pieces (64 contiguous bytes per point with the other lane half),
// two stores per k-step, spread evenly through that layer's MFMAs; the activity bits are formed right after the relu.
constexpr uint32_t SAVEX_OOB = 0xF0000000u;   // offset of a lane that stores nothing (beyond every buffer resource: the hardware drops the store)
struct SaveRowX { __amdgpu_buffer_rsrc_t rs; };   // the rows of one slot
struct NoSaveX {
  static constexpr bool on = false;
  __device__ __forceinline__ SaveRowX row(int) const { return SaveRowX{}; }
  __device__ __forceinline__ uint32_t offset() const { return 0; }
  template <int NT>
  __device__ __forceinline__ void masks(int, const f32x16 (&)[NT]) const {}
};
struct ActSaveX {
  static constexpr bool on = true;
  float* base; long P; long n; bool valid; int h;
  // row stores are UNCONDITIONAL raw-buffer stores (lanes without a point carry an out-of-range offset), so that their number between two
  // points of a layer's code is a compile-time constant the ring's vmcnt can allow for (mma_layer_x3).  (Measured: the allowance changes
  // nothing -- 11.70 vs 11.75 ms per 16,384-ray chunk; the twin's time is the inference kernel's plus the time of its 10.5 KB per point of
  // stores, 5.4 + 2.4 ms per 2^20 points, as for the bf16 twin)
  __device__ __forceinline__ SaveRowX row(int slot) const {
    return SaveRowX{__builtin_amdgcn_make_buffer_rsrc(base + (long)slot * P * 256, 0, (int)(uint32_t)(P * 1024), 0x00020000)};
  }
  __device__ __forceinline__ uint32_t offset() const {
#ifdef CRNERF_EXP_X3_NOSAVE   // (timing experiments only) every row store issued and dropped
    return SAVEX_OOB;
#endif
    return valid ? (uint32_t)n * 1024u + 16u * (uint32_t)h : SAVEX_OOB;
  }
  // lane (p, h) owns the mask words g = h and g = h + 2: bit 4T + r <-> register 4q + r of tile T >> 1, q = 2 (T & 1) + (g >> 1)
  template <int NT>
  __device__ __forceinline__ void masks(int slot, const f32x16 (&a)[NT]) const {
    unsigned long long* m = (unsigned long long*)(base + (size_t)10 * P * 256) + ((size_t)slot * P + n) * 4;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int T = 0; T < 2 * NT; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t bit = a[T >> 1][4 * (2 * (T & 1) + gg) + r] > 0.0f ? 1u : 0u;   // post-relu values: > 0 <=> pre-activation > 0
          const int k = 4 * T + r;
          if (k < 32) lo |= bit << k; else hi |= bit << (k - 32);
        }
      if (valid) m[h + 2 * gg] = ((unsigned long long)hi << 32) | lo;
    }
  }
};

#undef CRNERF_MFMA_X
#if CRNERF_X_NP == 2
#define CRNERF_MFMA_X(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(xbf16x8, (a)), (b), (c), 0, 0, 0)
#else
#define CRNERF_MFMA_X(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8, (a)), (b), (c), 0, 0, 0)
#endif

// Scheduling fence between a tile pair's queue refills (the ds_read_b128 of the fragments X_AHEAD further down the stream) and its MFMAs: VALU and
// SALU instructions may cross it, LDS reads and MFMAs may not.  Without it hipcc sinks every look-ahead read to just in front of the MFMA that
// consumes it (one live fragment register instead of the queue: ds_read, s_waitcnt lgkmcnt(0), v_mfma -- the LDS latency exposed once per MFMA).
// MEASURED on the h2 core (-DCRNERF_X_FENCE_ON=1): the ISA then shows the intended pattern (four reads, six MFMAs, lgkmcnt(6) / (4)) and the time
// does not move -- 0.816 vs 0.82 ms per 1,024 rays, 3.43 vs 3.39 ms per 2^20 points: what the h2 core waits for is the layer boundary (the
// accumulators' way out of the AGPRs: v_accvgpr_read, scale, relu, range max -- ~0.4 k VALU instructions per layer with no MFMA in flight) and
// the weight DMA (11-13 %, -DCRNERF_EXP_NOGLDS), not the LDS latency.  Off by default.
#ifndef CRNERF_X_FENCE_ON
#define CRNERF_X_FENCE_ON 0
#endif
#if CRNERF_X_FENCE_ON
#define CRNERF_X_FENCE() __builtin_amdgcn_sched_barrier(0x6)
#else
#define CRNERF_X_FENCE() ((void)0)
#endif

// One layer: NT output tiles; k-steps 0..NSA-1 take their B operands from srcA (registers 8(s%2).. of tile s/2), the following NSB from
// srcB.  FOFF: the layer's first fragment modulo the stage (0: every layer is whole stages); PAD: stage-padding fragments behind the
// layer (dir_encoding), skipped through the queue without being multiplied.  q always holds the next X_AHEAD fragments of the STREAM.
// SAVEA / SAVEB (training twin): source A / B is saved while it is walked -- rowA / rowB = the slot's rows, voff = this lane's byte offset.
// FINA / FINB (h2 core, "lazy epilogue"): the source holds the RAW accumulator values of the layer that produced it (scaled by XWSCALE, before
// the activation) and is finished here, eight values at a time, when its k-step comes up -- 1: scale down + relu, 2: scale down only (a linear
// layer's output) -- with the running max |activation| of the range guard kept in *amax.  Every activation is a B operand exactly once per layer,
// so this is the same work as an epilogue between the layers, moved from a stretch with no MFMA in flight into the shadow of this layer's MFMAs.
template <int NT, int NSA, int NSB, int PAD, bool SAVEA = false, bool SAVEB = false, int FINA = 0, int FINB = 0, int NA, int NB>
__device__ __forceinline__ void mma_layer_x3(WeightPipeX& p, const f32x16 (&srcA)[NA], const f32x16 (&srcB)[NB], f32x16 (&acc)[NT],
                                             xu32x4 (&q)[X_AHEAD], SaveRowX rowA = SaveRowX{}, SaveRowX rowB = SaveRowX{}, uint32_t voff = 0,
                                             float* amax = nullptr) {
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
    if (slot == X_STAGE_FRAGS - 1) {
      // row stores issued since this wave's pieces of the stage the barrier certifies (stage c + 2, whose last piece went out in the take() of
      // fragment X_STAGE_FRAGS (c - X_RING + 4) + last piece slot): the k-steps of THIS layer that start behind it, up to f -- a lower bound
      // (the previous layer's are ignored)
      constexpr int LASTP = (X_PIECES - 1) * (X_STAGE_FRAGS / X_PIECES);
      int st = 0;
      for (int ks = 0; ks < NSA + NSB; ++ks)
        if ((ks < NSA ? SAVEA : SAVEB) && ks * NT * XNP > f - (X_STAGE_FRAGS * (X_RING - 3) - 1 - LASTP) && ks * NT * XNP <= f) st += 2;
      p.advance(st);
    }
    return w;
  };
  // the split of k-step s + 1 is written BEFORE the MFMAs of k-step s (it does not depend on them): its ~50 VALU instructions and the two
  // row stores of the training twin can sit in the shadow of 6 NT MFMAs instead of in front of them.  (Measured against the split in front,
  // -DCRNERF_X3_NOPIPE: 1.379 vs 1.384 ms per 1,024 rays -- hipcc's scheduler had already found it; the kernel runs at the bf16 pipe's
  // power-limited rate)
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
    const int fin = s < NSA ? FINA : FINB;
    if (fin) {
      float m = *amax;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        v[e] *= 1.0f / XWSCALE;
        v[e + 1] *= 1.0f / XWSCALE;
        if (fin == 1) { v[e] = fmaxf(v[e], 0.0f); v[e + 1] = fmaxf(v[e + 1], 0.0f); }
        m = fmaxf(fmaxf(m, fabsf(v[e])), fabsf(v[e + 1]));
      }
      *amax = m;
    }
    x3_split(v, b1, b2, b3);
  };
  xbf16x8 b1, b2, b3;
  prepare(0, b1, b2, b3);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    xbf16x8 n1 = b1, n2 = b2, n3 = b3;
#ifndef CRNERF_X3_NOPIPE
    if (s + 1 < NS) prepare(s + 1, n1, n2, n3);
#endif
#pragma unroll
    for (int T = 0; T < NT; T += 2) {   // two tiles at a time: consecutive MFMAs belong to different accumulators
#if CRNERF_X_NP == 2
      const int f = (s * NT + T) * 2;
      const xu32x4 u1 = take(f), u2 = take(f + 1);
      const xu32x4 w1 = take(f + 2), w2 = take(f + 3);
      CRNERF_X_FENCE();
      acc[T] = CRNERF_MFMA_X(u2, b1, acc[T]);          // small terms first
      acc[T + 1] = CRNERF_MFMA_X(w2, b1, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u1, b2, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w1, b2, acc[T + 1]);
      acc[T] = CRNERF_MFMA_X(u1, b1, acc[T]);
      acc[T + 1] = CRNERF_MFMA_X(w1, b1, acc[T + 1]);
      CRNERF_X_FENCE();
#else
      const int f = (s * NT + T) * 3;
      const xu32x4 u1 = take(f), u2 = take(f + 1), u3 = take(f + 2);
      const xu32x4 w1 = take(f + 3), w2 = take(f + 4), w3 = take(f + 5);
      acc[T] = CRNERF_MFMA_X(u3, b1, acc[T]);
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
#endif
    }
#ifdef CRNERF_X3_NOPIPE
    if (s + 1 < NS) prepare(s + 1, n1, n2, n3);
#endif
    b1 = n1; b2 = n2; b3 = n3;
  }
#pragma unroll
  for (int f = NS * NT * XNP; f < NS * NT * XNP + PAD; ++f) (void)take(f);
}

// the packed weights carry XWSCALE (h2 core): the bias goes into the accumulator scaled up, the layer output comes out scaled down (powers of two)
template <int NT>
__device__ __forceinline__ void xscale(f32x16 (&acc)[NT], float f) {
#ifdef CRNERF_EXP_H2_NOEPI   // (timing experiments only; garbage) the h2 core without its scale multiplications and range tracking
  return;
#endif
  if (XWSCALE != 1.0f) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= f;
  }
}
template <int NT>
__device__ __forceinline__ void init_acc_x(f32x16 (&acc)[NT], const lds_float* bias, int h) {
  init_acc<NT>(acc, bias, h);   // h2 packs carry their biases scaled by XWSCALE (pack_consts_kernel): a plain LDS read, straight into the accumulator
}
// amax (h2 core): running max |activation| of this lane.  An activation beyond fp16's range would split into (inf, -inf) pieces, turn into NaN in
// the next layer and be clamped to 0 by its relu -- a finite, wrong result.  mlp_tile_x3 therefore POISONS the outputs of such a point with NaN.
template <int NT, int NDST>
__device__ __forceinline__ void store_act_x(f32x16 (&acc)[NT], f32x16 (&act)[NDST], float floor_, float& amax) {
  xscale<NT>(acc, 1.0f / XWSCALE);
  store_act<NT>(acc, act, floor_);
#ifdef CRNERF_EXP_H2_NOEPI
  return;
#endif
  if (XNP == 2) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) amax = fmaxf(fmaxf(amax, fabsf(act[t][r])), fabsf(act[t][r + 1]));
  }
}
constexpr float H2_ACT_LIMIT = 65504.0f;   // largest finite fp16
// -DCRNERF_X_LAZY=1 (h2 core only): finish a layer's output inside the NEXT layer's walk (mma_layer_x3 FINA / FINB) instead of between the layers.
// MEASURED: correct (tests/test_gpu_h2.py green) and not faster -- 0.824 vs 0.78-0.82 ms per 1,024 rays, 3.39 vs 3.36 ms per 2^20 points: holding the
// raw rows across the layer costs registers (28 -> 105 spilled VGPRs) and the spill traffic eats what the overlap gives.  Off by default.
#ifndef CRNERF_X_LAZY
#define CRNERF_X_LAZY 0
#endif
constexpr bool XLAZY = CRNERF_X_LAZY != 0;
static_assert(!XLAZY || XNP == 2, "the lazy epilogue is the h2 core's");
// lazy variant of store_act_x: the raw accumulators leave the AGPRs, nothing else
template <int NT, int NDST>
__device__ __forceinline__ void store_raw_x(const f32x16 (&acc)[NT], f32x16 (&act)[NDST]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) act[t] = acc[t];
}
// eager finish in place (the sigma head reads h8 on the VALU)
template <int NT, int NDST>
__device__ __forceinline__ void finish_act_x(f32x16 (&act)[NDST], float floor_, float& amax) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      act[t][r] = fmaxf(act[t][r] * (1.0f / XWSCALE), floor_);
      act[t][r + 1] = fmaxf(act[t][r + 1] * (1.0f / XWSCALE), floor_);
      amax = fmaxf(fmaxf(amax, fabsf(act[t][r])), fabsf(act[t][r + 1]));
    }
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
  float amax = 0.0f;
  if (XNP == 2) {   // the embeddings are MMA operands too
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(pe[t][r]));
#pragma unroll
    for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(dv[0][r]));
  }
  tm.tick(T_PROLOGUE);

  init_acc_x<8>(acc, C + C_BIAS, h);                       // xyz_encoding_1
  mma_layer_x3<8, KS_XYZ, 0, 0>(p, pe, pe, acc, q);
  if (XLAZY) store_raw_x<8>(acc, act); else store_act_x<8>(acc, act, 0.0f, amax);
  sv.template masks<8>(0, act);
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {                          // xyz_encoding_2..4 (their input h_l is saved in slot l - 1 on the way)
    init_acc_x<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_layer_x3<8, KS_HID, 0, 0, SAVE, false, XLAZY ? 1 : 0, 0>(p, act, act, acc, q, sv.row(l - 1), SaveRowX{}, vo, &amax);
    if (XLAZY) store_raw_x<8>(acc, act); else store_act_x<8>(acc, act, 0.0f, amax);
    sv.template masks<8>(l, act);
  }
  init_acc_x<8>(acc, C + C_BIAS + 4 * W_HIDDEN, h);        // xyz_encoding_5 = Linear(cat[xyz, h])
  mma_layer_x3<8, KS_XYZ, KS_HID, 0, false, SAVE, 0, XLAZY ? 1 : 0>(p, pe, act, acc, q, SaveRowX{}, sv.row(3), vo, &amax);
  if (XLAZY) store_raw_x<8>(acc, act); else store_act_x<8>(acc, act, 0.0f, amax);
  sv.template masks<8>(4, act);
#pragma unroll 1
  for (int l = 5; l < 8; ++l) {                          // xyz_encoding_6..8
    init_acc_x<8>(acc, C + C_BIAS + l * W_HIDDEN, h);
    mma_layer_x3<8, KS_HID, 0, 0, SAVE, false, XLAZY ? 1 : 0, 0>(p, act, act, acc, q, sv.row(l - 1), SaveRowX{}, vo, &amax);
    if (XLAZY) store_raw_x<8>(acc, act); else store_act_x<8>(acc, act, 0.0f, amax);
    sv.template masks<8>(l, act);
  }
  if (XLAZY) finish_act_x<8>(act, 0.0f, amax);           // h8 is finished eagerly: the sigma head reads it on the VALU
  tm.tick(T_MMA);
  {                                                      // static_sigma: 256 -> 1 on the VALU (fp32, as mlp_core.h)
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
  init_acc_x<8>(acc, C + C_BFIN, h);                       // xyz_encoding_final (no activation); h8 -> slot 7
  mma_layer_x3<8, KS_HID, 0, 0, SAVE, false>(p, act, act, acc, q, sv.row(7), SaveRowX{}, vo);
  if (XLAZY) store_raw_x<8>(acc, act); else store_act_x<8>(acc, act, NEG_INF, amax);
  {
    f32x16 acc4[4];                                      // dir_encoding = relu(Linear(cat[final, dir])); final -> slot 8
    init_acc_x<4>(acc4, C + C_BDIR, h);
    mma_layer_x3<4, KS_HID, KS_DIR, XPAD_DIR, SAVE, false, XLAZY ? 2 : 0, 0>(p, act, dv, acc4, q, sv.row(8), SaveRowX{}, vo, &amax);
    if (XLAZY) store_raw_x<4>(acc4, act); else store_act_x<4>(acc4, act, 0.0f, amax);
    {
      f32x16 a4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a4[t] = act[t];
      sv.template masks<4>(9, a4);
    }
  }
  {
    f32x16 acc2[2];                                      // static_rgb = sigmoid(Linear); the dir activation -> slot 9
    init_acc_x<2>(acc2, C + C_BRGB, h);
    mma_layer_x3<2, KS_HALF, 0, 0, SAVE, false, XLAZY ? 1 : 0, 0>(p, act, act, acc2, q, sv.row(9), SaveRowX{}, vo, &amax);
    tm.tick(T_MMA);
    xscale<2>(acc2, 1.0f / XWSCALE);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) feat[t][r] = sigmoid_ref(acc2[t][r]);
    if (XNP == 2) {   // an operand left fp16's range somewhere in this point's MLP: NaN out, not a finite wrong answer
      amax = fmaxf(amax, __shfl_xor(amax, 32));
      if (!(amax < H2_ACT_LIMIT)) {
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
}

}  // inline namespace xcore_*
}  // namespace crnerf
