// What the two fp32-accurate split cores share (mlp_core_x3.h: three bf16 pieces per operand; mlp_core_h2.h: two fp16 pieces): the piece
// count and its constants, the weight ring, and the row-saving hooks of the training twins.  Include through one of the two cores.
//
// CRNERF_X_NP: pieces per operand.  3 (default) = the bf16 x3 core.  2 = the "h2" core (mlp_forward_h2.hip, render_fused_h2.hip):
// x = h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) (11 + 11 mantissa bits and the sign of h2 -- one fp32 rounding, as long as h2 is a normal
// fp16 number; fp16 subnormals are honoured by the matrix cores, so below that the ABSOLUTE error is <= 2^-25), and a product is the THREE
// leading piece products w2 a1 + w1 a2 + w1 a1 (the dropped w2 a2 is <= 2^-24 of it): half the MFMAs of the x3 core and two thirds of its
// weight stream (layout.h "fragH": weights AND biases scaled by 2^8 at pack time so that the weights' second pieces stay normal).
// Each variant lives in its own inline namespace.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"
#include "mlp_core.h"

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

// The ring: mlp_core.h's protocol with SEVEN 16-KiB slots.  The 3.75 MB x3 stream of a model does not stay in an XCD's 4 MB L2, so a piece's
// latency is the memory side's, not L2's, and what hides it is the number of stages in flight (the protocol keeps four out of flight: c - 1
// being refilled, c and c + 1 readable, c + 2 certified).  Measured per 1,024 rays (x3): 6 x 16 KiB 1.38 ms, 7 x 16 KiB 1.34 ms, 14 x 8 KiB 1.36 ms.
constexpr int X_RING = 7;
constexpr int X_STAGE_FRAGS = 16;
constexpr int X_STAGE_BYTES = X_STAGE_FRAGS * FRAG_BYTES;
constexpr int X_PIECES = X_STAGE_FRAGS / 4;        // 1 KiB pieces per wave and stage (four waves)
static_assert(STAGE_FRAGS % X_STAGE_FRAGS == 0 && X_PIECES >= 1 && X_PIECES <= 4, "x stages divide the stream's 16-fragment alignment");
constexpr int LDS_SCRATCH_X = LDS_RING + X_RING * X_STAGE_BYTES + 1024;
static_assert(X_RING >= 5 && X_PIECES * (X_RING - 3) + 8 <= 63 && LDS_SCRATCH_X + 4 * 5120 <= 160 * 1024, "x ring: protocol depth and LDS budget (ring + four waves of ray scratch)");
// fragments read ahead of the one being multiplied: x3 -- the piece groups of one tile PAIR (12 MFMAs = 384 matrix-pipe cycles); h2 -- of FOUR tiles
// (12 MFMAs too: that core refills a tile's two queue slots behind the tile's last MFMA)
#ifndef CRNERF_H2_AHEAD
#define CRNERF_H2_AHEAD 8
#endif
constexpr int X_AHEAD = XNP == 2 ? CRNERF_H2_AHEAD : 6;
static_assert(X_AHEAD % XNP == 0 && X_AHEAD <= X_STAGE_FRAGS, "the queue holds whole piece groups and never reaches past the next stage");

// WeightPipe of mlp_core.h for the split streams: stages_per_pass stages per pass, LDS-DMA as asm.  Protocol as there: stages c and c + 1 may be
// read; advance() -- after the last read of stage c has been issued -- waits until this wave's pieces of stage c + 2 have landed (counted
// vmcnt) and barriers; the slot of stage c - 1 is then refilled with stage c + X_RING - 1 by four issue_piece() calls during stage c + 1.
struct WeightPipeX {
  lds_char* lds;
  const char* base[2];   // scalar: packed streams + this wave's 4 KiB column
  const char* pf_ptr;
  int pf_left, pf_pass, passes0, passes;
  int stages_per_pass = XSTREAM_FRAGS / X_STAGE_FRAGS;   // stages per pass: forward stream (set_stream_frags() for another)
  uint32_t pf_slot, rd_slot, rd_addr, lane16, lds_ring;
  int wave;   // scalar: this wave's index in the workgroup
  __device__ __forceinline__ void set_stream_frags(int frags) { stages_per_pass = frags / X_STAGE_FRAGS; }   // before start()

  // piece i (0 .. X_PIECES - 1) of the stage being fetched: this wave's fragments X_PIECES * wave + i.  Piece 0 writes M0 (the LDS destination of
  // the wave's column of the slot); pieces 1.. reuse it with their instruction offset, which applies to both sides.  Nothing else in the kernels
  // built on this pipe touches M0 (tests/test_host.py checks the ISA), so the four pieces of a stage may be any number of instructions apart.
  __device__ __forceinline__ void issue_piece(int i) {
    switch (i) {   // the instruction offset must be an immediate
      case 0: glds16(lds_ring + pf_slot * X_STAGE_BYTES, pf_ptr, lane16, 0); break;
      case 1: glds16_more(pf_ptr, lane16, FRAG_BYTES); break;
      case 2: glds16_more(pf_ptr, lane16, 2 * FRAG_BYTES); break;
      default: glds16_more(pf_ptr, lane16, 3 * FRAG_BYTES); break;
    }
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
  __device__ __forceinline__ void start(lds_char* lds_, const char* stream0, const char* stream1, int passes0_, int passes_, int lane, int wave_) {
    lds = lds_;
    wave = wave_;
    lane16 = (uint32_t)lane * 16u;
    const uint32_t wave4k = (uint32_t)wave_ * (X_PIECES * FRAG_BYTES);   // this wave's column of a stage
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
#pragma unroll
    for (int s = 0; s < X_RING - 1; ++s)
#pragma unroll
      for (int i = 0; i < X_PIECES; ++i) issue_piece(i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(X_PIECES * (X_RING - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  __device__ __forceinline__ uint32_t next_addr() const {
    const uint32_t n = (rd_slot + 1 == X_RING) ? 0u : rd_slot + 1;
    return LDS_RING + n * X_STAGE_BYTES + lane16;
  }
  // The wait counts LDS-DMA pieces only: three stages of them may be in flight.  Row stores of the training twins share the counter, so a window
  // that holds stores waits for those as well.  Rounds 3-4 allowed up to 8 of a window's stores on top (their counts are compile-time constants);
  // that is sound only if stores and LDS-DMA loads retire in issue order against EACH OTHER, and round 5 has a counter-example at larger
  // allowances (the x3 repair kernel differed from the x3 twin with the two-tile layers' 20 allowed).  With the plain wait only the order of
  // loads among themselves matters.  What the allowance was worth once the rows leave in bursts: 0.2 % on the x3 twin, 0.7 % on the h2 twin,
  // 1.5 % on the h2 backward (profiles/r5/row_store_experiments.txt).
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(X_PIECES * (X_RING - 4)) : "memory");
    __builtin_amdgcn_s_barrier();
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

// ---- training twins (crnerf_render_rays_train_f32x3 / _f32h2): the saved state of the fp32 training twins (mlp_train16.h: acts[10][P][256] fp32 in
// reference feature order, then the relu-activity bits masks[10][P][4] x 64 bit, bit 4T + r <-> feature 16T + 4g + r), written from the split cores'
// registers so that the fp32 backward twins (mlp_backward16_kernel, the weight-gradient kernels) read it unchanged.  A layer's output is stored when
// the NEXT layer walks it as its B operand -- the eight values of k-step s are two 16-byte pieces (64 contiguous bytes per point with the other lane
// half), two stores per k-step, spread through that layer's MFMAs.
constexpr uint32_t SAVEX_OOB = 0xF0000000u;   // offset of a lane that stores nothing (beyond every buffer resource: the hardware drops the store)
struct SaveRowX { __amdgpu_buffer_rsrc_t rs; };   // the rows of one slot
struct NoSaveX {
  static constexpr bool on = false;
  __device__ __forceinline__ SaveRowX row(int) const { return SaveRowX{}; }
  __device__ __forceinline__ uint32_t offset() const { return 0; }
  template <int NT>
  __device__ __forceinline__ void masks(int, const f32x16 (&)[NT]) const {}
  __device__ __forceinline__ void mask_words(int, int, uint32_t, uint32_t) const {}
  __device__ __forceinline__ void note_range(lds_char*, float) const {}
};
struct ActSaveX {
  static constexpr bool on = true;
  float* base; long P; long n; bool valid; int h;
  uint32_t range_lds = 0;     // LDS byte offset of this lane's range word for this pass (h2 training twin), 0: none
  // the h2 training form's largest |operand| of the tile (an Inf when the point left fp16's range): kept as a running maximum in the lane's own LDS
  // word -- one ds_max per tile, no register lives across the MLP; positive floats order like their bit patterns
  __device__ __forceinline__ void note_range(lds_char* lds, float amax) const {
    if (range_lds) __hip_atomic_fetch_max((__attribute__((address_space(3))) uint32_t*)(lds + range_lds), __float_as_uint(amax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // row stores are UNCONDITIONAL raw-buffer stores (lanes without a point carry an out-of-range offset): no branch in the layer's unrolled stream
  __device__ __forceinline__ SaveRowX row(int slot) const {
    return SaveRowX{__builtin_amdgcn_make_buffer_rsrc(base + (long)slot * P * 256, 0, (int)(uint32_t)(P * 1024), 0x00020000)};
  }
  __device__ __forceinline__ uint32_t offset() const {
    return valid ? (uint32_t)n * 1024u + 16u * (uint32_t)h : SAVEX_OOB;
  }
  // lane (p, h) owns the mask words g = h and g = h + 2: bit 4T + r <-> register 4q + r of tile T >> 1, q = 2 (T & 1) + (g >> 1)
  template <int NT>
  __device__ __forceinline__ void masks(int slot, const f32x16 (&a)[NT]) const {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      // post-relu values are >= +0, so value > 0 <=> its bit pattern != 0 <=> 0 - pattern has its top bit set; v_alignbit shifts the word left and
      // takes that bit in -- v_sub + v_alignbit per value, no SGPR in the chain -- with the bits walked from the top so that bit k ends up at k.
      // (Round 5: as compare -> SGPR pair -> select -> or3 the same bits cost the h2 forward twin 7 % of its time, 5.58 -> 5.19 ms per 2^20 points.)
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int k = (4 * 2 * NT < 64 ? 4 * 2 * NT : 64) - 1; k >= 0; --k) {
        const int T = k >> 2, r = k & 3;
        const uint32_t neg = 0u - __float_as_uint(a[T >> 1][4 * (2 * (T & 1) + gg) + r]);
        if (k < 32) lo = __builtin_amdgcn_alignbit(lo, neg, 31); else hi = __builtin_amdgcn_alignbit(hi, neg, 31);
      }
      mask_words(slot, gg, lo, hi);
    }
  }
  __device__ __forceinline__ void mask_words(int slot, int gg, uint32_t lo, uint32_t hi) const {
    unsigned long long* m = (unsigned long long*)(base + (size_t)10 * P * 256) + ((size_t)slot * P + n) * 4;
    if (valid) m[h + 2 * gg] = ((unsigned long long)hi << 32) | lo;
  }
};

}  // inline namespace xcore_*
}  // namespace crnerf
