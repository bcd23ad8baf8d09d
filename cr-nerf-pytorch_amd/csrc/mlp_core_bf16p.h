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
    if (i == 0) glds16(m0s[F % B_RING], src, voff, 0);            // writes M0; piece 1 reuses it with its instruction offset
    else glds16_more(src, voff, FRAG_BYTES);
  }
  __device__ __forceinline__ void cursor_update(int F) {
    voff = (F + 1 == STAGESB_PER_PASS) ? rd_base - LDS_RING : voff + STAGE_BYTES;   // = lane16, from the register that is live in every k-step
    asm volatile("" : "+v"(voff));                                                  // (lane16 itself is spilled in the training twin and came back behind a vmcnt(0))
  }
  // all_landed (training twin): the first tile's barriers use store windows that assume a previous tile; with its first three stages
  // already in LDS they have nothing left to wait for
  __device__ __forceinline__ void start(lds_char* lds_, const char* stream0, const char* stream1, int first_model, int lane, int wave, bool all_landed = false) {
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
    if (all_landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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
  // the wait counts LDS-DMA pieces only (two of this wave's may be in flight per open stage pair); row stores of the training twin share the counter
  // and are waited for with them -- see xcore_pipe.h advance() for why the per-window store allowance of rounds 3-4 is gone (0.7 % of the mixed step)
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

// ---- training twin (crnerf_render_rays_train_bf16): what the backward twins need is written from the registers it is born in.
// Saved state of a pass with P points (the buffer of crnerf_mlp_train_mixed_acts_bytes(P), mlp_gemm_bf16.hip):
//   rows  [10][P][256] bf16   the B operands themselves: k-step 2T + j of lane (p, h) is the 16-byte piece at byte 64T + 32j + 16h of the
//                             point's row ("fused" storage order: feature 32T + 8c + 4h + i at position 32T + 16(c>>1) + 8h + 4(c&1) + i;
//                             the weight-gradient kernel un-permutes when it writes dW, mlp_gemm_bf16.hip perm_fused)
//   bits  [10][P] x 32 B      relu-activity bits in the layout of linear_bf16_kernel (byte [q4][u], bit 4b + i <-> feature 32u + 16b + 4q4 + i):
//                             lane (p, h) owns q4 = h and q4 = 2 + h, one dword per four tiles
//   xb    [P][128] bf16       the embedded input as B operands, SLOT order (layout.h posenc_slot_to_col_b): xyz k-steps 0..5, dir k-steps 6, 7
//
// Rows leave through LDS in whole cache lines.  Stored straight from the registers a row piece is 32 contiguous bytes per point and
// instruction (two lanes per point), a 128-byte line is completed by four instructions of two different tiles, ~1 us apart: 2.76 ms per
// 2^20 points against 1.17 ms for the inference kernel (2.25 TB/s of saved state), and the same stores aimed at whole lines -- eight points
// x 128 bytes per instruction -- 2.16 ms (a round-3 tuning build).  So a wave parks the 64-byte pieces of two
// consecutive tiles in a private 32 x (128 + 16)-byte LDS block (ds_write_b128, conflict-free with the 16-byte row pad) and reads them back
// transposed: lane l takes bytes 16(l & 7).. of row 8e + (l >> 3), four instructions e = 0..3 per tile pair, each writing eight full lines.
// Every piece of that traffic rides in the MFMA loop of the following tiles (mma_layer_p: LDS writes in k-steps 9 / 11, activity bits in
// 10 / 12, the read -> store chain in k-steps 13, 15, 1, 3, 5): a store wave-instruction holds the CU's store path ~64 cycles, see ActSaver
// in mlp_train16.h.
//
// Stores and the weight ring share vmcnt (gfx9 has no separate store counter).  The ring's barrier in stage c must know that this wave's LDS-DMA
// pieces of stage c + 1 have landed; they were issued in stage c - 2, and the four pieces of stages c + 2 and c + 3 issued after them may stay in
// flight: vmcnt(4).  Every store issued since is waited for as well (rounds 3-4 counted them into the wait -- compile-time store windows --, which
// presumes that stores and LDS-DMA loads retire in issue order against each other; round 5 dropped that, xcore_pipe.h advance(): 0.7 % of the
// mixed step).  The stores stay UNCONDITIONAL raw-buffer stores (lanes without a point carry an out-of-range offset and the hardware drops them):
// no branch in the tile's static schedule.
constexpr uint32_t SAVE_OOB = 0xF0000000u;     // offset of a lane that stores nothing (>= every resource size; + instruction offsets stays < 2^32)
constexpr int SAVE_FLAGS = 0x00020000;         // buffer resource word 3 (gfx9: DATA_FORMAT_32), raw buffer: stride 0, range check on the byte offset
constexpr int SAVE_TILE_BURST = 8 + 9;         // between two tiles: the raw output row (8 x 16 B + sigma) and the embedded input (8 x 16 B)
constexpr int SAVE_LDS_ROW = 128 + 16;         // one point's bytes of a tile pair + pad (row stride 36 dwords: eight lanes' ds_write_b128 hit 32 different banks)
constexpr int SAVE_LDS_WAVE = 32 * SAVE_LDS_ROW;
struct NoSaveP {
  static constexpr bool on = false;
  __device__ __forceinline__ void drain() {}
};
struct ActSaveP {
  static constexpr bool on = true;
  const char* acts;      // scalar: this pass' buffer
  long slot_bytes;       // scalar: P * 512
  long P;                // scalar
  uint32_t boff;         // per lane: point * 32 + 8 h, or SAVE_OOB
  uint32_t toff;         // per lane: (the tile's first point + (lane >> 3)) * 512 + 16 (lane & 7): where the transposed stores go
  int rowlim;            // per lane: points of the tile - (lane >> 3); row 8e + (lane >> 3) of the tile is a point iff 8e < rowlim
  lds_char* lds;
  uint32_t lds_w;        // per lane: this wave's block + p * SAVE_LDS_ROW + 16 h
  uint32_t lds_r;        // per lane: this wave's block + (lane >> 3) * SAVE_LDS_ROW + 16 (lane & 7)
  // the tile pair being drained: rows of its slot, its byte column, and where the read -> store chain stands (5 = idle); all of it folds
  // to constants in the unrolled tile
  __amdgpu_buffer_rsrc_t pend;
  int pend_col = 0;
  int dstep = 5;
  u32x4 dval;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t rows(int slot) const {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(acts + slot * slot_bytes), 0, (int)(uint32_t)slot_bytes, SAVE_FLAGS);
  }
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bits(int slot) const {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(acts + 10 * slot_bytes + (long)slot * P * 32), 0, (int)(uint32_t)(P * 32), SAVE_FLAGS);
  }
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t xb() const {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(acts + 10 * slot_bytes + 10 * P * 32), 0, (int)(uint32_t)(P * 256), SAVE_FLAGS);
  }
  // one link of the chain (called in the k-steps p_drain_slot names): store the eight rows read last time, read the next eight
  __device__ __forceinline__ void drain() {
    if (dstep >= 5) return;
    if (dstep >= 1) {
      const int e = dstep - 1;
      const uint32_t off = 8 * e < rowlim ? toff + (uint32_t)(8 * e * 512 + pend_col) : SAVE_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(dval, pend, (int)off, 0, 0);
    }
    if (dstep <= 3) dval = *(const __attribute__((address_space(3))) u32x4*)(lds + lds_r + 8 * dstep * SAVE_LDS_ROW);
    ++dstep;
  }
};
constexpr bool p_drain_slot(int ns, int s) { return ns >= 16 ? (s == 1 || s == 3 || s == 5 || s == 13 || s == 15) : (s >= 1 && s <= 5); }

// activity byte of four packed post-relu dwords: bit 2d + e <-> half e of dword d is non-zero
__device__ __forceinline__ uint32_t act_byte(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 one = {1, 1};
  auto nz = [&](uint32_t d) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, d), one)); };   // v_pk_min_u16
  uint32_t m = nz(d0) | (nz(d1) << 2);
  m |= nz(d2) << 4;
  m |= nz(d3) << 6;
  return (m | (m >> 15)) & 0xffu;
}

template <class SV>
struct SaveSlot {   // per-epilogue state of the training twin; empty at inference
  __amdgpu_buffer_rsrc_t rows, bitp;
  uint32_t bA = 0, bB = 0;
  template <bool BITS>
  __device__ __forceinline__ void set(const SV& sv, int slot) {
    rows = sv.rows(slot);
    if (BITS) bitp = sv.bits(slot);
  }
  // step 0 / 2: the two row pieces of tile T go to the wave's LDS block (the second piece of an odd tile starts the drain chain);
  // step 1 / 3: the activity bytes of q4 = h / q4 = 2 + h (+ the dword stores behind every 4th tile).  p_make_stores() mirrors this.
  template <bool BITS>
  __device__ __forceinline__ void step(SV& sv, const u32x4 (&dst)[KS_HID], int T, int st) {
    if (st == 0 || st == 2) {
      const int j = st >> 1;
      *(__attribute__((address_space(3))) u32x4*)(sv.lds + sv.lds_w + (64 * (T & 1) + 32 * j)) = dst[2 * T + j];
      if ((T & 1) && j == 1) {
        sv.pend = rows;
        sv.pend_col = 128 * (T >> 1);
        sv.dstep = 0;
      }
    } else if (BITS) {
      const int e = st == 1 ? 0 : 2;
      const uint32_t by = act_byte(dst[2 * T][e], dst[2 * T][e + 1], dst[2 * T + 1][e], dst[2 * T + 1][e + 1]);
      uint32_t& acc = st == 1 ? bA : bB;
      acc = (T & 3) == 0 ? by : (acc | (by << (8 * (T & 3))));
      if ((T & 3) == 3) __builtin_amdgcn_raw_buffer_store_b32(acc, bitp, (int)(sv.boff + (uint32_t)((st == 1 ? 0 : 16) + 4 * (T >> 2))), 0, 0);
    }
  }
};
template <>
struct SaveSlot<NoSaveP> {
  template <bool BITS>
  __device__ __forceinline__ void set(const NoSaveP&, int) {}
  template <bool BITS>
  __device__ __forceinline__ void step(NoSaveP&, const u32x4 (&)[KS_HID], int, int) {}
};

// ---- the static store schedule of one tile (what ActSaveP::drain / SaveSlot::step issue where mma_layer_p calls them), for the ring's vmcnt
struct PLayerSched { int fbase, nt, ns, prev_kind, epi_kind, pt; };   // kind 0: stores nothing, 1: rows, 2: rows + activity bits
constexpr PLayerSched P_SCHED[11] = {
    {OFFB_L1, 8, KS_XYZ, 0, 2, 0},          {OFFB_L2, 8, KS_HID, 2, 2, 7},           {OFFB_L2 + FB_HID, 8, KS_HID, 2, 2, 7},
    {OFFB_L2 + 2 * FB_HID, 8, KS_HID, 2, 2, 7}, {OFFB_L5, 8, KS_XYZ + KS_HID, 2, 2, 7}, {OFFB_L6, 8, KS_HID, 2, 2, 7},
    {OFFB_L6 + FB_HID, 8, KS_HID, 2, 2, 7}, {OFFB_L6 + 2 * FB_HID, 8, KS_HID, 2, 2, 7}, {OFFB_FIN, 8, KS_HID, 2, 1, 7},
    {OFFB_DIR, 4, KS_HID + KS_DIR, 1, 2, 7}, {OFFB_RGB, 2, KS_HALF, 2, 0, 3}};
struct PStoreTable { int n[STREAMB_USED]; bool ok; };
constexpr PStoreTable p_make_stores() {   // n[i]: store instructions issued in k-step i (pass-relative fragment index)
  PStoreTable t{};
  t.ok = true;
  int dstep = 5;
  for (int l = 0; l < 11; ++l) {
    const PLayerSched& L = P_SCHED[l];
    for (int T = 0; T < L.nt; ++T)
      for (int s = 0; s < L.ns; ++s) {
        const int i = L.fbase + T * L.ns + s;
        if (p_drain_slot(L.ns, s) && dstep < 5) {   // ActSaveP::drain
          if (dstep >= 1) ++t.n[i];
          ++dstep;
        }
        const int kind = T == 0 ? L.prev_kind : L.epi_kind, tile = T == 0 ? L.pt : T - 1;
        if (kind == 0) continue;
        const int bits = (kind == 2 && (tile & 3) == 3) ? 1 : 0;
        const bool last_piece = L.ns >= 16 ? s == 11 : s == L.ns - 1;
        if (L.ns >= 16) t.n[i] += (s == 10 || s == 12) ? bits : 0;
        else if (s == L.ns - 1) t.n[i] += 2 * bits;
        if (last_piece && (tile & 1)) {
          if (dstep != 5) t.ok = false;        // the previous pair must have left the LDS block
          dstep = 0;
        }
      }
  }
  if (dstep != 5) t.ok = false;                // nothing pending across tiles
  return t;
}
constexpr PStoreTable P_STORES = p_make_stores();
constexpr int p_stores_at(int i) { return P_STORES.n[i]; }
constexpr int p_store_total() {
  int n = 0;
  for (int i = 0; i < STREAMB_USED; ++i) n += p_stores_at(i);
  return n;
}
static_assert(P_STORES.ok && p_store_total() == 9 * 16 + 8 + 8 * 4 + 2, "store schedule: 38 tile pairs x 4 line stores, activity dwords 8 x 4 + dir 2");

// ---- epilogues, one point group: quarter qc (0..7) = accumulator registers 2qc, 2qc+1 -> dword qc&3 of k-step
// 2T + (qc>>2) of the next layer's B operand.  The accumulators live in ARCHITECTURAL registers here (gfx950 MFMA takes
// VGPRs for C/D): an "a" constraint anywhere in the kernel makes hipcc split the 256-register budget of a two-waves-per-SIMD
// kernel 128 VGPR + 128 AGPR (SIRegisterInfo: usesAGPRs => MaxNumVGPRs /= 2), which does not hold the two 64-register
// activation buffers -- and a VGPR accumulator needs no v_accvgpr_read: a quarter is TWO VALU instructions.  The asm
// statements pin each quarter to the k-step it was written in (see PackEpi in mlp_core_bf16.h).
struct NoEpiP {
  static constexpr bool saving = false;
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void finish(int, int, const f32x16&) {}
  __device__ __forceinline__ void save_step(int, int) {}
};

template <bool RELU, class SV = NoSaveP>
struct PackEpiP {
  static constexpr bool saving = SV::on;
  u32x4 (&dst)[KS_HID];
  SV& sv;
  SaveSlot<SV> ss;
  __device__ __forceinline__ PackEpiP(u32x4 (&d)[KS_HID], SV& s) : dst(d), sv(s) {}
  __device__ __forceinline__ void slot(int sl) { ss.template set<RELU>(sv, sl); }
  __device__ __forceinline__ void save_step(int T, int st) { ss.template step<RELU>(sv, dst, T, st); }
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
template <class SV = NoSaveP>
struct SigmaEpiP {
  static constexpr bool saving = SV::on;
  u32x4 (&dst)[KS_HID];
  float& sg;
  const lds_float* wsig;
  int h;
  SV& sv;
  SaveSlot<SV> ss;
  f32x4 wv[4];
  __device__ __forceinline__ SigmaEpiP(u32x4 (&d)[KS_HID], float& s, const lds_float* w, int h_, SV& v) : dst(d), sg(s), wsig(w), h(h_), sv(v) {}
  __device__ __forceinline__ void slot(int sl) { ss.template set<true>(sv, sl); }
  __device__ __forceinline__ void save_step(int T, int st) { ss.template step<true>(sv, dst, T, st); }
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
  static constexpr bool saving = false;
  f32x16 (&feat)[2];
  __device__ __forceinline__ explicit RgbEpiP(f32x16 (&f)[2]) : feat(f) {}
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void save_step(int, int) {}
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
template <int NT, int NSA, int NSB, int FBASE, int G0, int PT, int NA, int NB, class PREV, class EPI, class SV>
__device__ __forceinline__ void mma_layer_p(WeightPipeP& p, const u32x4 (&srcA)[NA], const u32x4 (&srcB)[NB], u32x4 (&q)[B_AHEAD],
                                            f32x16 (&accs)[2], const lds_float* bias, const lds_float* next_bias, int h, PREV& prev,
                                            EPI& epi, SV& sv) {
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
      q[i % B_AHEAD] = p.read(b_pos(i + B_AHEAD));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < count; ++u) {
        if (T == 0) prev.finish(PT, first + u, pa);
        else epi.finish(T - 1, first + u, pa);
      }
      if (s == 0) {
        if (T == 0) prev.prefetch(PT);
        else epi.prefetch(T - 1);
      }
      if (SV::on && p_drain_slot(NS, s)) sv.drain();   // training twin: a tile pair's lines on their way out (no-op at inference)
      {   // training twin: the previous tile's row pieces / activity bits, behind its last epilogue quarter (no-ops at inference)
        const int st = LONG ? s - 9 : -1;
        if (LONG && st >= 0 && st < 4) {
          if (T == 0) prev.save_step(PT, st);
          else epi.save_step(T - 1, st);
        }
        if (!LONG && s == NS - 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (T == 0) prev.save_step(PT, u);
            else epi.save_step(T - 1, u);
          }
        }
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
template <class SV = NoSaveP>
__device__ __forceinline__ void mlp_tile_p(WeightPipeP& p, int model, int next_model, const u32x4 (&pe)[KS_XYZ], const lds_char* dirsrc,
                                           f32x16 (&feat)[2], float& sigma, int h, u32x4 (&q)[B_AHEAD], PhaseTimer& tm, SV& sv) {
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

  // slot(k): where the training twin keeps the output of the layer this epilogue closes (h1..h8 = 0..7, final = 8, dir act = 9); an epilogue's
  // last tile is stored from the NEXT layer's first tile, so the slot is switched between that layer and the epilogue's next own layer
  NoEpiP none;
  PackEpiP<true, SV> eA(actA, sv), eB(actB, sv);
  PackEpiP<false, SV> efin(actA, sv);
  SigmaEpiP<SV> e8(actB, sg, C + C_WSIG, h, sv);
  RgbEpiP ergb(feat);
  eA.slot(0);
  mma_layer_p<8, KS_XYZ, 0, OFFB_L1, 0, 0>(p, pe, pe, q, accs, B1, B1 + 1 * W_HIDDEN, h, none, eA, sv);                                // xyz_encoding_1
  eB.slot(1);
  mma_layer_p<8, KS_HID, 0, OFFB_L2, 8, 7>(p, actA, actA, q, accs, B1 + 1 * W_HIDDEN, B1 + 2 * W_HIDDEN, h, eA, eB, sv);               // 2
  eA.slot(2);
  mma_layer_p<8, KS_HID, 0, OFFB_L2 + FB_HID, 16, 7>(p, actB, actB, q, accs, B1 + 2 * W_HIDDEN, B1 + 3 * W_HIDDEN, h, eB, eA, sv);     // 3
  eB.slot(3);
  mma_layer_p<8, KS_HID, 0, OFFB_L2 + 2 * FB_HID, 24, 7>(p, actA, actA, q, accs, B1 + 3 * W_HIDDEN, B1 + 4 * W_HIDDEN, h, eA, eB, sv); // 4
  eA.slot(4);
  mma_layer_p<8, KS_XYZ, KS_HID, OFFB_L5, 32, 7>(p, pe, actB, q, accs, B1 + 4 * W_HIDDEN, B1 + 5 * W_HIDDEN, h, eB, eA, sv);           // 5 = Linear(cat[xyz, h])
  eB.slot(5);
  mma_layer_p<8, KS_HID, 0, OFFB_L6, 40, 7>(p, actA, actA, q, accs, B1 + 5 * W_HIDDEN, B1 + 6 * W_HIDDEN, h, eA, eB, sv);              // 6
  eA.slot(6);
  mma_layer_p<8, KS_HID, 0, OFFB_L6 + FB_HID, 48, 7>(p, actB, actB, q, accs, B1 + 6 * W_HIDDEN, B1 + 7 * W_HIDDEN, h, eB, eA, sv);     // 7
  e8.slot(7);
  mma_layer_p<8, KS_HID, 0, OFFB_L6 + 2 * FB_HID, 56, 7>(p, actA, actA, q, accs, B1 + 7 * W_HIDDEN, C + C_BFIN, h, eA, e8, sv);        // 8 (+ static_sigma)
  efin.slot(8);
  mma_layer_p<8, KS_HID, 0, OFFB_FIN, 64, 7>(p, actB, actB, q, accs, C + C_BFIN, C + C_BDIR, h, e8, efin, sv);                         // xyz_encoding_final
  tm.tick(T_MMA);
  sg += __shfl_xor(sg, 32);
  sigma = softplus_fast(sg + C[C_BSIG]);
  u32x4 dv[KS_DIR];
#pragma unroll
  for (int s = 0; s < KS_DIR; ++s) dv[s] = *(const __attribute__((address_space(3))) u32x4*)(dirsrc + 32 * s);
  tm.tick(T_SIGMA);
  eB.slot(9);
  mma_layer_p<4, KS_HID, KS_DIR, OFFB_DIR, 72, 7>(p, actA, dv, q, accs, C + C_BDIR, C + C_BRGB, h, efin, eB, sv);                      // dir_encoding
  mma_layer_p<2, KS_HALF, 0, OFFB_RGB, 76, 3>(p, actB, actB, q, accs, C + C_BRGB, C + C_BRGB, h, eB, ergb, sv);                         // static_rgb
  tm.tick(T_MMA);
#pragma unroll
  for (int qc = 0; qc < 8; ++qc) {   // rgb's last tile: nothing left to hide it behind
    ergb.finish(1, qc, accs[(76 + 1) & 1]);
  }
  tm.tick(T_EPILOGUE);
}

}  // namespace crnerf
