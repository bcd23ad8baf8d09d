// Saved-activation layout of the training twins (shared by mlp_train16.hip and the fused training renderer in
// render_fused16.hip): acts[10][P][256] fp32 in reference feature order (h1..h8, xyz_encoding_final, dir_encoding output), then
// the relu-activity bits masks[10][P][4] x 64 bit.
#pragma once
#include <hip/hip_runtime.h>
#include "mlp_core16.h"

namespace crnerf {

constexpr int ACT_SLOTS = 10;   // h1..h8, final, dir_act(128 used)
constexpr int ACT_W = 256;
constexpr int WG_RANGE_USED = ACT_SLOTS + 1;   // range words of a backward (launch_mlp_backward): max |delta| bits per delta slot, then d_rgb's

// relu-activity bits of the saved activations: masks[slot][point][g] = 64 bits, bit 4T + r <-> feature 16T + 4g + r.
// The backward-data kernel reads these 32 B per point and layer (all ten layers in ONE load batch per tile) instead of
// re-reading the 1 KiB activation row behind every layer -- each of those reads was consumed on the spot, i.e. a
// s_waitcnt vmcnt(0) that also drained the weight prefetch, ten times per tile.
__device__ __forceinline__ unsigned long long* mask_slot(float* acts, long P, int slot, long n, int g) {
  return (unsigned long long*)(acts + (size_t)ACT_SLOTS * P * ACT_W) + ((size_t)slot * P + n) * 4 + g;
}

// Stores of a 16-point tile's layer output: 16 row pieces (64 B per point and instruction) + the relu bits.  A global store
// wave-instruction occupies the CU's store data path for ~64 cycles (1 KiB at 16 B/clk), and a wave cannot issue anything else
// until its store has been taken: issued as a burst behind a layer (8 waves x 17 stores, all waves at the same program point)
// the matrix pipe idled ~8.7 k cycles per layer = 11 % of the forward (ablations, 2^20 points: no stores 9.09 ms, burst 10.30 ms,
// the same burst into an L2-resident window 10.16 ms, the ring's vmcnt wait relaxed by 20: 10.28 ms -- neither HBM bandwidth nor
// the vmcnt coupling).  So the pieces are DEFERRED into the next layer's MFMA loop, one piece every few fragment steps
// (mma_layer16's `def` argument): the 64 cycles of each store pass under the MFMAs of both waves of the SIMD.
struct ActSaver {
  float* base; long P; long n; bool valid; int g;
  // relu bits of the whole tile + the row pointer; call once, before the pieces
  template <int NT>
  __device__ __forceinline__ float* begin(int slot, const f32x4 (&a)[NT]) const {
    float* row = base + ((long)slot * P + n) * ACT_W + 4 * g;
    if (slot != 8) {   // xyz_encoding_final is linear: no mask
      const int nt = slot == 9 ? 8 : NT;
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int T = 0; T < NT; ++T)
        if (T < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t bit = a[T][r] > 0.0f ? 1u : 0u;       // post-relu values: > 0 <=> pre-activation > 0
            const int k = 4 * T + r;
            if (k < 32) lo |= bit << k; else hi |= bit << (k - 32);
          }
        }
      if (valid) *mask_slot(base, P, slot, n, g) = ((unsigned long long)hi << 32) | lo;
    }
    return row;
  }
  __device__ __forceinline__ void piece(float* row, int T, const f32x4& v) const {
    if (valid) *(f32x4*)(row + 16 * T) = v;
  }
  // everything at once (module-entry tail, tests)
  template <int NT>
  __device__ __forceinline__ void operator()(int slot, const f32x4 (&a)[NT]) const {
    float* row = begin(slot, a);
    const int nt = slot == 9 ? 8 : NT;
#pragma unroll
    for (int T = 0; T < NT; ++T)
      if (T < nt) piece(row, T, a[T]);
  }
};

}  // namespace crnerf
