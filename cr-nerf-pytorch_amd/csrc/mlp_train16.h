// Saved-activation layout of the training twins (shared by mlp_train16.hip and the fused training renderer in
// render_fused16.hip): acts[10][P][256] fp32 in reference feature order (h1..h8, xyz_encoding_final, dir_encoding output), then
// the relu-activity bits masks[10][P][4] x 64 bit.
#pragma once
#include <hip/hip_runtime.h>
#include "mlp_core16.h"

namespace crnerf {

constexpr int ACT_SLOTS = 10;   // h1..h8, final, dir_act(128 used)
constexpr int ACT_W = 256;

// relu-activity bits of the saved activations: masks[slot][point][g] = 64 bits, bit 4T + r <-> feature 16T + 4g + r.
// The backward-data kernel reads these 32 B per point and layer (all ten layers in ONE load batch per tile) instead of
// re-reading the 1 KiB activation row behind every layer -- each of those reads was consumed on the spot, i.e. a
// s_waitcnt vmcnt(0) that also drained the weight prefetch, ten times per tile.
__device__ __forceinline__ unsigned long long* mask_slot(float* acts, long P, int slot, long n, int g) {
  return (unsigned long long*)(acts + (size_t)ACT_SLOTS * P * ACT_W) + ((size_t)slot * P + n) * 4 + g;
}

struct ActSaver {
  float* base; long P; long n; bool valid; int g;
  template <int NT>
  __device__ __forceinline__ void operator()(int slot, const f32x4 (&a)[NT]) const {
    if (!valid) return;
    float* row = base + ((long)slot * P + n) * ACT_W + 4 * g;
    const int nt = slot == 9 ? 8 : NT;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int T = 0; T < NT; ++T)
      if (T < nt) {
        *(f32x4*)(row + 16 * T) = a[T];
        if (slot != 8) {   // xyz_encoding_final is linear: no mask
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t bit = a[T][r] > 0.0f ? 1u : 0u;       // post-relu values: > 0 <=> pre-activation > 0
            const int k = 4 * T + r;
            if (k < 32) lo |= bit << k; else hi |= bit << (k - 32);
          }
        }
      }
    if (slot != 8) *mask_slot(base, P, slot, n, g) = ((unsigned long long)hi << 32) | lo;
  }
};

}  // namespace crnerf
