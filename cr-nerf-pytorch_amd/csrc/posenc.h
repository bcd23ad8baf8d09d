// Positional embedding computed straight into MFMA B-operand registers.
// Reference: PosEmbedding.forward, models/nerf.py:17-30 -- out = [x, sin(2^0 x), cos(2^0 x), ...],
// freqs = 2**linspace(0, N-1, N) (exact powers of two, so freq*x is exact in fp32).
//
// Slot order is layout.h's pair-interleaved order: lane half h computes argument
// a = 4*(m/2) + 2h + (m%2) for m = 0..M-1 and keeps (sin, cos) in registers (2m, 2m+1), so one
// sincosf serves two embedding columns.  Arguments reach 2^14 * |x| ~ 1e5, so the accurate
// (Payne-Hanek capable) ocml sincosf is used -- never the fast __sinf/__cosf.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

namespace crnerf {

template <int F, int M>
__device__ __forceinline__ void posenc_regs(float x, float y, float z, int h, float (&out)[2 * M]) {
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int a0 = 4 * (m / 2) + (m % 2), a1 = a0 + 2;  // argument index for h = 0 / h = 1
    const bool trig0 = a0 < 3 * F, trig1 = a1 < 3 * F;
    float s = 0.0f, c = 0.0f;
    if (trig0 || trig1) {
      const int d0 = trig0 ? a0 % 3 : 0, d1 = trig1 ? a1 % 3 : 0;
      const float v0 = d0 == 0 ? x : (d0 == 1 ? y : z);
      const float v1 = d1 == 0 ? x : (d1 == 1 ? y : z);
      const float f0 = (float)(1 << (trig0 ? a0 / 3 : 0)), f1 = (float)(1 << (trig1 ? a1 / 3 : 0));
      const float arg = h ? f1 * v1 : f0 * v0;
      sincosf(arg, &s, &c);
    }
    // non-trig slots: a == 3F -> (x, y); a == 3F+1 -> (z, 0); beyond -> (0, 0)
    const float e0_0 = trig0 ? s : (a0 == 3 * F ? x : (a0 == 3 * F + 1 ? z : 0.0f));
    const float e0_1 = trig0 ? c : (a0 == 3 * F ? y : 0.0f);
    const float e1_0 = trig1 ? s : (a1 == 3 * F ? x : (a1 == 3 * F + 1 ? z : 0.0f));
    const float e1_1 = trig1 ? c : (a1 == 3 * F ? y : 0.0f);
    out[2 * m + 0] = h ? e1_0 : e0_0;
    out[2 * m + 1] = h ? e1_1 : e0_1;
  }
}

}  // namespace crnerf
