// Positional embedding computed straight into MFMA B-operand registers.
// Reference: PosEmbedding.forward, models/nerf.py:17-30 -- out = [x, sin(2^0 x), cos(2^0 x), ...],
// freqs = 2**linspace(0, N-1, N) (exact powers of two, so freq*x is exact in fp32).
//
// Slot order is layout.h's pair-interleaved order: lane half h computes argument
// a = 4*(m/2) + 2h + (m%2) for m = 0..M-1 and keeps (sin, cos) in registers (2m, 2m+1), so one
// sincos serves two embedding columns.  Arguments reach 2^14 * |x| ~ 1e5: the branch-free exact-reduction routine of
// sincos_pow2.h (~1.3 ulp; rounds 1-3 used ocml's sincosf, 2 ulp, five times the instructions) -- never the fast __sinf/__cosf.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"
#include "sincos_pow2.h"

namespace crnerf {

template <int F, int M>
__device__ __forceinline__ void posenc_regs(float x, float y, float z, int h, float (&out)[2 * M]) {
  const Rev2Pi rx = to_rev2pi(x), ry = to_rev2pi(y), rz = to_rev2pi(z);   // range reduction in revolutions, once per coordinate (sincos_pow2.h)
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int a0 = 4 * (m / 2) + (m % 2), a1 = a0 + 2;  // argument index for h = 0 / h = 1
    const bool trig0 = a0 < 3 * F, trig1 = a1 < 3 * F;
    float s = 0.0f, c = 0.0f;
    if (trig0 || trig1) {
      const int d0 = trig0 ? a0 % 3 : 0, d1 = trig1 ? a1 % 3 : 0;
      const Rev2Pi v0 = d0 == 0 ? rx : (d0 == 1 ? ry : rz);
      const Rev2Pi v1 = d1 == 0 ? rx : (d1 == 1 ? ry : rz);
      const int k0 = trig0 ? a0 / 3 : 0, k1 = trig1 ? a1 / 3 : 0;
      const Rev2Pi v = {h ? v1.p : v0.p, h ? v1.e : v0.e};
      sincos_rev2pi(v, h ? k1 : k0, s, c);
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
