// sin(2^k x), cos(2^k x) for the positional embeddings (PosEmbedding.forward, models/nerf.py:17-30: freqs are exact powers of two, arguments reach
// 2^14 |x| ~ 8e4), branch-free and accurate to ~1.3 ulp -- 26 VALU instructions per (sin, cos) pair against ~135 for ocml's sincosf, whose
// large-argument path every frequency above 2^3 takes.  The 24 pairs per lane and tile were 7 % of the one-wave-per-SIMD renderers' cycles.
//
// Range reduction in REVOLUTIONS, once per coordinate:  x / (2 pi) = p + e  with p = fl(x C_HI), e = the exact rounding error of that product
// + x C_LO (C_HI + C_LO = 1 / (2 pi) to 2^-52): p + e is x / (2 pi) to ~2^-47 relative.  Scaling by the frequency is exact (a power of two), so
// 4 2^k p is exact, its distance to the nearest integer j (v_rndne_f32) is exact, and the reduced argument  y = (4 2^k p - j) / 4 + 2^k e  in
// [-1/8, 1/8] revolutions carries one rounding (<= 3.7e-9 revolutions).  theta = 2 pi y (two-constant product), sin / cos of theta in
// [-pi/4, pi/4] by the Cephes single-precision polynomials, then the quadrant j mod 4 swaps / negates.  Measured against float64 over
// x in [-6, 6], k = 0..16 (tests/test_split_numerics.py, the same arithmetic in numpy): max |error| 9e-8, mean 1.4e-8 -- ocml's sincosf
// is specified to 2 ulp; torch's CPU sin / cos (the reference) sit at <= 1 ulp, so the embeddings differ from the reference's by <= ~2e-7
// either way (tests/test_gpu_parity.py::test_posenc_golden: 2.5e-7).
// Valid while |2^(k+2) x / (2 pi)| < 2^23, i.e. |x| < 200 at k = 14 (the datasets normalise every scene to far = 5,
// datasets/phototourism_mask_grid_sample.py:139-145); beyond that the result degrades towards sin(0), finite.
#pragma once
#include <hip/hip_runtime.h>

namespace crnerf {

struct Rev2Pi { float p, e; };   // x / (2 pi) = p + e

__device__ __forceinline__ Rev2Pi to_rev2pi(float x) {
  const float C_HI = 0.159154937f;        // fl(1 / (2 pi))
  const float C_LO = 6.42063824e-09f;     // 1 / (2 pi) - C_HI
  Rev2Pi r;
  r.p = x * C_HI;
  r.e = fmaf(x, C_HI, -r.p) + x * C_LO;
  return r;
}

// s = sin(2^k x), c = cos(2^k x), t = to_rev2pi(x)
__device__ __forceinline__ void sincos_rev2pi(Rev2Pi t, int k, float& s, float& c) {
  const float TP_HI = 6.28318548f, TP_LO = -1.74845553e-07f;   // 2 pi = TP_HI + TP_LO
  const float p4 = ldexpf(t.p, k + 2);                         // exact
  const float j = __builtin_rintf(p4);                         // v_rndne_f32
  const float y = fmaf(p4 - j, 0.25f, ldexpf(t.e, k));         // reduced argument in revolutions, |y| <= 1/8 (+ the tiny e term)
  const float th = fmaf(y, TP_HI, y * TP_LO);
  const float z = th * th;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float sn = fmaf(th * z, ps, th);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z, -0.5f);
  const float cs = fmaf(z, pc, 1.0f);
  const int q = (int)j;                                        // quadrant = q mod 4 (two's complement: & 3 works for negative q)
  const bool swap = (q & 1) != 0;
  const float s0 = swap ? cs : sn, c0 = swap ? sn : cs;
  s = __uint_as_float(__float_as_uint(s0) ^ (((unsigned)q << 30) & 0x80000000u));          // q = 2, 3: negative
  c = __uint_as_float(__float_as_uint(c0) ^ (((unsigned)(q + 1) << 30) & 0x80000000u));    // q = 1, 2: negative
}

}  // namespace crnerf
