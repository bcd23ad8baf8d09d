// Counter-based random draws for the renderer's two stochastic steps (models/rendering.py): the stratified jitter of the coarse
// depths (:169-176, torch.rand_like), the sample_pdf uniforms when perturb > 0 (:30, torch.rand) and the density noise (:125,
// torch.randn_like).  Philox4x32-10 keyed on the caller's seed, counter = (sample, stream, ray): every draw is a pure function of
// (seed, stream, global ray index, sample index), so ray chunks, recomputation in the backward and any launch geometry see the
// same numbers.  The reference draws from torch's global generator; its stream cannot be reproduced (and differs between its own
// CPU and CUDA runs) -- what is kept is the distribution: U[0,1) on a 2^-24 grid like torch.rand, and standard normals.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crnerf {

enum { RNG_STREAM_JITTER = 0, RNG_STREAM_U = 1, RNG_STREAM_NOISE_COARSE = 2, RNG_STREAM_NOISE_FINE = 3 };

struct Philox4 { uint32_t x[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

struct RayRng {
  uint32_t k0, k1;   // seed
  __host__ __device__ inline Philox4 bits(int stream, long ray, int sample) const {
    return philox4x32_10((uint32_t)sample, (uint32_t)stream, (uint32_t)((uint64_t)ray & 0xffffffffu), (uint32_t)((uint64_t)ray >> 32), k0, k1);
  }
  // U[0, 1) on the 2^-24 grid (what torch.rand produces for float32)
  __device__ inline float uniform(int stream, long ray, int sample) const { return (float)(bits(stream, ray, sample).x[0] >> 8) * 5.9604644775390625e-8f; }
  // standard normal, Box-Muller on two words of one Philox block
  __device__ inline float normal(int stream, long ray, int sample) const {
    const Philox4 b = bits(stream, ray, sample);
    const float u1 = (float)((b.x[0] >> 8) + 1u) * 5.9604644775390625e-8f;      // (0, 1]
    const float u2 = (float)(b.x[1] >> 8) * 5.9604644775390625e-8f;             // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  }
};

}  // namespace crnerf
