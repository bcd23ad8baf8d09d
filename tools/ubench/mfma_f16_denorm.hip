// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs (needed by the two-piece fp16 split: the second piece of a value below 0.25 is
// subnormal)?  A[i][k] = 2^-20 (subnormal in fp16), B[k][j] = 1024: C = 16 * 2^-10 = 2^-6 if honoured, 0 if flushed.  Also: the fp32 -> fp16
// pack conversion used for the split rounds to nearest even (not toward zero).
// build: hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1024.0f; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = c[0];
    const h2 p = {(_Float16)out[4], (_Float16)out[5]};   // 1 + 1.5 ulp16 -> RNE: 1 + 2 ulp, RTZ: 1 + 1 ulp;  6e-6 -> subnormal
    out[1] = (float)p[0];
    out[2] = (float)p[1];
  }
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  const float in[2] = {1.0f + 1.5f * 0.0009765625f, 6.0e-6f};
  hipMemcpy(d + 4, in, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("mfma f16 subnormal inputs: C = %.9g (honoured: %.9g, flushed: 0)\n", h[0], 16 * 9.5367431640625e-07 * 1024);
  printf("cvt f32->f16 of 1 + 1.5 ulp16: %.9g (RNE: %.9g, RTZ: %.9g)\n", h[1], 1 + 2 * 0.0009765625, 1 + 0.0009765625);
  printf("cvt f32->f16 of 6e-6: %.9g (kept as subnormal: ~5.96e-06, flushed: 0)\n", h[2]);
  return 0;
}
