// Ray generation on the device: pinhole directions, camera-to-world rotation,
// normalisation, near/far columns -> rays[H*W,8].  In the reference this is host-side torch code run
// per frame followed by an H2D copy (datasets/ray_utils.py:5-52, datasets/PhototourismDataset.py:12-25,
// eval.py:272-279); here one tiny kernel writes the renderer's input directly in HBM.
//   directions(i,j) = [(i - cx)/fx, -(j - cy)/fy, -1]          ray_utils.py:19-24 (no +0.5 pixel centre)
//   rays_d = normalize(directions @ c2w[:, :3].T), rays_o = c2w[:, 3]      ray_utils.py:44-48
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace crnerf {

struct Cam { float fx, fy, cx, cy; float c2w[12]; };

__global__ void ray_directions_kernel(float fx, float fy, float cx, float cy, int H, int W, float* __restrict__ dirs) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * W) return;
  const int i = (int)(idx % W), j = (int)(idx / W);
  dirs[idx * 3 + 0] = ((float)i - cx) / fx;
  dirs[idx * 3 + 1] = -((float)j - cy) / fy;
  dirs[idx * 3 + 2] = -1.0f;
}

__device__ __forceinline__ void rotate_normalize(const float* c2w, float dx, float dy, float dz, float (&d)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = fmaf(dz, c2w[4 * k + 2], fmaf(dy, c2w[4 * k + 1], dx * c2w[4 * k + 0]));
  const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  d[0] /= n; d[1] /= n; d[2] /= n;
}

__global__ void rays_from_directions_kernel(const float* __restrict__ dirs, Cam cam, long n, float* __restrict__ rays_o,
                                            float* __restrict__ rays_d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float d[3];
  rotate_normalize(cam.c2w, dirs[idx * 3 + 0], dirs[idx * 3 + 1], dirs[idx * 3 + 2], d);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rays_d[idx * 3 + k] = d[k];
    rays_o[idx * 3 + k] = cam.c2w[4 * k + 3];
  }
}

__global__ void generate_rays_kernel(Cam cam, int H, int W, float near, float far, float* __restrict__ rays) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * W) return;
  const int i = (int)(idx % W), j = (int)(idx / W);
  float d[3];
  rotate_normalize(cam.c2w, ((float)i - cam.cx) / cam.fx, -((float)j - cam.cy) / cam.fy, -1.0f, d);
  float* r = rays + idx * 8;
  r[0] = cam.c2w[3]; r[1] = cam.c2w[7]; r[2] = cam.c2w[11];
  r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
  r[6] = near; r[7] = far;
}

int launch_ray_directions(float fx, float fy, float cx, float cy, int H, int W, float* dirs, hipStream_t stream) {
  if (H <= 0 || W <= 0) return set_error(-2, "ray_directions: empty image");
  const long n = (long)H * W;
  hipLaunchKernelGGL(ray_directions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fx, fy, cx, cy, H, W, dirs);
  return check_launch("ray_directions_kernel");
}

int launch_rays_from_directions(const float* dirs, const float* c2w_host, long n, float* rays_o, float* rays_d, hipStream_t stream) {
  if (n <= 0) return 0;
  Cam cam{};
  for (int k = 0; k < 12; ++k) cam.c2w[k] = c2w_host[k];
  hipLaunchKernelGGL(rays_from_directions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dirs, cam, n, rays_o, rays_d);
  return check_launch("rays_from_directions_kernel");
}

int launch_generate_rays(const float* intr4_host, const float* c2w_host, int H, int W, float near, float far, float* rays, hipStream_t stream) {
  if (H <= 0 || W <= 0) return set_error(-2, "generate_rays: empty image");
  Cam cam{intr4_host[0], intr4_host[1], intr4_host[2], intr4_host[3], {}};
  for (int k = 0; k < 12; ++k) cam.c2w[k] = c2w_host[k];
  const long n = (long)H * W;
  hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cam, H, W, near, far, rays);
  return check_launch("generate_rays_kernel");
}

}  // namespace crnerf
