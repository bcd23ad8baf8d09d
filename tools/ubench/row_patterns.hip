// Micro-benchmark (GPU box): how fast can a wave stream 512-byte rows when every load / store instruction moves 16 B per lane and
// the lanes are laid out as the MFMA operand shapes dictate?
//   pattern 32x32 : lane (i = lane & 31, hh = lane >> 5) <-> row i, 16-byte piece hh of a 32-byte column slice  (v_mfma_*_32x32x16 operands)
//   pattern 16x64 : lane (i = lane & 15, q = lane >> 4)  <-> row i, 16-byte piece q of a 64-byte column slice  (v_mfma_*_16x16x32 operands)
//   pattern 4x256 : lane (i = lane >> 4, q = lane & 15)  <-> row i, 16-byte piece q of a 256-byte half row     (a plain coalesced copy)
// Each wave owns a block of rows and walks its columns; mode 0 = read only (sum to defeat DCE), 1 = write only, 2 = read + write.
// build: hipcc --offload-arch=gfx950 -O3 -o row_patterns row_patterns.hip ; run: ./row_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int ROWS, int PIECES>   // ROWS x PIECES = 64 lanes; a column slice is PIECES x 16 bytes
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long rows, int mode, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = (ROWS == 4) ? lane / PIECES : lane % ROWS, q = (ROWS == 4) ? lane % PIECES : lane / ROWS;
  constexpr int SLICES = 32 / PIECES;            // 16-byte pieces per 512-byte row = 32
  const long tiles = rows / ROWS;
  unsigned acc = 0;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < tiles; tile += (long)gridDim.x * 8) {
    const long row = tile * ROWS + i;
    uint4 v[SLICES];
    if (mode != 1) {
#pragma unroll
      for (int s = 0; s < SLICES; ++s) v[s] = in[row * 32 + s * PIECES + q];
    } else {
#pragma unroll
      for (int s = 0; s < SLICES; ++s) v[s] = make_uint4(lane, s, 0, 0);
    }
    if (mode != 0) {
#pragma unroll
      for (int s = 0; s < SLICES; ++s) out[row * 32 + s * PIECES + q] = v[s];
    } else {
#pragma unroll
      for (int s = 0; s < SLICES; ++s) acc += v[s].x ^ v[s].w;
    }
  }
  if (mode == 0 && acc == 0x12345678u) *sink = acc;
}

template <int ROWS, int PIECES>
static void run(const char* name, const uint4* in, uint4* out, long rows, unsigned* sink) {
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<ROWS, PIECES>), dim3(256 * 2), dim3(512), 0, 0, in, out, rows, mode, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((stream_kernel<ROWS, PIECES>), dim3(256 * 2), dim3(512), 0, 0, in, out, rows, mode, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)rows * 512 * (mode == 2 ? 2 : 1) * 5;
    printf("%-8s %-12s %7.2f TB/s\n", name, mode == 0 ? "read" : (mode == 1 ? "write" : "read+write"), bytes / (ms * 1e-3) / 1e12);
  }
}

int main() {
  const long rows = 1L << 22;   // 2 GiB per buffer
  uint4 *in, *out;
  unsigned* sink;
  hipMalloc(&in, rows * 512); hipMalloc(&out, rows * 512); hipMalloc(&sink, 4);
  hipMemset(in, 1, rows * 512);
  run<32, 2>("32x32B", in, out, rows, sink);
  run<16, 4>("16x64B", in, out, rows, sink);
  run<4, 16>("4x256B", in, out, rows, sink);
  return 0;
}
