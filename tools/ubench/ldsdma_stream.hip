// ubench (GPU box): how fast can every CU of the chip pull the SAME weight stream L2 -> LDS with global_load_lds_dwordx4, as the split cores do?
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_stream ldsdma_stream.hip && ./ldsdma_stream
// One 256-thread workgroup per CU; each wave issues 1 KiB pieces (4 per 16 KiB stage and wave, as WeightPipeX) and keeps DEPTH stages in flight
// (counted vmcnt, no LDS reads, optional per-stage s_barrier).  PHASE: workgroup b starts its walk at stage (b * PHASE) % stages, so CUs do (0) or
// do not (> 0) ask the L2 for the same lines at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int FRAG = 1024, STAGE = 16 * FRAG;
template <int DEPTH, bool BAR>
__global__ __launch_bounds__(256, 1) void stream_kernel(const char* __restrict__ buf, int stages, int passes, int phase, int slots, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem + wave * 4096;
  const unsigned voff = lane * 16;
  int st = (int)((blockIdx.x * (long)phase) % stages), slot = 0;
  const long total = (long)stages * passes;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (long i = 0; i < total; ++i) {
    const char* src = buf + (long)st * STAGE + wave * 4096;
    const unsigned dst = lds0 + slot * STAGE;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(dst), "v"(voff), "s"(src) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
    st = st + 1 == stages ? 0 : st + 1;
    slot = slot + 1 == slots ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
template <int DEPTH, bool BAR>
void run(const char* buf, int stages, int phase, int grid, unsigned long long* dcyc) {
  const int passes = 40, slots = DEPTH + 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)stream_kernel<DEPTH, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, slots * STAGE);
  stream_kernel<DEPTH, BAR><<<grid, 256, slots * STAGE>>>(buf, stages, 2, phase, slots, dcyc);
  hipEventRecord(e0);
  stream_kernel<DEPTH, BAR><<<grid, 256, slots * STAGE>>>(buf, stages, passes, phase, slots, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * stages * passes * STAGE;
  printf("depth %d  barrier %d  phase %3d  stream %.2f MB  grid %d: %.3f ms  %.2f TB/s chip  %.1f GB/s per CU\n", DEPTH, (int)BAR, phase, stages * STAGE / 1e6, grid, ms,
         bytes / ms / 1e9, bytes / ms / 1e6 / grid);
}
int main() {
  const int maxstages = 1024;   // 16 MB
  char* buf; hipMalloc(&buf, (size_t)maxstages * STAGE); hipMemset(buf, 1, (size_t)maxstages * STAGE);
  unsigned long long* dcyc; hipMalloc(&dcyc, 8 * 1024);
  for (int stages : {151, 228, 1024})
    for (int phase : {0, 1, 37}) {
      run<2, false>(buf, stages, phase, 256, dcyc);
      run<4, false>(buf, stages, phase, 256, dcyc);
      run<4, true>(buf, stages, phase, 256, dcyc);
      run<6, false>(buf, stages, phase, 256, dcyc);
      run<8, false>(buf, stages, phase, 256, dcyc);
    }
  run<4, false>(buf, 151, 0, 32, dcyc);     // one CU per ... (32 workgroups: 4 per XCD)
  run<4, false>(buf, 151, 0, 8, dcyc);
  return 0;
}
