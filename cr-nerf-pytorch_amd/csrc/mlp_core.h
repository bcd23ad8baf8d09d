// What every MLP core shares: vector typedefs, the LDS map (two consts blocks, the weight ring, scratch), the phase timer of
// -DCRNERF_TIMING builds, the LDS-DMA instruction as asm, the reference's softplus / sigmoid, the consts loader.
// (Rounds 1-4 kept the first core here as well -- one wave per SIMD on v_mfma_f32_32x32x2_f32, CRNERF_CORE=32; removed in round 5:
// the product cores are mlp_core16.h (fp32 MFMA), mlp_core_bf16p.h, mlp_core_x3.h, mlp_core_h2.h / mlp_core_h2t.h.)
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

namespace crnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) uint32_t lds_uint;
typedef const __attribute__((address_space(1))) char gbl_char;

constexpr int RING_SLOTS = 6;   // stages c-1 (being refilled), c (read), c+1 (read ahead), c+2 (certified), c+3, c+4 (in flight)
static_assert(RING_SLOTS >= 5, "ring too shallow for the read-ahead protocol");

// LDS map: byte offsets inside the single dynamic __shared__ array of every kernel using the core
constexpr int LDS_CONST0 = 0;
constexpr int LDS_CONST1 = CONST_BYTES;
constexpr int LDS_RING = 2 * CONST_BYTES;
constexpr int LDS_SCRATCH = LDS_RING + RING_SLOTS * STAGE_BYTES;  // 120,832

// Phase timer for tuning builds (-DCRNERF_TIMING): wave 0 of block 0 accumulates shader-clock cycles
// per phase into crnerf_timing[]; compiled out otherwise.
enum { T_PROLOGUE = 0, T_MMA, T_EPILOGUE, T_SIGMA, T_COMPOSITE, T_RAYLEVEL, T_TOTAL, T_X0, T_X1, T_X2, T_X3, T_X4, T_X5, T_X6, T_X7, T_REAL, T_COUNT };   // T_REAL: 100 MHz ticks from start() to flush()
#ifdef CRNERF_TIMING
static __device__ unsigned long long crnerf_timing[T_COUNT];   // one copy per translation unit
struct PhaseTimer {
  unsigned long long last, acc[T_COUNT], real0;
  bool on;
  __device__ __forceinline__ void start(bool on_) {
    on = on_;
    for (int i = 0; i < T_COUNT; ++i) acc[i] = 0;
    real0 = __builtin_amdgcn_s_memrealtime();
    last = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void tick(int phase) {
    __builtin_amdgcn_sched_barrier(0);   // the compiler may otherwise move MFMAs / VALU work across the clock read
    const unsigned long long t = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    acc[phase] += t - last;
    acc[T_TOTAL] += t - last;
    last = t;
  }
  __device__ __forceinline__ void flush() {
    acc[T_REAL] = __builtin_amdgcn_s_memrealtime() - real0;
    if (on)
      for (int i = 0; i < T_COUNT; ++i) crnerf_timing[i] = acc[i];
  }
};
#else
struct PhaseTimer {
  __device__ __forceinline__ void start(bool) {}
  __device__ __forceinline__ void tick(int) {}
  __device__ __forceinline__ void flush() {}
};
#endif

// One LDS-DMA piece: lanes copy 16 B each, global (base + voff + imm) -> LDS (lds_addr + imm + 16 * lane).  Written as
// asm on purpose: with the builtin, hipcc's waitcnt model stops counting LDS reads across an LDS-DMA instruction and
// the next use of ANY prefetched fragment becomes s_waitcnt lgkmcnt(0) -- the read-ahead ring is drained every fourth
// k-step.  (vmcnt for these loads is counted by hand in start()/advance() anyway.)  The asm rewrites M0 without telling
// the compiler (M0 is a reserved register, clang rejects it as a clobber): kernels that contain it must not index
// register arrays dynamically (s_set_gpr_idx / v_movrel keep their index in M0) -- checked by grepping the ISA.
//
// HAZARD (round 5, tools/isa_audit.py "vsgpr->vmem"): `base` is an SGPR pair the compiler owns.  Under SGPR pressure hipcc parks such pointers in VGPR
// lanes and restores them with v_readlane right in front of the statement -- and a VALU-written SGPR must not be read by a VMEM instruction for 5 wait
// states (CDNA3 ISA 4.5; hipcc pads its own VMEM, it cannot see the one in here).  mlp_forward_h2_kernel had 2 states in the round-4 build.  Two forms:
//   default          the base is used as it comes.  Correct as long as the audit finds no such reload in the unit's ISA -- tests/test_host.py runs it
//                    over every unit on every build, so a build that has one does not pass.
//   CRNERF_GLDS_SALU_COPY (per unit, build.py PER_FILE_FLAGS)   the base goes through an SALU copy inside the statement: an SALU read of a VALU-written
//                    SGPR is interlocked, and so is the VMEM read of the SALU's result.  Safe whatever the compiler does -- at a price: the
//                    interlock stalls the wave's issue, +7.5 % on the f32x3 renderer, +8.6 % on f32h2 (one wave per SIMD: nothing hides it;
//                    profiles/r5/glds_salu_copy_cost.txt).  Switched on for the units the audit flags (today: mlp_forward_h2.hip).
__device__ __forceinline__ void glds16(uint32_t lds_addr, const char* base, uint32_t voff, int imm) {
#ifdef CRNERF_GLDS_SALU_COPY
  uint64_t b;
  asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, %3\n\tglobal_load_lds_dwordx4 %2, %0 offset:%4"
               : "=&s"(b) : "s"(lds_addr), "v"(voff), "s"(base), "n"(imm) : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds_addr), "v"(voff), "s"(base), "n"(imm)
               : "memory");
#endif
}
// A further piece of the same stage: M0 as the last glds16 left it (nothing else in these kernels touches M0, tests/test_host.py), the
// instruction offset applies to both sides.
__device__ __forceinline__ void glds16_more(const char* base, uint32_t voff, int imm) {
#ifdef CRNERF_GLDS_SALU_COPY
  uint64_t b;
  asm volatile("s_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0 offset:%3" : "=&s"(b) : "v"(voff), "s"(base), "n"(imm) : "memory");
#else
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(base), "n"(imm) : "memory");
#endif
}

// 32x32 C/D layout (the split cores, mlp_core_x3.h / mlp_core_h2t.h): acc[t][4q+j] = bias[32t + 8q + 4h + j]
template <int NT>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NT], const lds_float* bias, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const __attribute__((address_space(3))) f32x4*)(bias + 32 * t + 8 * q + 4 * h);
      acc[t][4 * q + 0] = b[0];
      acc[t][4 * q + 1] = b[1];
      acc[t][4 * q + 2] = b[2];
      acc[t][4 * q + 3] = b[3];
    }
  }
}

template <int NT, int NDST>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NT], f32x16 (&act)[NDST], float floor_) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) act[t][r] = fmaxf(acc[t][r], floor_);
}

__device__ __forceinline__ float softplus_ref(float x) {
  // nn.Softplus(beta=1, threshold=20), models/nerf.py:146
  return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// Copy both models' consts blocks into LDS (all threads of a 256-thread workgroup).
__device__ __forceinline__ void load_consts(lds_char* lds, const char* packed0, const char* packed1) {
  const int tid = threadIdx.x;
  for (int i = tid; i < CONST_BYTES / 16; i += blockDim.x) {
    const f32x4 a = *(const f32x4*)(packed0 + i * 16);
    const f32x4 b = *(const f32x4*)(packed1 + i * 16);
    *(__attribute__((address_space(3))) f32x4*)(lds + LDS_CONST0 + i * 16) = a;
    *(__attribute__((address_space(3))) f32x4*)(lds + LDS_CONST1 + i * 16) = b;
  }
  __syncthreads();
}

}  // namespace crnerf
