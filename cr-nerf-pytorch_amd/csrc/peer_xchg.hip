// Peer-window all-reduce for the decoder's two cross-ray reductions (64 channel sums, 1024 Gram sums; SURVEY 8e option B,
// linearStyleTransfer.py:29-34 / :59-65 are the two global means they feed) when the rays of one grid are sharded over
// the GPUs of a node: an alternative to the two RCCL all-reduces of parallel.decode_sharded, which for 260 B / 4 KiB
// messages are pure launch + protocol latency.
//
// Every rank owns one WINDOW in its own HBM, exported to the other processes by HIP IPC (xGMI peer access):
//     data [2 parities][8 source ranks][1024 floats]     flags [2][8] u32     status u32
// One single-workgroup kernel per reduction and rank:
//   push   my n floats into slot [parity][my rank] of EVERY rank's window (remote writes are posted: no round trip),
//          system-scope release, then flag [parity][my rank] = epoch in every window;
//   wait   until all flags [parity][*] of MY window carry this epoch (bounded spin, s_memrealtime deadline);
//   sum    the slots of my window in RANK ORDER into `data` -- the same order on every rank, so all ranks hold
//          bit-identical sums (a ring all-reduce does not promise that; the replicated decode that follows relies on it).
// Two parities suffice: a peer can start reduction e + 2 (same parity as e) only after it finished e + 1, which needs
// my flag of e + 1, which my stream raises after my kernel of e has read its slots.  Epochs only grow, so flags are
// never reset.  A rank whose peer never shows up writes NaN sums and records the missing rank in `status`
// (crnerf_peer_window_status) instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/crnerf.h"
#include "kernels.h"

namespace crnerf {

struct PeerWindow {
  float data[2][CRNERF_PEER_MAX_RANKS][CRNERF_PEER_MAX_FLOATS];
  uint32_t flags[2][CRNERF_PEER_MAX_RANKS];
  uint32_t status;   // 0, or 1 + the first rank whose flag did not arrive in time
  uint32_t pad[15];
};

struct PeerWindows {
  PeerWindow* w[CRNERF_PEER_MAX_RANKS];
};

__device__ __forceinline__ void store_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float load_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(256) void peer_allreduce_kernel(float* data, int n, PeerWindows win, int rank, int world, uint32_t epoch,
                                                               unsigned long long timeout_ticks) {
  const int par = epoch & 1;
  const int tid = threadIdx.x;
  __shared__ int failed;
  if (tid == 0) failed = 0;
  for (int peer = 0; peer < world; ++peer) {
    float* dst = win.w[peer]->data[par][rank];
    for (int i = tid; i < n; i += blockDim.x) store_sys(dst + i, data[i]);
  }
  __threadfence_system();
  __syncthreads();   // every lane's pushes are out before any flag goes up
  if (tid < world) __hip_atomic_store(&win.w[tid]->flags[par][rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  PeerWindow* mine = win.w[rank];
  if (tid < world) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__hip_atomic_load(&mine->flags[par][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
        atomicCAS(&mine->status, 0u, 1u + (uint32_t)tid);
        failed = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __threadfence_system();
  __syncthreads();
  const bool bad = failed != 0;
  for (int i = tid; i < n; i += blockDim.x) {
    float s = 0.0f;
    for (int r = 0; r < world; ++r) s += load_sys(&mine->data[par][r][i]);
    data[i] = bad ? __uint_as_float(0x7fc00000u) : s;   // quiet NaN by bit pattern (the build has -fno-honor-nans)
  }
}

}  // namespace crnerf

using namespace crnerf;

extern "C" {

size_t crnerf_peer_window_bytes(void) { return sizeof(PeerWindow); }

int crnerf_peer_window_create(void** window, void* handle_out) {
  if (!window || !handle_out) return set_error(CRNERF_ERR_NULL, "peer_window_create: NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) == CRNERF_PEER_HANDLE_BYTES, "handle size is part of the ABI");
  void* p = nullptr;
  // uncached (MTYPE UC) memory: writes arriving over xGMI and the owner's polling loads meet in HBM, not in an L2 that the
  // fabric does not snoop.  A plain (coarse-grained, L2-cached) hipMalloc is NOT a safe substitute across GPUs -- polling it can
  // read stale flags and time out spuriously -- so it is opt-in only: CRNERF_PEER_WINDOW_COARSE=1 (all ranks on ONE GPU, tests).
  const char* coarse = getenv("CRNERF_PEER_WINDOW_COARSE");
  hipError_t e;
  if (coarse && coarse[0] == '1') {
    e = hipMalloc(&p, sizeof(PeerWindow));
  } else {
    e = hipExtMallocWithFlags(&p, sizeof(PeerWindow), hipDeviceMallocUncached);
    if (e != hipSuccess) {
      char msg[240];
      snprintf(msg, sizeof(msg), "peer_window_create: hipExtMallocWithFlags(hipDeviceMallocUncached): %s -- no uncached window, use the RCCL "
               "all-reduce (exchange=None); CRNERF_PEER_WINDOW_COARSE=1 is for single-GPU tests only", hipGetErrorString(e));
      (void)hipGetLastError();
      return set_error(CRNERF_ERR_HIP, msg);
    }
  }
  if (e != hipSuccess) return set_error(CRNERF_ERR_HIP, "peer_window_create: allocation failed");
  if (hipMemset(p, 0, sizeof(PeerWindow)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return set_error(CRNERF_ERR_HIP, "peer_window_create: hipMemset failed");
  }
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    char msg[200];
    snprintf(msg, sizeof(msg), "peer_window_create: hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
    return set_error(CRNERF_ERR_HIP, msg);
  }
  memcpy(handle_out, &h, sizeof(h));
  *window = p;
  return 0;
}

int crnerf_peer_window_open(const void* handle, void** window) {
  if (!window || !handle) return set_error(CRNERF_ERR_NULL, "peer_window_open: NULL argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    char msg[200];
    snprintf(msg, sizeof(msg), "peer_window_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return set_error(CRNERF_ERR_HIP, msg);
  }
  *window = p;
  return 0;
}

int crnerf_peer_window_close(void* window) {
  if (!window) return 0;
  return hipIpcCloseMemHandle(window) == hipSuccess ? 0 : set_error(CRNERF_ERR_HIP, "peer_window_close: hipIpcCloseMemHandle failed");
}

int crnerf_peer_window_destroy(void* window) {
  if (!window) return 0;
  return hipFree(window) == hipSuccess ? 0 : set_error(CRNERF_ERR_HIP, "peer_window_destroy: hipFree failed");
}

int crnerf_peer_window_status(void* own_window, int* status) {
  if (!own_window || !status) return set_error(CRNERF_ERR_NULL, "peer_window_status: NULL argument");
  uint32_t s = 0;
  if (hipMemcpy(&s, &((PeerWindow*)own_window)->status, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess)
    return set_error(CRNERF_ERR_HIP, "peer_window_status: hipMemcpy failed");
  *status = (int)s;
  return 0;
}

int crnerf_peer_allreduce_f32(float* data, int n, void* const* windows, int rank, int world_size, uint32_t epoch, int64_t timeout_us,
                              void* stream) {
  if (!data || !windows) return set_error(CRNERF_ERR_NULL, "peer_allreduce: NULL argument");
  if (n < 0 || n > CRNERF_PEER_MAX_FLOATS) return set_error(CRNERF_ERR_SHAPE, "peer_allreduce: n must be in [0, 1024]");
  if (world_size < 1 || world_size > CRNERF_PEER_MAX_RANKS || rank < 0 || rank >= world_size)
    return set_error(CRNERF_ERR_SHAPE, "peer_allreduce: world_size must be in [1, 8] and rank inside it");
  if (epoch == 0) return set_error(CRNERF_ERR_SHAPE, "peer_allreduce: epochs start at 1 (0 is the cleared flag)");
  if (n == 0) return 0;
  PeerWindows w;
  for (int r = 0; r < CRNERF_PEER_MAX_RANKS; ++r) {
    w.w[r] = r < world_size ? (PeerWindow*)windows[r] : nullptr;
    if (r < world_size && !w.w[r]) return set_error(CRNERF_ERR_NULL, "peer_allreduce: a window pointer is NULL");
  }
  const unsigned long long ticks = (unsigned long long)(timeout_us > 0 ? timeout_us : 1) * 100ull;
  hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, data, n, w, rank, world_size, epoch, ticks);
  return check_launch("peer_allreduce_kernel");
}

}  // extern "C"
