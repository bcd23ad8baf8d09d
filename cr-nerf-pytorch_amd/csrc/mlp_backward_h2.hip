// Backward-data pass of NeRF_sigma on the h2 core's training form (mlp_core_h2t.h): mlp_backward_x3_kernel's job (mlp_backward_x3.hip) -- the layer deltas
//   delta_{l-1} = (W_l^T delta_l) . relu'(h_{l-1})          (in the reference: PyTorch autograd over the addmm nodes of models/nerf.py:157-182)
// -- with every product formed from TWO-piece fp16 splits on the fp16 matrix cores (fp32-accurate, three MFMAs per product; the transposed weights
// are split and scaled 2^8 at pack time, layout.h "fragHT").  Reads what the training forwards saved (relu bits, raw outputs), writes what the
// weight-gradient kernels read (deltas[10][P][256] fp32, d_rgb[P][64], d_sig[P]) -- same buffers, same layouts as the fp32 / x3 kernels.
// Range: a point's delta vector is rescaled by a power of two before every layer so that its largest entry sits in [2^7, 2^8) (mlp_core_h2t.h
// "operand scale"): gradients of any magnitude go through, and an overflow cannot happen (|w| < 255 from the pack check, 256 terms).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "mlp_core_h2t.h"
#include "mlp_train16.h"

namespace crnerf {

template <int NT>
__device__ __forceinline__ void zero_acc_h(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

// The operand scale of this lane's point and the factor that undoes it together with the weights' 2^8: sc = 2^(134 - e), e = the biased exponent
// of the point's largest |delta| (clamped to [32, 254]: below 2^-95 the entries keep sc = 2^102 and simply use less of fp16's range; a zero
// vector stays zero).  Both are exact powers of two.
// slot_max: this LANE's LDS word for the largest |delta| of the tensor (see the kernel's last lines; one word per lane and slot: sixty-four lanes
// raising ONE word serialise in the LDS and held up the other waves' weight reads -- the data gradient ran 19 % longer), or null
template <int NT>
__device__ __forceinline__ void point_scale_h(const f32x16 (&d)[NT], float& sc, float& inv, lds_uint* slot_max = nullptr) {
  float m = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(m, fabsf(d[t][r])), fabsf(d[t][r + 1]));
  m = fmaxf(m, __shfl_xor(m, 32));
  if (slot_max) { const uint32_t b = __float_as_uint(m); *slot_max = *slot_max > b ? *slot_max : b; }   // m >= 0: its bits order like its value
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  e = e < 32 ? 32 : (e > 254 ? 254 : e);
  sc = __uint_as_float((uint32_t)(261 - e) << 23);     // 2^(134 - e)
  inv = __uint_as_float((uint32_t)(e - 15) << 23);     // 2^(e - 142) = 2^-8 / sc
}

// delta = acc * inv . relu' from the two mask words of this lane (finish_delta_x's bit map, mlp_backward_x3.hip); MASK = false: the layer is linear
template <int NT, bool MASK>
__device__ __forceinline__ void finish_delta_h(const f32x16 (&acc)[NT], f32x16 (&dl)[8], float inv, unsigned long long m0, unsigned long long m1) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float d = acc[t][4 * q + j] * inv;
        if (MASK) {
          const int bit = 4 * (2 * t + (q >> 1)) + j;
          const unsigned long long w = (q & 1) ? m1 : m0;
          d = ((w >> bit) & 1ull) ? d : 0.0f;
        }
        dl[t][4 * q + j] = d;
      }
}

__global__ __launch_bounds__(256, 1) void mlp_backward_h2_kernel(const char* __restrict__ packedT, const float* __restrict__ out, const float* __restrict__ d_out,
                                                                 const float* __restrict__ acts, float* __restrict__ deltas, float* __restrict__ d_rgb,
                                                                 float* __restrict__ d_sig, long P, int iters, uint32_t* __restrict__ dmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 31, h = lane >> 5;
  load_consts(lds, packedT, packedT);   // the consts block carries the sigma-head weights (unscaled)
  const lds_float* C = (const lds_float*)(lds + LDS_CONST0);
  if (__float_as_uint(C[H2_FLAG_WORD]) != 0u) return;   // a weight is outside fp16's range: the f32x3 data gradient runs instead (launch_mlp_backward)
  // the largest |delta| per slot, collected per lane in the (unused) second consts block -- 11 slots x 256 lanes = its 11 KB exactly -- and raised
  // into dmax[] once at the end: the range words of the f16x2 weight gradients (mlp_train16.hip wgrad_h2_kernel).  Every delta row but slot 0
  // leaves right behind a point_scale_h of exactly its values, so the maximum costs one LDS read-max-write per lane and layer.  Slots 0..9: the
  // delta rows; slot 10: d_rgb.
  static_assert(WG_RANGE_USED * 256 * 4 <= CONST_BYTES, "range words: one per lane and slot in the second consts block");
  lds_uint* smax = dmax ? (lds_uint*)(lds + LDS_CONST1) + threadIdx.x : nullptr;
  if (dmax) {                                         // (load_consts, which filled the block, ends with a barrier)
#pragma unroll
    for (int sl = 0; sl < WG_RANGE_USED; ++sl) smax[sl * 256] = 0u;
  }
  WeightPipeX pipe;
  pipe.set_stream_frags(STREAMHT_FRAGS);
  pipe.start(lds, packedT + CONST_BYTES, packedT + CONST_BYTES, 1, 1, lane, wave);
  xu32x4 q[X_AHEAD];
  pipe.prime(q);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long tile = ((long)it * gridDim.x + blockIdx.x) * 4 + wave;
    const long n = tile * 32 + p;
    const bool valid = n < P;
    const long nn = valid ? n : 0;
    const float* orow = out + nn * OUT_DIM;
    const float* grow = d_out + nn * OUT_DIM;
    // the relu bits of this lane's features, all layers in one load batch per tile (words g = h and g = h + 2 of every slot but the linear one)
    unsigned long long b0[ACT_SLOTS], b1[ACT_SLOTS];
#pragma unroll
    for (int sl = 0; sl < ACT_SLOTS; ++sl) {
      b0[sl] = (valid && sl != 8) ? *mask_slot((float*)acts, P, sl, nn, h) : 0ull;
      b1[sl] = (valid && sl != 8) ? *mask_slot((float*)acts, P, sl, nn, h + 2) : 0ull;
    }
    const ActSaveX dsv{deltas, P, nn, valid, h};     // the split cores' row saver, aimed at the delta rows
    const uint32_t vo = dsv.offset();
    float sc, inv;

    f32x16 dl[8], acc[8];
    f32x16 drgb[2];                                   // static_rgb: sigmoid'            nerf.py:154,180
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float f = orow[32 * t + 8 * qq + 4 * h + j], gf = valid ? grow[32 * t + 8 * qq + 4 * h + j] : 0.0f;
          v[j] = gf * f * (1.0f - f);
          drgb[t][4 * qq + j] = v[j];
        }
        if (valid) *(f32x4*)(d_rgb + n * FEAT_DIM + 32 * t + 8 * qq + 4 * h) = v;
      }
    // static_sigma: softplus' = sigmoid(pre) = 1 - exp(-sigma)      nerf.py:146,172
    const float sg = orow[FEAT_DIM];
    const float dsp = valid ? grow[FEAT_DIM] * (1.0f - expf(-sg)) : 0.0f;
    if (valid && h == 0) d_sig[n] = dsp;

    {  // through static_rgb^T -> dir_encoding output (relu)
      f32x16 acc4[4];
      zero_acc_h<4>(acc4);
      point_scale_h<2>(drgb, sc, inv, smax ? smax + ACT_SLOTS * 256 : nullptr);   // slot 10: d_rgb
      mma_layer_h2t<4, FEAT_DIM / 16, 0>(pipe, drgb, drgb, acc4, q, sc);
      finish_delta_h<4, true>(acc4, dl, inv, b0[9], b1[9]);
    }
    zero_acc_h<8>(acc);                               // through dir_encoding^T[:, :256] -> xyz_encoding_final output (linear); delta_9 leaves on the way
    {
      f32x16 d4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) d4[t] = dl[t];
      point_scale_h<4>(d4, sc, inv, smax ? smax + 9 * 256 : nullptr);
    }
    mma_layer_h2t<8, 128 / 16, 0, true, false>(pipe, dl, dl, acc, q, sc, dsv.row(9), SaveRowX{}, vo);
    finish_delta_h<8, false>(acc, dl, inv, 0ull, 0ull);
    zero_acc_h<8>(acc);                               // through xyz_encoding_final^T, + sigma head -> h8 (relu)
    point_scale_h<8>(dl, sc, inv, smax ? smax + 8 * 256 : nullptr);
    mma_layer_h2t<8, KS_HID, 0, true, false>(pipe, dl, dl, acc, q, sc, dsv.row(8), SaveRowX{}, vo);
#pragma unroll
    for (int t = 0; t < 8; ++t)                       // (the sigma-head term joins AFTER the accumulator is scaled back: dsp need not fit the deltas' scale)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 w = *(const __attribute__((address_space(3))) f32x4*)(C + C_WSIG + 32 * t + 8 * qq + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int bit = 4 * (2 * t + (qq >> 1)) + j;
          const float d = fmaf(w[j], dsp, acc[t][4 * qq + j] * inv);
          dl[t][4 * qq + j] = ((((qq & 1) ? b1[7] : b0[7]) >> bit) & 1ull) ? d : 0.0f;
        }
      }
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {                    // through xyz_encoding_{l+1}^T -> h_l (relu); l+1 = 8..2; delta_l leaves on the way
      zero_acc_h<8>(acc);
      point_scale_h<8>(dl, sc, inv, smax ? smax + l * 256 : nullptr);
      mma_layer_h2t<8, KS_HID, 0, true, false>(pipe, dl, dl, acc, q, sc, dsv.row(l), SaveRowX{}, vo);
      unsigned long long m0 = b0[0], m1 = b1[0];      // words of slot l - 1 by selects: a dynamic register index would go through M0, which the
#pragma unroll                                        // LDS-DMA asm (glds16) rewrites behind the compiler's back
      for (int t = 1; t < 7; ++t) { m0 = (l - 1 == t) ? b0[t] : m0; m1 = (l - 1 == t) ? b1[t] : m1; }
      finish_delta_h<8, true>(acc, dl, inv, m0, m1);
    }
    {   // the first layer's deltas: nothing left to hide them behind
      if (smax) { float s0, i0; point_scale_h<8>(dl, s0, i0, smax); }   // (only the range word is used)
      const SaveRowX r0 = dsv.row(0);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const xu32x4 v = {__float_as_uint(dl[t][4 * qq]), __float_as_uint(dl[t][4 * qq + 1]), __float_as_uint(dl[t][4 * qq + 2]), __float_as_uint(dl[t][4 * qq + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, r0.rs, (int)(vo + 4u * (32 * t + 8 * qq)), 0, 0);
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (dmax) {
    __syncthreads();
    if (threadIdx.x < WG_RANGE_USED) {
      const lds_uint* col = (const lds_uint*)(lds + LDS_CONST1) + threadIdx.x * 256;
      uint32_t b = 0u;
      for (int t = 0; t < 256; ++t) b = col[t] > b ? col[t] : b;
      if (b != 0u) atomicMax(dmax + threadIdx.x, b);
    }
  }
}

int launch_mlp_dgrad_h2(const void* packedT_h2, const float* out, const float* d_out, const float* acts, float* deltas, float* d_rgb, float* d_sig, long P,
                        hipStream_t stream, uint32_t* dmax) {
  if (P <= 0) return 0;
  if ((unsigned long long)P * 1024ull >= (unsigned long long)SAVEX_OOB)
    return set_error(-2, "mlp_backward_h2: more than 3.9 M points per call (the delta rows are addressed with 32-bit offsets)");
  const long groups = (P + 127) / 128;
  const int cus = num_cus();
  const int grid = (int)(groups < cus ? groups : cus), iters = (int)((groups + grid - 1) / grid);
  if (int rc = ensure_dynamic_lds((const void*)mlp_backward_h2_kernel, LDS_SCRATCH_X, "mlp_backward_h2_kernel")) return rc;
  hipLaunchKernelGGL(mlp_backward_h2_kernel, dim3(grid), dim3(256), LDS_SCRATCH_X, stream, (const char*)packedT_h2, out, d_out, acts, deltas, d_rgb, d_sig, P, iters, dmax);
  return check_launch("mlp_backward_h2_kernel");
}

}  // namespace crnerf
