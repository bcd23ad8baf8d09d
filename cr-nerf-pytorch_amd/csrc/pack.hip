// Weight pre-pack (24 nn.Linear tensors of one NeRF_sigma -> consts + MFMA fragment stream) and the
// stand-alone positional-embedding kernel.
// Reference layouts: NeRF_sigma.__init__ models/nerf.py:137-154; PosEmbedding.forward models/nerf.py:17-30.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "sincos_pow2.h"
#include "layout.h"

namespace crnerf {

__global__ void pack_stream_kernel(MlpTensors t, float* __restrict__ stream) {   // fp32 "v16" fragment order (layout.h): 16-row tiles
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAM_FRAGS * FRAG_FLOATS) return;
  const int frag = (int)(idx / FRAG_FLOATS);
  const int lane = (int)(idx % FRAG_FLOATS) / 4, j = (int)(idx % 4);
  const int i = lane & 15, kk = lane >> 4;

  const float* W;
  int in_dim, nt, phi, kind;  // kind: 0 hidden, 1 L1, 2 L5 (skip), 3 dir, 4 rgb
  if (frag < OFF_L2) { W = t.w[0]; in_dim = XYZ_DIM; nt = 8; phi = frag - OFF_L1; kind = 1; }
  else if (frag < OFF_L5) { const int l = (frag - OFF_L2) / FR_HID; W = t.w[1 + l]; in_dim = W_HIDDEN; nt = 8; phi = (frag - OFF_L2) % FR_HID; kind = 0; }
  else if (frag < OFF_L6) { W = t.w[4]; in_dim = XYZ_DIM + W_HIDDEN; nt = 8; phi = frag - OFF_L5; kind = 2; }
  else if (frag < OFF_FIN) { const int l = (frag - OFF_L6) / FR_HID; W = t.w[5 + l]; in_dim = W_HIDDEN; nt = 8; phi = (frag - OFF_L6) % FR_HID; kind = 0; }
  else if (frag < OFF_DIR) { W = t.w_final; in_dim = W_HIDDEN; nt = 8; phi = frag - OFF_FIN; kind = 0; }
  else if (frag < OFF_RGB) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; nt = 4; phi = frag - OFF_DIR; kind = 3; }
  else { W = t.w_rgb; in_dim = 128; nt = 2; phi = frag - OFF_RGB; kind = 4; }

  nt *= 2;                                              // 16-row tiles
  const int v = phi / nt, tile = phi % nt;
  const int k = 16 * v + 4 * kk + j;
  const int row = 16 * tile + i;
  int col;
  switch (kind) {
    case 1: col = posenc_slot_to_col16(k, XYZ_FREQS); break;
    case 2: col = k < XYZ_PAD ? posenc_slot_to_col16(k, XYZ_FREQS) : XYZ_DIM + (k - XYZ_PAD); break;  // nerf.py:169 cat([xyz, h])
    case 3: {
      const int dc = k < W_HIDDEN ? 0 : posenc_slot_to_col16(k - W_HIDDEN, DIR_FREQS);
      col = k < W_HIDDEN ? k : (dc < 0 ? -1 : W_HIDDEN + dc);
      break;
    }
    default: col = k; break;
  }
  stream[idx] = col >= 0 ? W[(long)row * in_dim + col] : 0.0f;
}

// bias_scale: factor on the biases of the ten matrix-core layers (h2 packs: H2_WSCALE, the factor their weights carry -- the accumulator is then
// initialised with a plain LDS read); the sigma head (fp32 VALU) is never scaled
__global__ void pack_consts_kernel(MlpTensors t, float* __restrict__ c, float bias_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CONST_BYTES / 4) return;
  float v = 0.0f;
  if (i < C_BFIN) v = t.b[i / W_HIDDEN][i % W_HIDDEN] * bias_scale;
  else if (i < C_WSIG) v = t.b_final[i - C_BFIN] * bias_scale;
  else if (i < C_BSIG) v = t.w_sigma[i - C_WSIG];
  else if (i == C_BSIG) v = t.b_sigma[0];
  else if (i < C_BDIR) v = 0.0f;
  else if (i < C_BRGB) v = t.b_dir[i - C_BDIR] * bias_scale;
  else if (i < CONST_FLOATS) v = t.b_rgb[i - C_BRGB] * bias_scale;
  c[i] = v;
}

// transposed stream (layout.h, "WT"); consts block = the forward one (sigma weights are read from it)
__global__ void pack_streamT_kernel(MlpTensors t, float* __restrict__ stream) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAMT_FRAGS * FRAG_FLOATS) return;
  const int frag = (int)(idx / FRAG_FLOATS);
  const int lane = (int)(idx % FRAG_FLOATS) / 4, r = (int)(idx % 4);
  const int i = lane & 15, kq = lane >> 4;
  const float* W;
  int in_dim, in_off, nt, phi;
  if (frag < OFFT_DIR) { W = t.w_rgb; in_dim = 128; in_off = 0; nt = 8; phi = frag - OFFT_RGB; }
  else if (frag < OFFT_FIN) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; in_off = 0; nt = 16; phi = frag - OFFT_DIR; }
  else if (frag < OFFT_L8) { W = t.w_final; in_dim = W_HIDDEN; in_off = 0; nt = 16; phi = frag - OFFT_FIN; }
  else {
    const int l = 7 - (frag - OFFT_L8) / FT_HID;       // 7 = xyz_encoding_8 ... 1 = xyz_encoding_2
    W = t.w[l]; nt = 16; phi = (frag - OFFT_L8) % FT_HID;
    in_dim = (l == 4) ? XYZ_DIM + W_HIDDEN : W_HIDDEN;
    in_off = (l == 4) ? XYZ_DIM : 0;                   // nerf.py:169 cat([xyz, h]): h starts at column 93
  }
  const int u = phi / nt, T = phi % nt;
  const int out = 16 * u + 4 * kq + r, in = in_off + 16 * T + i;
  stream[idx] = W[(long)out * in_dim + in];
}

int launch_pack_mlpT(const MlpTensors& t, void* packed, hipStream_t stream) {
  float* consts = (float*)packed;
  float* wstream = (float*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, 1.0f);
  const long n = (long)STREAMT_FRAGS * FRAG_FLOATS;
  hipLaunchKernelGGL(pack_streamT_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream);
  return check_launch("pack_mlpT");
}

// bf16 stream (layout.h "fragB"): one thread per bf16 element, round-to-nearest-even like v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ void pack_stream_bf16_kernel(MlpTensors t, unsigned short* __restrict__ stream) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAMB_FRAGS * 512) return;
  const int frag = (int)(idx / 512);
  const int lane = (int)(idx % 512) / 8, e = (int)(idx % 8);
  const int i = lane & 31, hh = lane >> 5;
  if (frag >= STREAMB_USED) { stream[idx] = 0; return; }

  const float* W;
  int in_dim, ks, phi, kind;  // kind: 0 hidden, 1 L1, 2 L5 (skip), 3 dir; ks = k-steps per tile
  if (frag < OFFB_L2) { W = t.w[0]; in_dim = XYZ_DIM; ks = KS_XYZ; phi = frag - OFFB_L1; kind = 1; }
  else if (frag < OFFB_L5) { const int l = (frag - OFFB_L2) / FB_HID; W = t.w[1 + l]; in_dim = W_HIDDEN; ks = KS_HID; phi = (frag - OFFB_L2) % FB_HID; kind = 0; }
  else if (frag < OFFB_L6) { W = t.w[4]; in_dim = XYZ_DIM + W_HIDDEN; ks = KS_XYZ + KS_HID; phi = frag - OFFB_L5; kind = 2; }
  else if (frag < OFFB_FIN) { const int l = (frag - OFFB_L6) / FB_HID; W = t.w[5 + l]; in_dim = W_HIDDEN; ks = KS_HID; phi = (frag - OFFB_L6) % FB_HID; kind = 0; }
  else if (frag < OFFB_DIR) { W = t.w_final; in_dim = W_HIDDEN; ks = KS_HID; phi = frag - OFFB_FIN; kind = 0; }
  else if (frag < OFFB_RGB) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; ks = KS_HID + KS_DIR; phi = frag - OFFB_DIR; kind = 3; }
  else { W = t.w_rgb; in_dim = 128; ks = KS_HALF; phi = frag - OFFB_RGB; kind = 0; }

  const int T = phi / ks, s = phi % ks;
  const int row = 32 * T + i;
  // hidden-activation k-step sh: the feature lane-half hh holds in element e
  #define CRNERF_HID_FEATURE(sh) (32 * ((sh) >> 1) + 16 * ((sh) & 1) + 4 * hh + (e & 3) + 8 * (e >> 2))
  int col;
  switch (kind) {
    case 1: col = posenc_slot_to_col_b(16 * s + 8 * hh + e, XYZ_FREQS); break;
    case 2: col = s < KS_XYZ ? posenc_slot_to_col_b(16 * s + 8 * hh + e, XYZ_FREQS)
                             : XYZ_DIM + CRNERF_HID_FEATURE(s - KS_XYZ); break;   // nerf.py:169 cat([xyz, h])
    case 3: {
      if (s < KS_HID) col = CRNERF_HID_FEATURE(s);
      else {
        const int dc = posenc_slot_to_col_b(16 * (s - KS_HID) + 8 * hh + e, DIR_FREQS);
        col = dc < 0 ? -1 : W_HIDDEN + dc;                                          // nerf.py:177 cat([final, dir])
      }
      break;
    }
    default: col = CRNERF_HID_FEATURE(s); break;
  }
  #undef CRNERF_HID_FEATURE
  stream[idx] = col >= 0 ? f32_to_bf16_rne(W[(long)row * in_dim + col]) : (unsigned short)0;
}

int launch_pack_mlp_bf16(const MlpTensors& t, void* packed, hipStream_t stream) {
  float* consts = (float*)packed;
  unsigned short* wstream = (unsigned short*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, 1.0f);
  const long n = (long)STREAMB_FRAGS * 512;
  hipLaunchKernelGGL(pack_stream_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream);
  return check_launch("pack_mlp_bf16");
}

// x3 stream (layout.h "fragX"): one thread per bf16 element; the three pieces of a weight by repeated round-to-nearest-even
__global__ void pack_stream_x3_kernel(MlpTensors t, unsigned short* __restrict__ stream) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAMX_FRAGS * 512) return;
  const int frag = (int)(idx / 512);
  const int lane = (int)(idx % 512) / 8, e = (int)(idx % 8);
  const int i = lane & 31, hh = lane >> 5;
  const float* W;
  int in_dim, nt, phi, kind;   // kind: 0 hidden, 1 L1, 2 L5 (skip), 3 dir
  if (frag < OFFX_L2) { W = t.w[0]; in_dim = XYZ_DIM; nt = 8; phi = frag - OFFX_L1; kind = 1; }
  else if (frag < OFFX_L5) { const int l = (frag - OFFX_L2) / FX_HID; W = t.w[1 + l]; in_dim = W_HIDDEN; nt = 8; phi = (frag - OFFX_L2) % FX_HID; kind = 0; }
  else if (frag < OFFX_L6) { W = t.w[4]; in_dim = XYZ_DIM + W_HIDDEN; nt = 8; phi = frag - OFFX_L5; kind = 2; }
  else if (frag < OFFX_FIN) { const int l = (frag - OFFX_L6) / FX_HID; W = t.w[5 + l]; in_dim = W_HIDDEN; nt = 8; phi = (frag - OFFX_L6) % FX_HID; kind = 0; }
  else if (frag < OFFX_DIR) { W = t.w_final; in_dim = W_HIDDEN; nt = 8; phi = frag - OFFX_FIN; kind = 0; }
  else if (frag < OFFX_RGB) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; nt = 4; phi = frag - OFFX_DIR; kind = 3; }
  else { W = t.w_rgb; in_dim = 128; nt = 2; phi = frag - OFFX_RGB; kind = 0; }
  if (kind == 3 && phi >= FX_DIR_USED) { stream[idx] = 0; return; }      // stage padding behind dir_encoding
  const int piece = phi % 3, st = phi / 3;
  const int s = st / nt, T = st % nt;
  const int row = 32 * T + i;
  auto kidx = [&](int sl) { return 16 * sl + 8 * (e >> 2) + 4 * hh + (e & 3); };
  int col;
  switch (kind) {
    case 1: col = posenc_slot_to_col(kidx(s), XYZ_FREQS); break;
    case 2: col = s < KS_XYZ ? posenc_slot_to_col(kidx(s), XYZ_FREQS) : XYZ_DIM + kidx(s - KS_XYZ); break;      // nerf.py:169 cat([xyz, h])
    case 3: {
      if (s < KS_HID) col = kidx(s);
      else { const int dc = posenc_slot_to_col(kidx(s - KS_HID), DIR_FREQS); col = dc < 0 ? -1 : W_HIDDEN + dc; }   // nerf.py:177 cat([final, dir])
      break;
    }
    default: col = kidx(s); break;
  }
  if (col < 0) { stream[idx] = 0; return; }
  const float w = W[(long)row * in_dim + col];
  const unsigned short p1 = f32_to_bf16_rne(w);
  const float r1 = w - __uint_as_float((unsigned)p1 << 16);
  const unsigned short p2 = f32_to_bf16_rne(r1);
  const float r2 = r1 - __uint_as_float((unsigned)p2 << 16);
  const unsigned short p3 = f32_to_bf16_rne(r2);
  stream[idx] = piece == 0 ? p1 : (piece == 1 ? p2 : p3);
}

// h2 stream (layout.h "fragH"): one thread per fp16 element; weights scaled by H2_WSCALE, two pieces by repeated round-to-nearest-even.
// A weight whose scaled value leaves the fp16 range becomes inf: the outputs are then inf / nan, not silently wrong.
// range_flag: one word of the pack's own consts block (behind CONST_FLOATS, zeroed by pack_consts_kernel; no kernel reads it), set when a scaled
// weight is not a finite fp16 number -- per pack, so concurrent packs on different streams / threads do not share state
__global__ void pack_stream_h2_kernel(MlpTensors t, unsigned short* __restrict__ stream, int* __restrict__ range_flag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAMH_FRAGS * 512) return;
  const int frag = (int)(idx / 512);
  const int lane = (int)(idx % 512) / 8, e = (int)(idx % 8);
  const int i = lane & 31, hh = lane >> 5;
  const float* W;
  int in_dim, nt, phi, kind;   // kind: 0 hidden, 1 L1, 2 L5 (skip), 3 dir
  if (frag < OFFH_L2) { W = t.w[0]; in_dim = XYZ_DIM; nt = 8; phi = frag - OFFH_L1; kind = 1; }
  else if (frag < OFFH_L5) { const int l = (frag - OFFH_L2) / FH_HID; W = t.w[1 + l]; in_dim = W_HIDDEN; nt = 8; phi = (frag - OFFH_L2) % FH_HID; kind = 0; }
  else if (frag < OFFH_L6) { W = t.w[4]; in_dim = XYZ_DIM + W_HIDDEN; nt = 8; phi = frag - OFFH_L5; kind = 2; }
  else if (frag < OFFH_FIN) { const int l = (frag - OFFH_L6) / FH_HID; W = t.w[5 + l]; in_dim = W_HIDDEN; nt = 8; phi = (frag - OFFH_L6) % FH_HID; kind = 0; }
  else if (frag < OFFH_DIR) { W = t.w_final; in_dim = W_HIDDEN; nt = 8; phi = frag - OFFH_FIN; kind = 0; }
  else if (frag < OFFH_RGB) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; nt = 4; phi = frag - OFFH_DIR; kind = 3; }
  else { W = t.w_rgb; in_dim = 128; nt = 2; phi = frag - OFFH_RGB; kind = 0; }
  const int piece = phi % 2, st = phi / 2;
  const int s = st / nt, T = st % nt;
  const int row = 32 * T + i;
  auto kidx = [&](int sl) { return 16 * sl + 8 * (e >> 2) + 4 * hh + (e & 3); };
  int col;
  switch (kind) {
    case 1: col = posenc_slot_to_col(kidx(s), XYZ_FREQS); break;
    case 2: col = s < KS_XYZ ? posenc_slot_to_col(kidx(s), XYZ_FREQS) : XYZ_DIM + kidx(s - KS_XYZ); break;      // nerf.py:169 cat([xyz, h])
    case 3: {
      if (s < KS_HID) col = kidx(s);
      else { const int dc = posenc_slot_to_col(kidx(s - KS_HID), DIR_FREQS); col = dc < 0 ? -1 : W_HIDDEN + dc; }   // nerf.py:177 cat([final, dir])
      break;
    }
    default: col = kidx(s); break;
  }
  if (col < 0) { stream[idx] = 0; return; }
  const float w = W[(long)row * in_dim + col] * H2_WSCALE;
  if ((__float_as_uint(w) & 0x7fffffffu) >= 0x477fe000u && piece == 0) atomicOr(range_flag, 1);   // |w| >= 65,504, inf or NaN (bit test: this unit is built with -fno-honor-nans)
  const _Float16 p1 = (_Float16)w;                       // round to nearest even; subnormals kept
  const _Float16 p2 = (_Float16)(w - (float)p1);
  stream[idx] = __builtin_bit_cast(unsigned short, piece == 0 ? p1 : p2);
}

int launch_pack_mlp_h2(const MlpTensors& t, void* packed, hipStream_t stream, bool check) {
  float* consts = (float*)packed;
  unsigned short* wstream = (unsigned short*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, H2_WSCALE);
  const long n = (long)STREAMH_FRAGS * 512;
  // range check: the ONE place this library waits for the stream -- packing happens once per set of weights, and a weight beyond fp16's range
  // would otherwise surface as a finite, wrong output (inf - inf = NaN in a hidden layer is clamped to 0 by the relu)
  int flag = 0;
  int* dflag = (int*)consts + H2_FLAG_WORD;
  hipLaunchKernelGGL(pack_stream_h2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream, dflag);
  if (int rc = check_launch("pack_mlp_h2")) return rc;
  if (!check) return 0;   // crnerf_pack_mlp_weights_h2_async: the flag stays on the device (crnerf_pack_h2_status reads it; the h2 kernels honour it)
  if (hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
    return set_error(-10, "pack_mlp_h2: reading the range flag failed");
  if (flag) return set_error(-4, "pack_mlp_weights_h2: a weight is outside the h2 core's range (|w| < 255, finite); use the f32x3 or fp32 entry points");
  return 0;
}

int launch_pack_mlp_x3(const MlpTensors& t, void* packed, hipStream_t stream) {
  float* consts = (float*)packed;
  unsigned short* wstream = (unsigned short*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, 1.0f);
  const long n = (long)STREAMX_FRAGS * 512;
  hipLaunchKernelGGL(pack_stream_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream);
  return check_launch("pack_mlp_x3");
}

// transposed x3 stream (layout.h "fragXT")
__global__ void pack_stream_x3t_kernel(MlpTensors t, unsigned short* __restrict__ stream) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAMXT_FRAGS * 512) return;
  const int frag = (int)(idx / 512);
  const int lane = (int)(idx % 512) / 8, e = (int)(idx % 8);
  const int i = lane & 31, hh = lane >> 5;
  const float* W;
  int in_dim, col0, nt, phi;   // W[k][col0 + r]: k = the layer's output feature (contraction), r = its hidden input (tile row)
  if (frag < OFFXT_DIR) { W = t.w_rgb; in_dim = 128; col0 = 0; nt = 4; phi = frag - OFFXT_RGB; }
  else if (frag < OFFXT_FIN) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; col0 = 0; nt = 8; phi = frag - OFFXT_DIR; }
  else if (frag < OFFXT_L8) { W = t.w_final; in_dim = W_HIDDEN; col0 = 0; nt = 8; phi = frag - OFFXT_FIN; }
  else {
    const int j = (frag - OFFXT_L8) / FXT_HID;          // 0..6 <-> xyz_encoding_{8 - j}
    const int l = 7 - j;                                 // index into t.w
    W = t.w[l]; in_dim = l == 4 ? XYZ_DIM + W_HIDDEN : W_HIDDEN; col0 = l == 4 ? XYZ_DIM : 0; nt = 8; phi = (frag - OFFXT_L8) % FXT_HID;
  }
  const int piece = phi % 3, st = phi / 3;
  const int s = st / nt, T = st % nt;
  const int k = 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3);
  const int r = 32 * T + i;
  const float w = W[(long)k * in_dim + col0 + r];
  const unsigned short p1 = f32_to_bf16_rne(w);
  const float r1 = w - __uint_as_float((unsigned)p1 << 16);
  const unsigned short p2 = f32_to_bf16_rne(r1);
  const float r2 = r1 - __uint_as_float((unsigned)p2 << 16);
  const unsigned short p3 = f32_to_bf16_rne(r2);
  stream[idx] = piece == 0 ? p1 : (piece == 1 ? p2 : p3);
}

int launch_pack_mlp_x3t(const MlpTensors& t, void* packed, hipStream_t stream) {
  float* consts = (float*)packed;
  unsigned short* wstream = (unsigned short*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, 1.0f);
  const long n = (long)STREAMXT_FRAGS * 512;
  hipLaunchKernelGGL(pack_stream_x3t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream);
  return check_launch("pack_mlp_x3t");
}

// transposed h2 stream (layout.h "fragHT"): fragXT's element map, two fp16 pieces of 2^8 w; the range flag as in the forward pack (never read back
// here: mlp_backward_h2_kernel leaves at once when it is set and the f32x3 data gradient runs in its place, mlp_train16.hip launch_mlp_backward).
__global__ void pack_stream_h2t_kernel(MlpTensors t, unsigned short* __restrict__ stream, int* __restrict__ range_flag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)STREAMHT_FRAGS * 512) return;
  const int frag = (int)(idx / 512);
  const int lane = (int)(idx % 512) / 8, e = (int)(idx % 8);
  const int i = lane & 31, hh = lane >> 5;
  const float* W;
  int in_dim, col0, nt, phi;   // W[k][col0 + r]: k = the layer's output feature (contraction), r = its hidden input (tile row)
  if (frag < OFFHT_DIR) { W = t.w_rgb; in_dim = 128; col0 = 0; nt = 4; phi = frag - OFFHT_RGB; }
  else if (frag < OFFHT_FIN) { W = t.w_dir; in_dim = W_HIDDEN + DIR_DIM; col0 = 0; nt = 8; phi = frag - OFFHT_DIR; }
  else if (frag < OFFHT_L8) { W = t.w_final; in_dim = W_HIDDEN; col0 = 0; nt = 8; phi = frag - OFFHT_FIN; }
  else {
    const int j = (frag - OFFHT_L8) / FHT_HID;          // 0..6 <-> xyz_encoding_{8 - j}
    const int l = 7 - j;                                 // index into t.w
    W = t.w[l]; in_dim = l == 4 ? XYZ_DIM + W_HIDDEN : W_HIDDEN; col0 = l == 4 ? XYZ_DIM : 0; nt = 8; phi = (frag - OFFHT_L8) % FHT_HID;
  }
  const int piece = phi % 2, st = phi / 2;
  const int s = st / nt, T = st % nt;
  const int k = 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3);
  const int r = 32 * T + i;
  const float w = W[(long)k * in_dim + col0 + r] * H2_WSCALE;
  if ((__float_as_uint(w) & 0x7fffffffu) >= 0x477fe000u && piece == 0) atomicOr(range_flag, 1);   // as pack_stream_h2_kernel
  const _Float16 p1 = (_Float16)w;
  const _Float16 p2 = (_Float16)(w - (float)p1);
  stream[idx] = __builtin_bit_cast(unsigned short, piece == 0 ? p1 : p2);
}

int pack_h2_status(const void* packed, hipStream_t stream) {
  int flag = 0;
  if (hipMemcpyAsync(&flag, (const int*)packed + H2_FLAG_WORD, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
    return set_error(-10, "pack_h2_status: reading the range flag failed");
  if (flag) return set_error(-4, "pack_h2_status: a weight of this pack is outside the h2 core's range (|w| < 255, finite)");
  return 0;
}

int launch_pack_mlp_h2t(const MlpTensors& t, void* packed, hipStream_t stream) {
  float* consts = (float*)packed;
  unsigned short* wstream = (unsigned short*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, H2_WSCALE);
  const long n = (long)STREAMHT_FRAGS * 512;
  hipLaunchKernelGGL(pack_stream_h2t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream, (int*)consts + H2_FLAG_WORD);
  return check_launch("pack_mlp_h2t");
}

int launch_pack_mlp(const MlpTensors& t, void* packed, hipStream_t stream) {
  float* consts = (float*)packed;
  float* wstream = (float*)((char*)packed + CONST_BYTES);
  hipLaunchKernelGGL(pack_consts_kernel, dim3((CONST_BYTES / 4 + 255) / 256), dim3(256), 0, stream, t, consts, 1.0f);
  const long n = (long)STREAM_FRAGS * FRAG_FLOATS;
  hipLaunchKernelGGL(pack_stream_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t, wstream);
  return check_launch("pack_mlp");
}

// sin / cos of 2^f v: the fused kernels' routine (sincos_pow2.h) wherever it is valid, so that the stand-alone entry points and the fused
// renderers produce the SAME embeddings; ocml's sincosf beyond its range (crnerf_posenc_f32 accepts up to 30 frequencies)
__device__ __forceinline__ void sincos_freq(float v, int f, float& s, float& c) {
  if (f <= 16 && fabsf(v) < 64.0f) sincos_rev2pi(to_rev2pi(v), f, s, c);
  else sincosf(ldexpf(v, f), &s, &c);
}

// x[n,3] -> out[n, 6F+3]; one thread per (row, argument) pair; accurate sin / cos (see posenc.h).
__global__ void posenc_kernel(const float* __restrict__ x, float* __restrict__ out, long n, int F) {
  // one thread per (row, group): group 0 = the identity columns, group 1 + f = sin and cos of the three coordinates at frequency 2^f
  // = 24 contiguous bytes (a thread per argument wrote two 4-byte pieces 12 bytes apart)
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = F + 1;
  if (idx >= n * per_row) return;
  const long row = idx / per_row;
  const int a = (int)(idx % per_row);
  const int D = 6 * F + 3;
  const float* xr = x + row * 3;
  float* o = out + row * D;
  const float v0 = xr[0], v1 = xr[1], v2 = xr[2];
  if (a == 0) {
    o[0] = v0; o[1] = v1; o[2] = v2;
  } else {
    const int f = a - 1;
    float s0, c0, s1, c1, s2, c2;
    sincos_freq(v0, f, s0, c0);
    sincos_freq(v1, f, s1, c1);
    sincos_freq(v2, f, s2, c2);
    float* of = o + 3 + 6 * f;
    of[0] = s0; of[1] = s1; of[2] = s2; of[3] = c0; of[4] = c1; of[5] = c2;
  }
}

// x[r*N + n] = cat(PosEmbedding_xyz(o_r + d_r * z[r][n]), dir_emb[r]) -- rendering.py:100-114 (xyz_ = rays_o + rays_d * z_vals, the
// embedding, the repeat of dir_embedded, the cat) in one pass that writes the 480-byte row once; the torch composition of the
// same arithmetic moves ~1.5 KB per point (points, embedding, expanded directions, cat).  One thread per (point, output group):
// 15 frequency groups (sin and cos of the three coordinates = 24 contiguous bytes), 1 group for the identity columns, 7 groups of
// four direction columns (27 padded to 28); neighbouring threads write neighbouring bytes of the row.
__global__ void embed_points_kernel(const float* __restrict__ rays, const float* __restrict__ z, const float* __restrict__ dir_emb,
                                    float* __restrict__ x, long R, int N) {
  constexpr int G = 1 + XYZ_FREQS + 7;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * N * G) return;
  const long pt = idx / G;
  const int a = (int)(idx - pt * G);
  const long r = pt / N;
  float* o = x + pt * IN_DIM;
  if (a > XYZ_FREQS) {
    const int c0 = 4 * (a - XYZ_FREQS - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < DIR_DIM) o[XYZ_DIM + c0 + e] = dir_emb[r * DIR_DIM + c0 + e];
    return;
  }
  const float* ray = rays + r * 8;
  const float zz = z[pt];
  float v[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) v[d] = ray[d] + ray[3 + d] * zz;       // separate mul and add (-ffp-contract=off), as torch evaluates it
  if (a == 0) {
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  } else {
    const int f = a - 1;
    float sn[3], cs[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) sincos_freq(v[d], f, sn[d], cs[d]);
    float* of = o + 3 + 6 * f;
    of[0] = sn[0]; of[1] = sn[1]; of[2] = sn[2]; of[3] = cs[0]; of[4] = cs[1]; of[5] = cs[2];
  }
}

int launch_embed_points(const float* rays, const float* z, const float* dir_emb, float* x, long R, int N, hipStream_t stream) {
  if (R <= 0 || N <= 0) return 0;
  const long total = R * N * (1 + XYZ_FREQS + 7);
  hipLaunchKernelGGL(embed_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rays, z, dir_emb, x, R, N);
  return check_launch("embed_points_kernel");
}

int launch_posenc(const float* x, float* out, long n, int n_freqs, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n_freqs < 0 || n_freqs > 30) return set_error(-2, "posenc: n_freqs must be in [0, 30]");
  const long total = n * (n_freqs + 1);
  hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, out, n, n_freqs);
  return check_launch("posenc_kernel");
}

}  // namespace crnerf
