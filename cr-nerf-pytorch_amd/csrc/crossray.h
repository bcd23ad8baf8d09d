// Launchers of the cross-ray decoder kernels (crossray.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace crnerf {

constexpr int CROSSRAY_SUM_BLOCKS = 256;   // partial rows of the channel-sum reduction (one 1024-thread workgroup per CU)
constexpr int CROSSRAY_GRAM_BLOCKS = 256;  // partial Grams (one workgroup per CU)
// two jobs x (sum partials + Gram partials) + stats; see WS_* in crossray.hip
constexpr size_t CROSSRAY_WORKSPACE_BYTES = (size_t)2560 * 1024;

struct CnnTensors {  // CNN.convs, models/linearStyleTransfer.py:11-15 (1x1 convs = [out,in] matrices)
  const float* w1; const float* b1;  // 64 -> 128
  const float* w2; const float* b2;  // 128 -> 64
  const float* w3; const float* b3;  // 64 -> 32
};
struct FoldTensors {  // MulLayer.compress/unzip :54-55, NeuralRenderer.feat_2_rgb_list[0]
  const float* comp_w; const float* comp_b;    // [32,64], [32]
  const float* unzip_w; const float* unzip_b;  // [64,32], [64]
  const float* rgb_w; const float* rgb_b;      // [3,64], [3]
};
struct DecodeArgs {
  const float* content; long HW;
  const float* style; long HWs;      // style == nullptr: type == "content" (decoder only)
  CnnTensors snet; const float* snet_fc_w; const float* snet_fc_b;
  CnnTensors cnet; const float* cnet_fc_w; const float* cnet_fc_b;
  FoldTensors lin;
  void* workspace;
  float* rgb; long plane_stride;
};

int launch_crossray_chansum(const float* x, long HW, float* sum_out, float* workspace, hipStream_t stream);
int launch_crossray_gram(const float* x, long HW, const float* mean, const CnnTensors& w, float* gram_sum, float* workspace,
                         hipStream_t stream);
int launch_crossray_matrix(const float* gram_sum, double count, const float* fc_w, const float* fc_b, float* out, hipStream_t stream);
int launch_crossray_fold(const float* sM, const float* cM, const float* c_mean, const float* s_mean, const FoldTensors& w,
                         float* affine, hipStream_t stream);
int launch_crossray_apply(const float* x, long HW, const float* affine, float* rgb, long plane_stride, hipStream_t stream);
int launch_crossray_decode(const DecodeArgs& d, hipStream_t stream);
int launch_crossray_decode_sharded(const DecodeArgs& d, int phase, float* xchg, double count_global, hipStream_t stream);
size_t crossray_backward_workspace_floats(long HW, long HWs);
int launch_crossray_decode_backward(const DecodeArgs& d, const float* d_rgb, long d_plane_stride, float* workspace, float* d_content,
                                    float* d_style, float* const* grads, hipStream_t stream);
// the same in three phases around two all-reduces, for a ray-sharded content grid (crossray.hip); phase -1 = the one-GPU call above
int launch_crossray_decode_backward_sharded(const DecodeArgs& d, const float* d_rgb, long d_plane_stride, float* workspace, float* d_content,
                                            float* d_style, float* const* grads, int phase, const float* fwd_xchg, double count_global, float* xb,
                                            hipStream_t stream);

}  // namespace crnerf
