// The transient-mask network of a training step as TWO calls: Context_Guided_Network(classes=1, M=2, N=2) forward with everything its
// backward needs kept in one arena, and the whole backward (models/lightweight_seg.py:274-368; train_mask_grid_sample.py:170-176 runs it
// once per step on the 1/8-scale photo, in train mode: BatchNorm on batch statistics, running buffers updated).
//
// Nothing new is computed here: the chain enqueues cgnet.hip's operators back to back, in the order the module tree evaluates them.  What
// it removes is everything between the launches (and, for F_loc | F_sur -> BatchNorm + PReLU, the launches themselves: one per direction) --
// 56 autograd nodes per direction, nine torch.cat / add launches in the forward and
// seventeen gradient-sum launches in the backward: with batch 1 and NCHW a channel concatenation is two producers writing adjacent slices
// of one buffer, a fan-out's gradient sum is the second data gradient accumulating into the first (ConvGeom::accum), and a block's
// residual add rides in FGlo's scaling pass.  The forward's values are those of the operator-by-operator path bit for bit; gradient sums
// associate differently (tests/test_gpu_cgnet.py holds both against each other and against the reference's fixture).
//
// Parameter order (CGNET_PARAMS = 76 pointers; `grads` mirrors it), names as in the reference's state_dict:
//    0  level1_0.{conv.weight, bn.weight, bn.bias, act.weight}     4  level1_1.{...}     8  level1_2.{...}     12  b1.{bn.weight, bn.bias, act.weight}
//   15  level2_0 (ContextGuidedBlock_Down, 14): conv1x1.{conv.weight, bn.weight, bn.bias, act.weight}, F_loc.conv.weight, F_sur.conv.weight,
//       bn.weight, bn.bias, act.weight, reduce.conv.weight, F_glo.fc.{0.weight, 0.bias, 2.weight, 2.bias}
//   29  level2.0 (ContextGuidedBlock, 13): conv1x1.{4}, F_loc.conv.weight, F_sur.conv.weight, bn_prelu.{bn.weight, bn.bias, act.weight}, F_glo.fc.{4}
//   42  bn_prelu_2.{3}     45  level3_0 (14)     59  level3.0 (13)     72  bn_prelu_3.{3}     75  classifier.0.conv.weight
// BatchNorm order (CGNET_BNS = 14): level1_0, level1_1, level1_2, b1, level2_0.conv1x1, level2_0.bn, level2.0.conv1x1, level2.0.bn_prelu,
//   bn_prelu_2, level3_0.conv1x1, level3_0.bn, level3.0.conv1x1, level3.0.bn_prelu, bn_prelu_3.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace crnerf {

namespace {

inline int half_size(int n) { return (n - 1) / 2 + 1; }         // Conv2d(k=3, stride 2, padding 1) and AvgPool2d(3, 2, 1) alike

struct DownBuf { long d0, y, cat, z, r, stats, bn_a, bn_b; };    // offsets in floats
struct BlockBuf { long e0, y, cat, z, stats, bn_a, bn_b; };
struct Plan {
  int cin, H, W, H1, W1, H2, W2, H3, W3;
  long c0, a0, c1, a1, c2, cat1, b1o, bn10, bn11, bn12, bnb1;
  DownBuf l20; long cat2; BlockBuf l2; long g2, bn2;
  DownBuf l30; long cat3; BlockBuf l3; long g3, bn3, logits, fglo_scratch;
  long total;
};

struct Bump {
  long off = 0;
  long take(long n) { const long o = off; off += (n + 63) & ~63L; return o; }
};

void plan_down(Bump& b, DownBuf& d, int nOut, int R, long hw) {
  d.d0 = b.take(nOut * hw); d.y = b.take(nOut * hw); d.cat = b.take(2 * nOut * hw); d.z = b.take(2 * nOut * hw); d.r = b.take(nOut * hw);
  d.stats = b.take(2 * nOut + R); d.bn_a = b.take(3 * nOut); d.bn_b = b.take(6 * nOut);
}
void plan_block(Bump& b, BlockBuf& k, int nOut, int R, long hw) {
  const int n = nOut / 2;
  k.e0 = b.take(n * hw); k.y = b.take(n * hw); k.cat = b.take(nOut * hw); k.z = b.take(nOut * hw);
  k.stats = b.take(2 * nOut + R); k.bn_a = b.take(3 * n); k.bn_b = b.take(3 * nOut);
}

Plan make_plan(int cin, int H, int W) {
  Plan p{};
  p.cin = cin; p.H = H; p.W = W;
  p.H1 = half_size(H); p.W1 = half_size(W); p.H2 = half_size(p.H1); p.W2 = half_size(p.W1); p.H3 = half_size(p.H2); p.W3 = half_size(p.W2);
  const long hw1 = (long)p.H1 * p.W1, hw2 = (long)p.H2 * p.W2, hw3 = (long)p.H3 * p.W3;
  Bump b;
  p.c0 = b.take(32 * hw1); p.a0 = b.take(32 * hw1); p.c1 = b.take(32 * hw1); p.a1 = b.take(32 * hw1); p.c2 = b.take(32 * hw1);
  p.cat1 = b.take((32 + cin) * hw1); p.b1o = b.take((32 + cin) * hw1);
  p.bn10 = b.take(96); p.bn11 = b.take(96); p.bn12 = b.take(96); p.bnb1 = b.take(3 * (32 + cin));
  plan_down(b, p.l20, 64, 8, hw2);
  p.cat2 = b.take((128 + cin) * hw2);
  plan_block(b, p.l2, 64, 8, hw2);
  p.g2 = b.take((128 + cin) * hw2); p.bn2 = b.take(3 * (128 + cin));
  plan_down(b, p.l30, 128, 8, hw3);
  p.cat3 = b.take(256 * hw3);
  plan_block(b, p.l3, 128, 8, hw3);
  p.g3 = b.take(256 * hw3); p.bn3 = b.take(3 * 256); p.logits = b.take(hw3);
  p.fglo_scratch = b.take(2 * 256);
  p.total = b.off;
  return p;
}

ConvGeom geom(int cin, int cout, int H, int W, int k, int stride, int dil, int depthwise, int accum = 0) {
  ConvGeom g{cin, cout, H, W, 0, 0, k, stride, (k - 1) / 2 * dil, dil, depthwise};
  g.Ho = (H + 2 * g.pad - dil * (k - 1) - 1) / stride + 1;
  g.Wo = (W + 2 * g.pad - dil * (k - 1) - 1) / stride + 1;
  g.accum = accum;
  return g;
}

#define CG_TRY(call) do { const int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

struct Fwd {
  const CgNetArgs& a; float* S; hipStream_t st;
  const float* P(int i) const { return a.params[i]; }
  // BatchNorm2d (batch statistics, running buffers updated) + PReLU; pb = index of bn.weight (bn.bias, act.weight follow); stats: mean, invstd, var_u
  int bn(int ibn, int pb, int C, long hw, const float* x, long stats, float* y) const {
    return launch_cg_bn_prelu_forward(x, P(pb), P(pb + 1), P(pb + 2), S + stats, S + stats + C, S + stats + 2 * C, y, C, (int)hw, a.eps, 1, st,
                                      a.run_mean[ibn], a.run_var[ibn], a.tracked ? a.tracked[ibn] : nullptr, a.momentum);
  }
  // F_loc | F_sur -> cat[2n] -> BatchNorm + PReLU in one launch; pw = index of F_loc.conv.weight (F_sur's, bn.weight, bn.bias, act.weight follow)
  int dwpair(int ibn, int pw, int n, int H, int W, int dil, const float* y, float* cat, long stats, float* z) const {
    const int C = 2 * n;
    return launch_cg_dwpair_bn_prelu_forward(y, P(pw), P(pw + 1), n, H, W, dil, cat, P(pw + 2), P(pw + 3), P(pw + 4), S + stats, S + stats + C, S + stats + 2 * C,
                                             z, a.eps, a.run_mean[ibn], a.run_var[ibn], a.tracked ? a.tracked[ibn] : nullptr, a.momentum, st);
  }
  // ContextGuidedBlock_Down (lightweight_seg.py:164-211): x[nIn,H,W] -> out[nOut,Ho,Wo]
  int down(const float* x, int nIn, int nOut, int H, int W, int dil, int R, int pb, int ibn, const DownBuf& d, float* out) const {
    const ConvGeom g1 = geom(nIn, nOut, H, W, 3, 2, 1, 0);
    const long hw = (long)g1.Ho * g1.Wo;
    CG_TRY(launch_cg_conv_forward(g1, x, P(pb), S + d.d0, st));
    CG_TRY(bn(ibn, pb + 1, nOut, hw, S + d.d0, d.bn_a, S + d.y));
    CG_TRY(dwpair(ibn + 1, pb + 4, nOut, g1.Ho, g1.Wo, dil, S + d.y, S + d.cat, d.bn_b, S + d.z));
    CG_TRY(launch_cg_conv_forward(geom(2 * nOut, nOut, g1.Ho, g1.Wo, 1, 1, 1, 0), S + d.z, P(pb + 9), S + d.r, st));
    return launch_cg_fglo_forward(S + d.r, P(pb + 10), P(pb + 11), P(pb + 12), P(pb + 13), S + d.stats, out, nOut, R, (int)hw, st);
  }
  // ContextGuidedBlock, add=True (lightweight_seg.py:214-255): x[nOut,H,W] -> out = x + F_glo(...)
  int block(const float* x, int nOut, int H, int W, int dil, int R, int pb, int ibn, const BlockBuf& k, float* out) const {
    const int n = nOut / 2;
    const long hw = (long)H * W;
    CG_TRY(launch_cg_conv_forward(geom(nOut, n, H, W, 1, 1, 1, 0), x, P(pb), S + k.e0, st));
    CG_TRY(bn(ibn, pb + 1, n, hw, S + k.e0, k.bn_a, S + k.y));
    CG_TRY(dwpair(ibn + 1, pb + 4, n, H, W, dil, S + k.y, S + k.cat, k.bn_b, S + k.z));
    return launch_cg_fglo_forward(S + k.z, P(pb + 9), P(pb + 10), P(pb + 11), P(pb + 12), S + k.stats, out, nOut, R, (int)hw, st, x);
  }
};

struct Bwd {
  const CgNetArgs& a; const float* S; float* D; float* const* G; hipStream_t st;     // D: the gradient of S + off lives at D + off
  const float* P(int i) const { return a.params[i]; }
  int bn(int pb, int C, long hw, const float* x, long stats, const float* dy, float* dx) const {
    return launch_cg_bn_prelu_backward(x, P(pb), P(pb + 1), P(pb + 2), S + stats, S + stats + C, dy, dx, G[pb], G[pb + 1], G[pb + 2], C, (int)hw, 1, st);
  }
  // backward of Fwd::dwpair: dz[2n] -> d_cat[2n] (scratch), the five parameter gradients, d_y[n] = both data gradients summed
  int dwpair(int pw, int n, int H, int W, int dil, const float* y, const float* cat, long stats, const float* dz, float* d_cat, float* d_y) const {
    const int C = 2 * n;
    return launch_cg_dwpair_bn_prelu_backward(y, P(pw), P(pw + 1), n, H, W, dil, cat, P(pw + 2), P(pw + 3), P(pw + 4), S + stats, S + stats + C, dz, d_cat,
                                              G[pw + 2], G[pw + 3], G[pw + 4], G[pw], G[pw + 1], d_y, st);
  }
  int conv(const ConvGeom& g, const float* x, int pw, const float* dy, float* dx) const { return launch_cg_conv_backward(g, x, P(pw), dy, dx, G[pw], st); }
  // d_out[nOut,Ho,Wo] -> d_x (written, or added to when accum)
  int down(const float* x, float* d_x, int accum, int nIn, int nOut, int H, int W, int dil, int R, int pb, const DownBuf& d, const float* d_out,
           const Plan& p) const {
    const ConvGeom g1 = geom(nIn, nOut, H, W, 3, 2, 1, 0, accum);
    const long hw = (long)g1.Ho * g1.Wo;
    CG_TRY(launch_cg_fglo_backward(S + d.r, P(pb + 10), P(pb + 12), S + d.stats, d_out, D + p.fglo_scratch, D + d.r, G[pb + 10], G[pb + 11], G[pb + 12],
                                   G[pb + 13], nOut, R, (int)hw, st));
    CG_TRY(conv(geom(2 * nOut, nOut, g1.Ho, g1.Wo, 1, 1, 1, 0), S + d.z, pb + 9, D + d.r, D + d.z));
    CG_TRY(dwpair(pb + 4, nOut, g1.Ho, g1.Wo, dil, S + d.y, S + d.cat, d.bn_b, D + d.z, D + d.cat, D + d.y));
    CG_TRY(bn(pb + 1, nOut, hw, S + d.d0, d.bn_a, D + d.y, D + d.d0));
    return conv(g1, x, pb, D + d.d0, d_x);
  }
  // d_out[nOut,H,W]; d_x ALREADY holds the gradient x receives from its other consumers: the conv's data gradient and the residual's are added
  int block(const float* x, float* d_x, int nOut, int H, int W, int dil, int R, int pb, const BlockBuf& k, const float* d_out, const Plan& p) const {
    const int n = nOut / 2;
    const long hw = (long)H * W;
    CG_TRY(launch_cg_fglo_backward(S + k.z, P(pb + 9), P(pb + 11), S + k.stats, d_out, D + p.fglo_scratch, D + k.z, G[pb + 9], G[pb + 10], G[pb + 11],
                                   G[pb + 12], nOut, R, (int)hw, st));
    CG_TRY(dwpair(pb + 4, n, H, W, dil, S + k.y, S + k.cat, k.bn_b, D + k.z, D + k.cat, D + k.y));
    CG_TRY(bn(pb + 1, n, hw, S + k.e0, k.bn_a, D + k.y, D + k.e0));
    CG_TRY(conv(geom(nOut, n, H, W, 1, 1, 1, 0, 1), x, pb, D + k.e0, d_x));
    return launch_cg_add_inplace(d_x, d_out, (int)(nOut * hw), st);
  }
};

}  // namespace

size_t cgnet_arena_floats(int cin, int H, int W) { return (size_t)make_plan(cin, H, W).total; }

int launch_cgnet_forward_train(const CgNetArgs& a, const float* image, float* saved, float* mask, hipStream_t st) {
  const Plan p = make_plan(a.cin, a.H, a.W);
  const Fwd f{a, saved, st};
  float* S = saved;
  const int cin = a.cin;
  const long hw1 = (long)p.H1 * p.W1, hw2 = (long)p.H2 * p.W2, hw3 = (long)p.H3 * p.W3;
  // level 1: three ConvBNPReLU at 1/2 scale; the image re-injected at 1/2 and 1/4 (lightweight_seg.py:332-338)
  CG_TRY(launch_cg_conv_forward(geom(cin, 32, a.H, a.W, 3, 2, 1, 0), image, f.P(0), S + p.c0, st));
  CG_TRY(f.bn(0, 1, 32, hw1, S + p.c0, p.bn10, S + p.a0));
  CG_TRY(launch_cg_conv_forward(geom(32, 32, p.H1, p.W1, 3, 1, 1, 0), S + p.a0, f.P(4), S + p.c1, st));
  CG_TRY(f.bn(1, 5, 32, hw1, S + p.c1, p.bn11, S + p.a1));
  CG_TRY(launch_cg_conv_forward(geom(32, 32, p.H1, p.W1, 3, 1, 1, 0), S + p.a1, f.P(8), S + p.c2, st));
  CG_TRY(f.bn(2, 9, 32, hw1, S + p.c2, p.bn12, S + p.cat1));                                   // cat1 = [level1_2 output | half-scale image]
  CG_TRY(launch_cg_avgpool(image, S + p.cat1 + 32 * hw1, cin, a.H, a.W, 0, st));
  CG_TRY(launch_cg_avgpool(S + p.cat1 + 32 * hw1, S + p.cat2 + 128 * hw2, cin, p.H1, p.W1, 0, st));   // cat2 = [stage2 | stage2_in | quarter-scale image]
  CG_TRY(f.bn(3, 12, 32 + cin, hw1, S + p.cat1, p.bnb1, S + p.b1o));
  // level 2 (:341-349)
  CG_TRY(f.down(S + p.b1o, 32 + cin, 64, p.H1, p.W1, 2, 8, 15, 4, p.l20, S + p.cat2 + 64 * hw2));
  CG_TRY(f.block(S + p.cat2 + 64 * hw2, 64, p.H2, p.W2, 2, 8, 29, 6, p.l2, S + p.cat2));
  CG_TRY(f.bn(8, 42, 128 + cin, hw2, S + p.cat2, p.bn2, S + p.g2));
  // level 3 (:352-359); cat3 = [stage3_in | stage3]
  CG_TRY(f.down(S + p.g2, 128 + cin, 128, p.H2, p.W2, 4, 8, 45, 9, p.l30, S + p.cat3));
  CG_TRY(f.block(S + p.cat3, 128, p.H3, p.W3, 4, 8, 59, 11, p.l3, S + p.cat3 + 128 * hw3));
  CG_TRY(f.bn(13, 72, 256, hw3, S + p.cat3, p.bn3, S + p.g3));
  // classifier + x8 bilinear up-sampling + sigmoid (:362-367)
  CG_TRY(launch_cg_conv_forward(geom(256, 1, p.H3, p.W3, 1, 1, 1, 0), S + p.g3, f.P(75), S + p.logits, st));
  return launch_cg_bilinear(S + p.logits, nullptr, mask, (long)a.H * a.W, p.H3, p.W3, a.H, a.W, 1, st);
}

int launch_cgnet_backward(const CgNetArgs& a, const float* image, const float* saved, const float* mask, const float* d_mask, float* scratch,
                          float* const* grads, hipStream_t st) {
  const Plan p = make_plan(a.cin, a.H, a.W);
  const Bwd b{a, saved, scratch, grads, st};
  const float* S = saved;
  float* D = scratch;
  const int cin = a.cin;
  const long hw1 = (long)p.H1 * p.W1, hw2 = (long)p.H2 * p.W2, hw3 = (long)p.H3 * p.W3;
  CG_TRY(launch_cg_bilinear_backward(mask, d_mask, nullptr, D + p.logits, (long)a.H * a.W, p.H3, p.W3, a.H, a.W, 1, st));
  CG_TRY(b.conv(geom(256, 1, p.H3, p.W3, 1, 1, 1, 0), S + p.g3, 75, D + p.logits, D + p.g3));
  CG_TRY(b.bn(72, 256, hw3, S + p.cat3, p.bn3, D + p.g3, D + p.cat3));
  CG_TRY(b.block(S + p.cat3, D + p.cat3, 128, p.H3, p.W3, 4, 8, 59, p.l3, D + p.cat3 + 128 * hw3, p));
  CG_TRY(b.down(S + p.g2, D + p.g2, 0, 128 + cin, 128, p.H2, p.W2, 4, 8, 45, p.l30, D + p.cat3, p));
  CG_TRY(b.bn(42, 128 + cin, hw2, S + p.cat2, p.bn2, D + p.g2, D + p.cat2));
  CG_TRY(b.block(S + p.cat2 + 64 * hw2, D + p.cat2 + 64 * hw2, 64, p.H2, p.W2, 2, 8, 29, p.l2, D + p.cat2, p));
  CG_TRY(b.down(S + p.b1o, D + p.b1o, 0, 32 + cin, 64, p.H1, p.W1, 2, 8, 15, p.l20, D + p.cat2 + 64 * hw2, p));
  CG_TRY(b.bn(12, 32 + cin, hw1, S + p.cat1, p.bnb1, D + p.b1o, D + p.cat1));                  // the image slices' gradients are not used
  CG_TRY(b.bn(9, 32, hw1, S + p.c2, p.bn12, D + p.cat1, D + p.c2));
  CG_TRY(b.conv(geom(32, 32, p.H1, p.W1, 3, 1, 1, 0), S + p.a1, 8, D + p.c2, D + p.a1));
  CG_TRY(b.bn(5, 32, hw1, S + p.c1, p.bn11, D + p.a1, D + p.c1));
  CG_TRY(b.conv(geom(32, 32, p.H1, p.W1, 3, 1, 1, 0), S + p.a0, 4, D + p.c1, D + p.a0));
  CG_TRY(b.bn(1, 32, hw1, S + p.c0, p.bn10, D + p.a0, D + p.c0));
  return b.conv(geom(cin, 32, a.H, a.W, 3, 2, 1, 0), image, 0, D + p.c0, nullptr);
}

}  // namespace crnerf
