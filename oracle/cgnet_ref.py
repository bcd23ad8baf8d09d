"""CPU restatement (numpy, float64 accumulation optional) of the transient-mask network's forward pass -- TEST
INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

Follows models/lightweight_seg.py of the reference:
  conv2d            nn.Conv2d(bias=False), dense or depth-wise, as configured in :12-140
  bn_prelu          nn.BatchNorm2d(eps=1e-3) (batch statistics when training) + nn.PReLU          :12-52
  avgpool3s2        nn.AvgPool2d(3, stride=2, padding=1) (padding counted)                          :258-270
  fglo              FGlo.forward                                                                    :143-162
  bilinear          F.interpolate(mode='bilinear', align_corners=False) (ATen area_pixel_compute_source_index)
  cgnet_forward     Context_Guided_Network.forward                                                  :329-368
  mask_at_pixels    train_mask_grid_sample.py:172-175
Pinned by tests/golden/g13_cgnet.npz (outputs of the imported reference; tests/test_oracle.py).
"""
import numpy as np


def conv2d(x, w, stride=1, dilation=1, groups=1):
    """x[C,H,W], w[Co,C/groups,k,k]; padding = (k-1)//2 * dilation as every reference layer sets it."""
    C, H, W = x.shape
    Co, cpg, k, _ = w.shape
    pad = (k - 1) // 2 * dilation
    Ho = (H + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    xp = np.zeros((C, H + 2 * pad, W + 2 * pad), x.dtype)
    xp[:, pad:pad + H, pad:pad + W] = x
    y = np.zeros((Co, Ho, Wo), x.dtype)
    for ky in range(k):
        for kx in range(k):
            win = xp[:, ky * dilation: ky * dilation + (Ho - 1) * stride + 1: stride, kx * dilation: kx * dilation + (Wo - 1) * stride + 1: stride]
            if groups == 1:
                y += np.einsum("oc,chw->ohw", w[:, :, ky, kx], win)
            else:
                y += w[:, 0, ky, kx][:, None, None] * win
    return y


def bn_prelu(x, p, prefix_bn, prefix_act, training, eps=1e-3, stats_out=None):
    if training:
        mean, var = x.mean(axis=(1, 2)), x.var(axis=(1, 2))
        if stats_out is not None:
            n = x.shape[1] * x.shape[2]
            stats_out[prefix_bn + ".running_mean"] = 0.9 * p[prefix_bn + ".running_mean"] + 0.1 * mean
            stats_out[prefix_bn + ".running_var"] = 0.9 * p[prefix_bn + ".running_var"] + 0.1 * var * n / (n - 1)
    else:
        mean, var = p[prefix_bn + ".running_mean"], p[prefix_bn + ".running_var"]
    z = (x - mean[:, None, None]) / np.sqrt(var[:, None, None] + eps) * p[prefix_bn + ".weight"][:, None, None] + p[prefix_bn + ".bias"][:, None, None]
    return np.where(z > 0, z, p[prefix_act + ".weight"][:, None, None] * z)


def avgpool3s2(x):
    C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = np.zeros((C, H + 2, W + 2), x.dtype)
    xp[:, 1:H + 1, 1:W + 1] = x
    y = np.zeros((C, Ho, Wo), x.dtype)
    for ky in range(3):
        for kx in range(3):
            y += xp[:, ky: ky + 2 * (Ho - 1) + 1: 2, kx: kx + 2 * (Wo - 1) + 1: 2]
    return y / 9.0


def fglo(x, p, prefix):
    m = x.mean(axis=(1, 2))
    h = np.maximum(p[prefix + ".fc.0.weight"] @ m + p[prefix + ".fc.0.bias"], 0.0)
    s = 1.0 / (1.0 + np.exp(-(p[prefix + ".fc.2.weight"] @ h + p[prefix + ".fc.2.bias"])))
    return x * s[:, None, None]


def _src(o, n_in, n_out, dtype):
    s = (np.arange(n_out, dtype=dtype)[o] + dtype(0.5)) * (dtype(n_in) / dtype(n_out)) - dtype(0.5)
    s = np.maximum(s, dtype(0))
    i0 = s.astype(np.int64)
    i1 = i0 + (i0 < n_in - 1)
    return i0, i1, s - i0.astype(dtype)


def bilinear(x, size, idx=None):
    """x[h,w] -> interpolate(size) values at flat output indices idx (None: the whole [Ho,Wo] map)."""
    h, w = x.shape
    Ho, Wo = size
    flat = np.arange(Ho * Wo) if idx is None else np.asarray(idx)
    dt = np.float32   # ATen computes the source index in the tensor's opmath type; the reference runs in fp32
    y0, y1, ly = _src(flat // Wo, h, Ho, dt)
    x0, x1, lx = _src(flat % Wo, w, Wo, dt)
    ly, lx = ly.astype(x.dtype), lx.astype(x.dtype)
    out = (1 - ly) * ((1 - lx) * x[y0, x0] + lx * x[y0, x1]) + ly * ((1 - lx) * x[y1, x0] + lx * x[y1, x1])
    return out.reshape(Ho, Wo) if idx is None else out


def _cg_down(x, p, name, dil, training, stats):
    o = bn_prelu(conv2d(x, p[name + ".conv1x1.conv.weight"], stride=2), p, name + ".conv1x1.bn", name + ".conv1x1.act", training, stats_out=stats)
    C = o.shape[0]
    joi = np.concatenate([conv2d(o, p[name + ".F_loc.conv.weight"], groups=C), conv2d(o, p[name + ".F_sur.conv.weight"], dilation=dil, groups=C)], 0)
    joi = bn_prelu(joi, p, name + ".bn", name + ".act", training, stats_out=stats)
    return fglo(conv2d(joi, p[name + ".reduce.conv.weight"]), p, name + ".F_glo")


def _cg_block(x, p, name, dil, training, stats):
    o = bn_prelu(conv2d(x, p[name + ".conv1x1.conv.weight"]), p, name + ".conv1x1.bn", name + ".conv1x1.act", training, stats_out=stats)
    C = o.shape[0]
    joi = np.concatenate([conv2d(o, p[name + ".F_loc.conv.weight"], groups=C), conv2d(o, p[name + ".F_sur.conv.weight"], dilation=dil, groups=C)], 0)
    joi = bn_prelu(joi, p, name + ".bn_prelu.bn", name + ".bn_prelu.act", training, stats_out=stats)
    return x + fglo(joi, p, name + ".F_glo")


def cgnet_forward(img, p, training, M=2, N=2, stats_out=None):
    """img[3,H,W], p: state_dict as numpy arrays -> mask[H,W] in (0,1).  stats_out (dict) receives the updated running statistics."""
    x = img
    for l in ("level1_0", "level1_1", "level1_2"):
        x = bn_prelu(conv2d(x, p[l + ".conv.weight"], stride=2 if l == "level1_0" else 1), p, l + ".bn", l + ".act", training, stats_out=stats_out)
    inp1 = avgpool3s2(img)
    inp2 = avgpool3s2(inp1)
    o1_0 = _cg_down(bn_prelu(np.concatenate([x, inp1], 0), p, "b1.bn", "b1.act", training, stats_out=stats_out), p, "level2_0", 2, training, stats_out)
    o1 = o1_0
    for i in range(M - 1):
        o1 = _cg_block(o1, p, "level2.%d" % i, 2, training, stats_out)
    cat = bn_prelu(np.concatenate([o1, o1_0, inp2], 0), p, "bn_prelu_2.bn", "bn_prelu_2.act", training, stats_out=stats_out)
    o2_0 = _cg_down(cat, p, "level3_0", 4, training, stats_out)
    o2 = o2_0
    for i in range(N - 1):
        o2 = _cg_block(o2, p, "level3.%d" % i, 4, training, stats_out)
    cat = bn_prelu(np.concatenate([o2_0, o2], 0), p, "bn_prelu_3.bn", "bn_prelu_3.act", training, stats_out=stats_out)
    score = conv2d(cat, p["classifier.0.conv.weight"])[0]
    return 1.0 / (1.0 + np.exp(-bilinear(score, img.shape[1:])))


def mask_at_pixels(mask, hw_whole, rgb_idx):
    """interpolate(mask, hw_whole)[rgb_idx] -> [n,1]"""
    return bilinear(mask, tuple(int(v) for v in hw_whole), rgb_idx)[:, None]
