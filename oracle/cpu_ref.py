"""CPU oracle: a plain-PyTorch (fp32, CPU) restatement of the reference's rendering hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.

The reference is pure Python/PyTorch (no native code), so the oracle is Python/PyTorch too; a C
restatement would have to re-implement ATen's sin/cos/exp/sgemm and would pin nothing extra.
PARITY PIN: tests/golden/*.npz were produced by importing the reference itself in the build
container (tests/golden/make_golden.py); tests/test_oracle.py checks every function below against
them, so the oracle is pinned to the reference's own outputs.

Each function cites the reference lines (under /root/reference) that it restates.  Weights are plain
dicts name -> tensor using the reference's state_dict keys.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ models/nerf.py
def posenc(x, n_freqs):
    """PosEmbedding.forward, models/nerf.py:17-30, with freqs 2^0..2^(n_freqs-1) (ctor :12-13)."""
    cols = [x]
    for k in range(n_freqs):
        scaled = float(2 ** k) * x
        cols.append(scaled.sin())
        cols.append(scaled.cos())
    return torch.cat(cols, dim=-1)


def mlp_forward(w, x, sigma_only=False):
    """NeRF_sigma.forward, models/nerf.py:157-182.  x: [p,120] (or [p,93] when sigma_only)."""
    xyz = x if sigma_only else x[:, :93]
    h = xyz
    for layer in range(1, 9):
        if layer == 5:  # skip connection, input first (:168-169)
            h = torch.cat((xyz, h), dim=1)
        h = F.relu_(F.linear(h, w["xyz_encoding_%d.0.weight" % layer], w["xyz_encoding_%d.0.bias" % layer]))  # nn.ReLU(True), :142
    sigma = F.softplus(F.linear(h, w["static_sigma.0.weight"], w["static_sigma.0.bias"]))  # :172, beta=1 thr=20
    if sigma_only:
        return sigma
    final = F.linear(h, w["xyz_encoding_final.weight"], w["xyz_encoding_final.bias"])       # :176
    g = F.relu_(F.linear(torch.cat((final, x[:, 93:]), dim=1), w["dir_encoding.0.weight"], w["dir_encoding.0.bias"]))  # :153
    feat = torch.sigmoid(F.linear(g, w["static_rgb.0.weight"], w["static_rgb.0.bias"]))      # :180
    return torch.cat((feat, sigma), dim=-1)                                                  # :181


def bf16_round(t):
    """Round-to-nearest-even to bfloat16, returned as fp32 (what v_cvt_pk_bf16_f32 does to an MFMA operand)."""
    return t.to(torch.bfloat16).to(torch.float32)


def mlp_forward_bf16(w, x, sigma_only=False):
    """NeRF_sigma.forward, models/nerf.py:157-182, in the mixed precision of the crnerf_*_bf16 entry points
    (include/crnerf.h): operands of every Linear except static_sigma -- weights and input activations, incl. the
    embedded input x -- rounded to bf16, fp32 accumulation; biases / activations / the sigma head in fp32, the latter
    on the UN-rounded output of xyz_encoding_8.  Products of two bf16 values are exact in fp32, so F.linear on the
    rounded operands differs from the MFMA only in summation order."""
    q = bf16_round
    lin = lambda h, name: F.linear(q(h), q(w[name + ".weight"]), w[name + ".bias"])  # noqa: E731
    xyz = x if sigma_only else x[:, :93]
    h = xyz
    for layer in range(1, 9):
        if layer == 5:
            h = torch.cat((xyz, h), dim=1)
        h = F.relu(lin(h, "xyz_encoding_%d.0" % layer))
    sigma = F.softplus(F.linear(h, w["static_sigma.0.weight"], w["static_sigma.0.bias"]))
    if sigma_only:
        return sigma
    final = lin(h, "xyz_encoding_final")
    g = F.relu(lin(torch.cat((final, x[:, 93:]), dim=1), "dir_encoding.0"))
    feat = torch.sigmoid(lin(g, "static_rgb.0"))
    return torch.cat((feat, sigma), dim=-1)


class _LinearBf16(torch.autograd.Function):
    """nn.Linear in the arithmetic of the mixed-precision training twins (crnerf_mlp_*_mixed_f32): both operands of each of the
    three products -- forward, data gradient, weight gradient -- rounded to bf16, fp32 accumulation.  The twins STORE activations and
    layer deltas as bf16, so the bias gradient is the column sum of the rounded delta."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(bf16_round(x), bf16_round(w), b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gq = bf16_round(g)
        return gq @ bf16_round(w), gq.t() @ bf16_round(x), gq.sum(0)


class _SigmaHeadStoredBf16(torch.autograd.Function):
    """static_sigma in the mixed-precision twins: evaluated in fp32 on the un-rounded output of xyz_encoding_8 (forward and the branch
    into d(h8)); its WEIGHT gradient reads the activation row as it was stored, i.e. rounded to bf16."""

    @staticmethod
    def forward(ctx, h, w, b):
        ctx.save_for_backward(h, w)
        return F.linear(h, w, b)

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        return g @ w, g.t() @ bf16_round(h), g.sum(0)


def mlp_forward_bf16_train(w, x):
    """mlp_forward_bf16 with a backward: same forward values, gradients as the mixed-precision twins define them."""
    lin = lambda h, name: _LinearBf16.apply(h, w[name + ".weight"], w[name + ".bias"])  # noqa: E731
    xyz = x[:, :93]
    h = xyz
    for layer in range(1, 9):
        if layer == 5:
            h = torch.cat((xyz, h), dim=1)
        h = F.relu(lin(h, "xyz_encoding_%d.0" % layer))
    sigma = F.softplus(_SigmaHeadStoredBf16.apply(h, w["static_sigma.0.weight"], w["static_sigma.0.bias"]))
    final = lin(h, "xyz_encoding_final")
    g = F.relu(lin(torch.cat((final, x[:, 93:]), dim=1), "dir_encoding.0"))
    feat = torch.sigmoid(lin(g, "static_rgb.0"))
    return torch.cat((feat, sigma), dim=-1)


# ------------------------------------------------------------------ models/rendering.py
def composite(raw, z, noise=None, noise_std=0.0):
    """Compositing half of inference(), models/rendering.py:116-143.
    raw [R,N,65], z [R,N] -> weights [R,N], feature [R,64], depth [R]."""
    feats, sigma = raw[..., :64], raw[..., 64]
    delta = torch.cat((z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e2)), dim=-1)     # :121-123
    if noise is not None:
        sigma = sigma + noise * noise_std                                                  # :125-126
    alpha = 1 - torch.exp(-delta * torch.relu(sigma))
    trans = torch.cumprod(torch.cat((torch.ones_like(alpha[:, :1]), 1 - alpha), dim=-1)[:, :-1], dim=-1)  # :128-130
    weights = alpha * trans                                                                # :132
    feature = (weights.unsqueeze(-1) * feats).sum(dim=1)                                   # :136-138
    depth = (weights * z).sum(dim=1)                                                       # :143
    return weights, feature, depth


def sample_pdf(bins, weights, n_importance, det=True, u=None, eps=1e-5):
    """sample_pdf, models/rendering.py:7-46.  bins [R,M+1], weights [R,M]; u overrides the draw."""
    R, M = weights.shape
    wts = weights + eps
    pdf = wts / wts.sum(dim=1, keepdim=True)
    cdf = torch.cat((torch.zeros(R, 1, dtype=pdf.dtype), torch.cumsum(pdf, dim=-1)), dim=-1)  # :22-23
    if u is None:
        u = torch.linspace(0, 1, n_importance).expand(R, n_importance) if det else torch.rand(R, n_importance)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)                                           # :33
    lo, hi = (idx - 1).clamp_min(0), idx.clamp_max(M)
    c_lo, c_hi = cdf.gather(1, lo), cdf.gather(1, hi)
    b_lo, b_hi = bins.gather(1, lo), bins.gather(1, hi)
    denom = c_hi - c_lo
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)                        # :41-42
    return b_lo + (u - c_lo) / denom * (b_hi - b_lo)                                        # :45


def coarse_depths(rays, n_samples, use_disp=False, z_steps=None):
    """models/rendering.py:160-167 (perturb == 0).  z_steps: optional precomputed linspace(0,1,n) table
    (ATen's CPU linspace differs in the last bit between SIMD widths; fixtures carry the table used)."""
    near, far = rays[:, 6:7], rays[:, 7:8]
    s = torch.linspace(0, 1, n_samples) if z_steps is None else z_steps
    if use_disp:
        z = 1 / (1 / near * (1 - s) + 1 / far * s)
    else:
        z = near * (1 - s) + far * s
    return z.expand(rays.shape[0], n_samples)


def fine_depths(z_coarse, weights_coarse, n_importance, det=True, u=None):
    """models/rendering.py:183-187: z_mid, sample_pdf on weights[:,1:-1], sort(cat).
    u: [R,Ni] uniforms, or a shared [Ni] row (e.g. a precomputed linspace table)."""
    mid = 0.5 * (z_coarse[:, :-1] + z_coarse[:, 1:])
    if u is not None and u.dim() == 1:
        u = u.expand(z_coarse.shape[0], n_importance)
    extra = sample_pdf(mid, weights_coarse[:, 1:-1], n_importance, det=det, u=u)
    return torch.sort(torch.cat((z_coarse, extra), dim=-1), dim=-1)[0], extra


def _run_model(w, rays, z, dir_emb, chunk, mlp=None):
    """Point-chunk loop of inference(), models/rendering.py:100-116."""
    mlp = mlp or mlp_forward
    R, N = z.shape
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)          # :178 / :188
    dirs = dir_emb[:, None, :].expand(R, N, dir_emb.shape[-1]).reshape(R * N, -1)           # :108
    outs = []
    for i in range(0, R * N, chunk):
        outs.append(mlp(w, torch.cat((posenc(pts[i:i + chunk], 15), dirs[i:i + chunk]), dim=1)))
    return torch.cat(outs, dim=0).view(R, N, 65)


def render_rays(w_coarse, w_fine, rays, n_samples, n_importance, use_disp=False, chunk=32768, view_dir=None,
                z_coarse=None, u=None, noise_coarse=None, noise_fine=None, noise_std=0.0, return_raw=False, z_steps=None,
                precision="f32", z_fine=None):
    """render_rays_cross_ray, models/rendering.py:50-196, with perturb == 0 unless z_coarse/u are supplied.
    precision="bf16": NeRF_sigma through mlp_forward_bf16, everything else unchanged.  z_fine: evaluate the fine
    pass at these depths instead of sampling them (used to compare fine outputs at identical depths)."""
    mlp = mlp_forward_bf16 if precision == "bf16" else mlp_forward
    dir_emb = posenc(rays[:, 3:6] if view_dir is None else view_dir, 4)                     # :155
    z = coarse_depths(rays, n_samples, use_disp, z_steps) if z_coarse is None else z_coarse
    out = {}
    raw_c = _run_model(w_coarse, rays, z, dir_emb, chunk, mlp)
    out["weights_coarse"], out["feature_coarse"], out["depth_coarse"] = composite(raw_c, z, noise_coarse, noise_std)
    if return_raw:
        out["raw_coarse"], out["z_coarse"] = raw_c, z
    if n_importance > 0:
        z_f = fine_depths(z, out["weights_coarse"], n_importance, det=(u is None), u=u)[0] if z_fine is None else z_fine
        raw_f = _run_model(w_fine, rays, z_f, dir_emb, chunk, mlp)
        out["weights_fine"], out["feature_fine"], out["depth_fine"] = composite(raw_f, z_f, noise_fine, noise_std)
        out["z_fine"] = z_f
        if return_raw:
            out["raw_fine"] = raw_f
    return out


# ------------------------------------------------------------------ models/linearStyleTransfer.py, nerf_decoder_stylenerf.py
def _conv1x1(x, w, b):
    """1x1 Conv2d on a [C,P] matrix of P pixels."""
    return w.reshape(w.shape[0], -1) @ x + b[:, None]


def cnn_matrix(d, net, x):
    """CNN.forward, models/linearStyleTransfer.py:28-37 on a centred [64,P] matrix -> [32,32]."""
    p = "multi_net.%s." % net
    h = F.leaky_relu(_conv1x1(x, d[p + "convs.0.weight"], d[p + "convs.0.bias"]), 0.2)
    h = F.leaky_relu(_conv1x1(h, d[p + "convs.2.weight"], d[p + "convs.2.bias"]), 0.2)
    h = _conv1x1(h, d[p + "convs.4.weight"], d[p + "convs.4.bias"])
    gram = (h @ h.t()) / x.shape[1]
    return F.linear(gram.reshape(1, -1), d[p + "fc.weight"], d[p + "fc.bias"]).view(32, 32)


def crossray_decode(d, content, style, mode=None):
    """style_net.forward :284-291 -> MulLayer.forward :58-94 -> NeuralRenderer.forward (n_blocks=0)
    nerf_decoder_stylenerf.py:279-291.  content [1,64,H,W], style [1,64,h,w] or None -> [1,3,H,W]."""
    _, C, H, W = content.shape
    x = content.reshape(C, H * W)
    if style is None and mode == "content":
        fused = x
    else:
        s = style.reshape(C, -1)
        c_mean, s_mean = x.mean(dim=1, keepdim=True), s.mean(dim=1, keepdim=True)
        xc, sc = x - c_mean, s - s_mean
        comp = _conv1x1(xc, d["multi_net.compress.weight"], d["multi_net.compress.bias"])    # :76-78
        trans = cnn_matrix(d, "snet", sc) @ cnn_matrix(d, "cnet", xc)                         # :81-86
        fused = _conv1x1(trans @ comp, d["multi_net.unzip.weight"], d["multi_net.unzip.bias"]) + s_mean  # :87-89
    rgb = torch.sigmoid(_conv1x1(fused, d["decoder.feat_2_rgb_list.0.weight"], d["decoder.feat_2_rgb_list.0.bias"]))
    return rgb.view(1, 3, H, W)


def feature_to_grid(feature, H, W):
    """Caller-side glue, eval.py:291-292 / train_mask_grid_sample.py:132-133: [HW,64] -> [1,64,H,W]."""
    return feature.t().reshape(1, feature.shape[1], H, W)


def encoder_forward(d, img):
    """encoder_sameoutputsize.forward, models/linearStyleTransfer.py:252-276.  d: conv{1..7}.{weight,bias}; img [1,3,H,W]."""
    def c3(x, i):
        return F.leaky_relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), d["conv%d.weight" % i], d["conv%d.bias" % i]), 0.2)
    x = F.conv2d(img, d["conv1.weight"], d["conv1.bias"])
    x = F.max_pool2d(c3(c3(x, 2), 3), 2, 2)
    x = F.max_pool2d(c3(c3(x, 4), 5), 2, 2)
    x = F.adaptive_avg_pool2d(c3(x, 6), 32)
    return F.leaky_relu(F.conv2d(x, d["conv7.weight"], d["conv7.bias"]), 0.2)


def generate_rays(H, W, K, c2w, near=0.0, far=5.0):
    """datasets/ray_utils.py:5-52 (get_ray_directions + get_rays) and the near/far columns of
    datasets/PhototourismDataset.py:17-22.  Returns (directions[H,W,3], rays[H*W,8])."""
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    K = torch.as_tensor(K, dtype=torch.float64)
    dirs = torch.stack(((i - float(K[0, 2])) / float(K[0, 0]), -(j - float(K[1, 2])) / float(K[1, 1]), -torch.ones_like(i)), -1)
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    d = dirs @ c2w[:, :3].t()
    d = (d / d.norm(dim=-1, keepdim=True)).reshape(-1, 3)
    o = c2w[:, 3].expand_as(d)
    return dirs, torch.cat((o, d, torch.full_like(d[:, :1], near), torch.full_like(d[:, :1], far)), 1)


# ------------------------------------------------------------------ losses.py, datasets/phototourism_mask_grid_sample.py
LOSS_KEYS = ("kl_a", "rec_a_random", "c_l", "content_constraint", "r_ms", "r_md", "f_l")


def annealing_weight(step, vmax, vmin, k):
    """ExponentialAnnealingWeight.getWeight, losses.py:30-39."""
    return max(vmin, vmax * math.exp(-step * k))


def crnerf_loss(inputs, targets, hp, global_step, coef=1.0):
    """CRNeRFLoss.forward, losses.py:49-78 (+ mask_regularize :80-91, _l2_regularize :93-96).  inputs: dict of tensors
    with the reference's keys; hp: object with maskrs_max/min/k, maskrd, weightKL, weightRecA, weightcontent,
    mse_on_appearance.  Returns (dict in the reference's insertion order, annealing weight)."""
    ret = {}
    ann = annealing_weight(global_step, hp.maskrs_max, hp.maskrs_min, hp.maskrs_k)
    if "a_embedded" in inputs:
        ret["kl_a"] = (inputs["a_embedded"] ** 2).mean() * hp.weightKL
        if "a_embedded_random_rec" in inputs:
            d = inputs["a_embedded_random"].detach() - inputs["a_embedded_random_rec"]
            ret["rec_a_random"] = ((d ** 2).mean() if hp.mse_on_appearance else d.abs().mean()) * hp.weightRecA
    if "out_mask" in inputs:
        mask = inputs["out_mask"]
        ret["c_l"] = 0.5 * ((1 - mask.detach()) * (inputs["rgb_coarse"] - targets) ** 2).mean()
    else:
        ret["c_l"] = 0.5 * ((inputs["rgb_coarse"] - targets) ** 2).mean()
    if "content_wo_a_embed" in inputs and "content_with_a_embed" in inputs:
        ret["content_constraint"] = ((inputs["content_wo_a_embed"] - inputs["content_with_a_embed"]) ** 2).mean() * hp.weightcontent
    if "rgb_fine" in inputs:
        if "out_mask" in inputs:
            ret["r_ms"] = (mask ** 2).mean() * ann
            ret["r_md"] = (1 / ((mask - 0.5) ** 2 + 0.02)).mean() * hp.maskrd
            ret["f_l"] = 0.5 * ((1 - mask) * (inputs["rgb_fine"] - targets) ** 2).mean()
        else:
            ret["f_l"] = 0.5 * ((inputs["rgb_fine"] - targets) ** 2).mean()
    return {k: coef * v for k, v in ret.items()}, ann


def grid_sample_indices(img_w, img_h, side, scale, h_offset, w_offset):
    """The index arithmetic of PhototourismDataset.__getitem__ (train), phototourism_mask_grid_sample.py:247-262:
    a side x side lattice over [0, 1-1/w) x [0, 1-1/h), shrunk by `scale` and shifted, floor-ed to pixels.
    scale / offsets are the three torch.Tensor(1).uniform_ draws (fp32).  Returns (img_sample_points int64 [side^2],
    uv_sample fp32 [side^2, 2])."""
    # img_w / img_h are elements of the int64 tensor all_imgs_wh in the reference, so 1 - 1/img_w is an fp32 TENSOR
    # expression (not Python double arithmetic) -- the lattice end points carry that rounding
    img_w, img_h = torch.as_tensor(img_w, dtype=torch.int64), torch.as_tensor(img_h, dtype=torch.int64)
    w_lin = torch.linspace(0, 1 - 1 / img_w, side)
    h_lin = torch.linspace(0, 1 - 1 / img_h, side)
    w_samples, h_samples = torch.meshgrid([w_lin, h_lin], indexing="ij")
    scale, h_offset, w_offset = (torch.as_tensor(v, dtype=torch.float32).reshape(1) for v in (scale, h_offset, w_offset))
    h_sb = h_samples * scale + h_offset
    w_sb = w_samples * scale + w_offset
    h = (h_sb * img_h).floor()
    w = (w_sb * img_w).floor()
    pts = (w + h * img_w).permute(1, 0).contiguous().view(-1).long()
    uv = torch.cat((h_sb.permute(1, 0).contiguous().view(-1, 1), w_sb.permute(1, 0).contiguous().view(-1, 1)), -1)
    return pts, uv


def grid_sample_batch(all_rays, all_rgbs, wh, sample_ts, side, scale, h_offset, w_offset):
    """The gathers of the same method, :264-275: rows of the flat ray / rgb buffers at image `sample_ts`."""
    img_w, img_h = int(wh[sample_ts][0]), int(wh[sample_ts][1])
    pts, uv = grid_sample_indices(img_w, img_h, side, scale, h_offset, w_offset)
    rows = pts + int((wh[:sample_ts, 0] * wh[:sample_ts, 1]).sum())
    return {"rays": all_rays[rows, :8], "ts": all_rays[rows, 8].long(), "rgbs": all_rgbs[rows], "rgb_idx": pts, "uv_sample": uv}


def psnr(a, b):
    """metrics.py:12-13."""
    return -10.0 * math.log10(float(((a - b) ** 2).mean()))


def to_torch(state):
    return {k: torch.from_numpy(v) if not torch.is_tensor(v) else v for k, v in state.items()}
