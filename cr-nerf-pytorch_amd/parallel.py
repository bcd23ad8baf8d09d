"""Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).

The renderer (A0-A7) is per-ray independent, so rays are sharded in contiguous blocks with NO
data-path collective.  The only cross-ray step is the decoder's two global reductions (SURVEY 8e,
option B): 64 channel sums and a 32x32 Gram -- two tiny all-reduces (65 and 1024 floats) instead of
an all-gather of the [R,64] feature grid -- after which every rank decodes its own pixels; an optional
all-gather of RGB (12 B/pixel) assembles the image on every rank.

`kernels` is the compute backend (default: the HIP ops); tests inject a CPU stand-in to exercise the
exchange protocol under gloo.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous near-equal block [lo, hi) of n items for `rank` (the first n % world_size ranks get one extra)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rays(rays, group=None):
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(rays.shape[0], ws, rk)
    return rays[lo:hi].contiguous(), (lo, hi)


class _HipKernels:
    def __getattr__(self, name):
        from . import ops
        return getattr(ops, name)


def decode_sharded(net, feature_local, style_feature, group=None, gather=True, kernels=None, equal_shards=False):
    """Cross-ray decode of a ray-sharded feature grid.

    net: style_net; feature_local: this rank's [R_local,64] block of feature_fine (pixel-major, rank
    order = pixel order); style_feature: [1,64,h,w], replicated.  Returns RGB planar: [3, R_total] on
    every rank when gather=True (ranks may hold different R_local), else this rank's [3, R_local].
    equal_shards=True promises every rank holds the same R_local: the pixel count is then known on the
    host and the call enqueues without any device->host synchronisation."""
    k = kernels or _HipKernels()
    dev = feature_local.device
    n_local = feature_local.shape[0]
    # reduction 1: channel sums + pixel count  -> global mean   (linearStyleTransfer.py:59-65)
    stat = torch.cat([k.crossray_chansum(feature_local) if n_local else torch.zeros(64, device=dev),
                      torch.tensor([float(n_local)], device=dev)])
    dist.all_reduce(stat, group=group)
    ws = dist.get_world_size(group)
    c_sum, count = stat[:64].contiguous(), (float(n_local * ws) if equal_shards else float(stat[64].item()))
    # reduction 2: Gram of the centred conv chain        (linearStyleTransfer.py:29-34)
    cnet = net.multi_net.cnet
    gram = k.crossray_gram(feature_local, (c_sum / count).contiguous(), cnet.conv_tensors()) if n_local else torch.zeros(1024, device=dev)
    dist.all_reduce(gram, group=group)
    affine = net.affine_from_stats(c_sum, gram, count, style_feature, kernels=k)
    rgb_local = k.crossray_apply(feature_local, affine) if n_local else torch.zeros(3, 0, device=dev)
    if not gather:
        return rgb_local
    if equal_shards:
        full = torch.empty(ws * 3, n_local, device=dev)
        dist.all_gather_into_tensor(full, rgb_local.contiguous(), group=group)
        return full.view(ws, 3, n_local).permute(1, 0, 2).reshape(3, ws * n_local)
    sizes = [torch.zeros(1, dtype=torch.long, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.long, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    pad = torch.zeros(3, width, device=dev)
    pad[:, :n_local] = rgb_local
    parts = [torch.empty(3, width, device=dev) for _ in range(ws)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:, :s] for p, s in zip(parts, sizes)], dim=1)


def allreduce_gradients(modules, group=None, average=True):
    """Data-parallel training (the reference: Lightning DDP, train_mask_grid_sample.py:441-450): sum (or
    average) every parameter gradient over the ranks with ONE flat all-reduce (~16 MB for the two MLPs +
    decoder: a single large message suits xGMI's per-link-bound rings better than per-tensor buckets).
    Call after loss.backward(), before optimizer.step().  Parameters without a gradient contribute zeros."""
    params = [p for m in modules for p in m.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
