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
    host and the call enqueues without any device->host synchronisation.

    Three kernel phases around two all-reduces (crnerf_crossray_decode_sharded_f32): channel sums ->
    all-reduce(64 floats) -> Gram of the centred conv chain -> all-reduce(1024 floats) -> fc / fold /
    apply on the local pixels -> all-gather of RGB."""
    k = kernels or _HipKernels()
    dev = feature_local.device
    n_local = feature_local.shape[0]
    ws = dist.get_world_size(group)
    if equal_shards:
        count = float(n_local * ws)
    else:
        cnt = torch.tensor([float(n_local)], device=dev)
        dist.all_reduce(cnt, group=group)
        count = float(cnt.item())
    sp = style_feature.permute(0, 2, 3, 1).reshape(-1, style_feature.shape[1]).contiguous()
    weights = net.decoder_tensors()
    xchg = torch.zeros(64 + 1024, dtype=torch.float32, device=dev)
    k.crossray_decode_sharded(feature_local, sp, weights, 0, xchg, count)
    dist.all_reduce(xchg[:64], group=group)           # reduction 1: channel sums -> global mean  (linearStyleTransfer.py:59-65)
    k.crossray_decode_sharded(feature_local, sp, weights, 1, xchg, count)
    dist.all_reduce(xchg[64:], group=group)           # reduction 2: Gram of the centred conv chain (linearStyleTransfer.py:29-34)
    rgb_local = k.crossray_decode_sharded(feature_local, sp, weights, 2, xchg, count)
    if rgb_local is None:
        rgb_local = torch.zeros(3, 0, device=dev)
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
    return torch.cat([q[:, :s] for q, s in zip(parts, sizes)], dim=1)


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


class GatherRays(torch.autograd.Function):
    """Ray-parallel training of ONE batch (BASELINE configs[3]: a 65,536-ray grid sharded over the GPUs of a node): the
    renderer runs on this rank's contiguous block of rays; the rows it produces ([n_local, C] features) are all-gathered
    so that every rank decodes the whole grid and evaluates the same full-batch loss (the decoder's statistics are
    cross-ray, and it is ~1 % of the step).  Backward: the local block of the incoming gradient -- every rank holds
    dL/d(full grid) already, so the adjoint of the gather needs NO communication; what does need a sum afterwards are
    the gradients of the sharded part's parameters (the two MLPs), see sync_ray_parallel_gradients."""

    @staticmethod
    def forward(ctx, x, n_total, group):
        ws, rk = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_bounds(n_total, ws, rk)
        if x.shape[0] != hi - lo:
            raise ValueError("crnerf_amd: GatherRays: this rank holds %d rows, its block of %d is [%d, %d)" % (x.shape[0], n_total, lo, hi))
        ctx.block = (lo, hi)
        biggest = shard_bounds(n_total, ws, 0)[1]
        mine = x.contiguous()
        if mine.shape[0] < biggest:                       # uneven split: pad to the largest block
            mine = torch.cat([mine, mine.new_zeros((biggest - mine.shape[0],) + tuple(mine.shape[1:]))])
        parts = [torch.empty_like(mine) for _ in range(ws)]
        dist.all_gather(parts, mine, group=group)
        return torch.cat([p[:shard_bounds(n_total, ws, r)[1] - shard_bounds(n_total, ws, r)[0]] for r, p in enumerate(parts)])

    @staticmethod
    def backward(ctx, g):
        lo, hi = ctx.block
        return g[lo:hi], None, None


def gather_rays(x, n_total, group=None):
    return GatherRays.apply(x, n_total, group)


def sync_ray_parallel_gradients(sharded_modules, replicated_modules, group=None):
    """After loss.backward() of a ray-parallel step: parameters of the sharded part (the MLPs) hold partial sums over this
    rank's rays -> SUM over ranks; parameters of the replicated part (decoder, encoders, mask network) hold the full
    gradient on every rank -> averaged, which only removes rank-to-rank rounding differences (float atomics) so the
    replicas cannot drift.  One flat all-reduce."""
    ws = dist.get_world_size(group)
    for m in replicated_modules:
        for p in m.parameters():
            if p.grad is not None:
                p.grad /= ws
    allreduce_gradients(list(sharded_modules) + list(replicated_modules), group=group, average=False)
