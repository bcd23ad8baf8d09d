"""Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).

The renderer (A0-A7) is per-ray independent, so rays are sharded in contiguous blocks with NO
data-path collective.  The only cross-ray step is the decoder's two global reductions (SURVEY 8e,
option B): 64 channel sums and a 32x32 Gram -- two tiny all-reduces (65 and 1024 floats) instead of
an all-gather of the [R,64] feature grid -- after which every rank decodes its own pixels; an optional
all-gather of RGB (12 B/pixel) assembles the image on every rank.

`kernels` is the compute backend (default: the HIP ops); tests inject a CPU stand-in to exercise the
exchange protocol under gloo.

The two reductions travel by RCCL all-reduce by default.  `PeerExchange` is the opt-in alternative for them (ranks of ONE
node): a HIP-IPC window per rank and one single-workgroup push / flag / sum kernel per reduction (csrc/peer_xchg.hip) --
pass it as `decode_sharded(..., exchange=...)`.
"""
import ctypes
import os

import torch
import torch.distributed as dist


# ---- optional timing of the collectives (bench.py's `collectives` record): HIP events on the current stream around every
# collective this module issues, so that a multi-GPU run can show what the exchange steps cost per step.  Off by default.
_TIMERS = None


def collect_collective_times(enable=True):
    """Start (or stop) recording an event pair around every collective issued through this module."""
    global _TIMERS
    _TIMERS = {} if enable else None


_TIMER_PENDING_MAX = 64      # event pairs kept per collective before they are folded into (calls, total ms)


def collective_times_ms():
    """{name: {"calls": n, "ms_per_call": t}} of what was recorded since collect_collective_times(); synchronises the device."""
    if not _TIMERS:
        return {}
    torch.cuda.synchronize()
    out = {}
    for k, rec in _TIMERS.items():
        _fold(rec, 0)
        out[k] = {"calls": rec["calls"], "ms_per_call": rec["ms"] / max(rec["calls"], 1)}
    return out


def _fold(rec, keep):
    """Fold all but the newest `keep` pending event pairs into the running totals (waits for the pairs it folds)."""
    while len(rec["pending"]) > keep:
        a, b = rec["pending"].pop(0)
        b.synchronize()
        rec["calls"] += 1
        rec["ms"] += a.elapsed_time(b)


def _timed(name, fn, device):
    if _TIMERS is None or torch.device(device).type != "cuda":
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    rec = _TIMERS.setdefault(name, {"calls": 0, "ms": 0.0, "pending": []})
    rec["pending"].append((e0, e1))
    if len(rec["pending"]) > _TIMER_PENDING_MAX:          # a long run does not grow without bound: the oldest pairs finished long ago
        _fold(rec, _TIMER_PENDING_MAX // 2)
    return out



def shard_bounds(n, world_size, rank):
    """Contiguous near-equal block [lo, hi) of n items for `rank` (the first n % world_size ranks get one extra)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rays(rays, group=None):
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(rays.shape[0], ws, rk)
    return rays[lo:hi].contiguous(), (lo, hi)


class PeerExchange:
    """All-reduce of <= 1024 floats through HIP-IPC peer windows (crnerf_peer_* in include/crnerf.h) instead of RCCL.

    Collective constructor: every rank of `group` (one process per GPU, all on one node, at most 8) creates its window and
    the 64-byte IPC handles are exchanged through the group itself (all_gather_object).  `all_reduce(t)` then sums a
    contiguous float32 device tensor in place on the CURRENT stream with one kernel launch and no host synchronisation;
    every rank must issue the same sequence of calls.  Sums are formed in rank order on every rank (bit-identical
    replicas).  A peer that does not arrive within `timeout_s` (default 60 s: first-call weight packing, uneven shards and a
    slow rank easily skew ranks by seconds, and RCCL would simply have waited) turns the result into NaN; `check()`
    (synchronising) raises and names it -- after that the exchange is out of step and must be rebuilt.  decode_sharded calls
    check() itself unless told not to.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver only supports dmabuf IPC."""

    def __init__(self, group=None, timeout_s=60.0):
        from . import _lib
        self._lib_mod = _lib
        lib = _lib.load()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise ValueError("crnerf_amd: PeerExchange serves the GPUs of one node (world_size <= 8), got %d" % self.world)
        self.timeout_us = int(timeout_s * 1e6)
        self.device = torch.cuda.current_device()
        self._require_peer_access()         # before any window exists: a missing xGMI / PCIe peer path would otherwise surface as a timeout
        own = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        _lib.check(lib.crnerf_peer_window_create(ctypes.byref(own), handle), "crnerf_peer_window_create")
        self._own = own
        handles = [None] * self.world
        dist.all_gather_object(handles, (bytes(handle.raw), os.uname().nodename), group=group)
        if len({h[1] for h in handles}) != 1:
            lib.crnerf_peer_window_destroy(own)
            raise RuntimeError("crnerf_amd: PeerExchange: ranks on different hosts (%s); IPC windows are node-local" % sorted({h[1] for h in handles}))
        self._windows = (ctypes.c_void_p * self.world)()
        self._opened = []
        for r, (h, _) in enumerate(handles):
            if r == self.rank:
                self._windows[r] = own.value
                continue
            w = ctypes.c_void_p()
            _lib.check(lib.crnerf_peer_window_open(h, ctypes.byref(w)), "crnerf_peer_window_open")
            self._windows[r] = w.value
            self._opened.append(w)
        self.epoch = 0
        dist.barrier(group=group)   # every window is open everywhere before the first push

    def _require_peer_access(self):
        """Collective.  Every rank asks the runtime (hipDeviceCanAccessPeer through torch.cuda.can_device_access_peer) whether its GPU can map
        the GPU of every other rank, and ALL ranks raise together when any pair cannot -- instead of the first reduction waiting `timeout_s`
        for a push that can never land.  Peers are identified by PCI address (domain:bus:device), so the check is independent of how each
        process numbers its devices; a peer this process cannot see at all (HIP_VISIBLE_DEVICES isolation) counts as not verifiable and is
        refused as well unless CRNERF_PEER_SKIP_ACCESS_CHECK=1."""
        def pci(i):
            p = torch.cuda.get_device_properties(i)
            return (int(getattr(p, "pci_domain_id", 0)), int(getattr(p, "pci_bus_id", -1)), int(getattr(p, "pci_device_id", -1)))
        mine = pci(self.device)
        local = {pci(i): i for i in range(torch.cuda.device_count())}
        addrs = [None] * self.world
        dist.all_gather_object(addrs, mine, group=self.group)
        problems = []
        if os.environ.get("CRNERF_PEER_SKIP_ACCESS_CHECK") != "1":
            for r, a in enumerate(addrs):
                if r == self.rank or tuple(a) == mine:          # (ranks sharing one GPU: the single-GPU test mode)
                    continue
                idx = local.get(tuple(a))
                if idx is None:
                    problems.append("rank %d cannot see rank %d's GPU %04x:%02x:%02x (device isolation): peer access not verifiable" % ((self.rank, r) + tuple(a)))
                elif not torch.cuda.can_device_access_peer(self.device, idx):
                    problems.append("rank %d (GPU %04x:%02x:%02x) has no peer access to rank %d (GPU %04x:%02x:%02x)" % ((self.rank,) + mine + (r,) + tuple(a)))
        everyone = [None] * self.world
        dist.all_gather_object(everyone, problems, group=self.group)
        flat = [p for ps in everyone for p in ps]
        if flat:
            raise RuntimeError("crnerf_amd: PeerExchange needs hipDeviceCanAccessPeer for every pair of ranks; " + "; ".join(flat) +
                               " -- use the RCCL path (exchange=None)")

    def all_reduce(self, t):
        if self._own is None:
            raise RuntimeError("crnerf_amd: PeerExchange is closed")
        if t.numel() > 1024:
            raise ValueError("crnerf_amd: PeerExchange.all_reduce carries at most 1024 floats, got %d" % t.numel())
        self.epoch = self.epoch + 1 if self.epoch < 0xFFFFFFFE else 1
        lib = self._lib_mod.load()
        self._lib_mod.check(lib.crnerf_peer_allreduce_f32(self._lib_mod.dev_ptr(t, "all_reduce tensor"), t.numel(), self._windows, self.rank, self.world,
                                                          self.epoch, self.timeout_us, self._lib_mod.stream_ptr()), "crnerf_peer_allreduce_f32")
        return t

    def check(self):
        """Synchronises the device; raises if any reduction so far gave up waiting for a peer."""
        st = ctypes.c_int(0)
        self._lib_mod.check(self._lib_mod.load().crnerf_peer_window_status(self._own, ctypes.byref(st)), "crnerf_peer_window_status")
        if st.value:
            raise RuntimeError("crnerf_amd: PeerExchange: rank %d did not arrive within %.1f s at rank %d; results since then are NaN and the "
                               "exchange must be rebuilt" % (st.value - 1, self.timeout_us / 1e6, self.rank))

    def close(self):
        """Collective: no rank may free its window while a peer can still push into it."""
        if self._own is None:
            return
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        lib = self._lib_mod.load()
        for w in self._opened:
            lib.crnerf_peer_window_close(w)
        dist.barrier(group=self.group)
        lib.crnerf_peer_window_destroy(self._own)
        self._own, self._opened = None, []


class _HipKernels:
    def __getattr__(self, name):
        from . import ops
        return getattr(ops, name)


def decode_sharded(net, feature_local, style_feature, group=None, gather=True, kernels=None, equal_shards=False, exchange=None,
                   check_exchange=True):
    """Cross-ray decode of a ray-sharded feature grid.

    net: style_net; feature_local: this rank's [R_local,64] block of feature_fine (pixel-major, rank
    order = pixel order); style_feature: [1,64,h,w], replicated.  Returns RGB planar: [3, R_total] on
    every rank when gather=True (ranks may hold different R_local), else this rank's [3, R_local].
    equal_shards=True promises every rank holds the same R_local: the pixel count is then known on the
    host and the call enqueues without any device->host synchronisation.

    Three kernel phases around two all-reduces (crnerf_crossray_decode_sharded_f32): channel sums ->
    all-reduce(64 floats) -> Gram of the centred conv chain -> all-reduce(1024 floats) -> fc / fold /
    apply on the local pixels -> all-gather of RGB.  exchange: a PeerExchange built on the same group carries the two
    reductions instead of RCCL (default None = RCCL).  check_exchange: with a PeerExchange, exchange.check() -- one device
    synchronisation; a peer that timed out raises there instead of leaving the exchange out of step -- True (default): after every
    call; an int n: after the first and then every n-th call on that exchange (a timed-out reduction hands back NaN images until then --
    for latency-critical inference loops only, never where the image feeds a loss); False: never, the caller checks where it
    synchronises anyway."""
    k = kernels or _HipKernels()
    dev = feature_local.device
    n_local = feature_local.shape[0]
    ws = dist.get_world_size(group)
    if equal_shards:
        count = float(n_local * ws)
    else:
        cnt = torch.tensor([float(n_local)], device=dev)
        _timed("allreduce_pixel_count_1f", lambda: dist.all_reduce(cnt, group=group), dev)
        count = float(cnt.item())
    sp = style_feature.permute(0, 2, 3, 1).reshape(-1, style_feature.shape[1]).contiguous()
    weights = net.decoder_tensors()
    xchg = torch.zeros(64 + 1024, dtype=torch.float32, device=dev)
    k.crossray_decode_sharded(feature_local, sp, weights, 0, xchg, count)
    reduce = exchange.all_reduce if exchange is not None else (lambda t: dist.all_reduce(t, group=group))
    _timed("allreduce_channel_sums_64f", lambda: reduce(xchg[:64]), dev)     # reduction 1: channel sums -> global mean  (linearStyleTransfer.py:59-65)
    k.crossray_decode_sharded(feature_local, sp, weights, 1, xchg, count)
    _timed("allreduce_gram_1024f", lambda: reduce(xchg[64:]), dev)          # reduction 2: Gram of the centred conv chain (linearStyleTransfer.py:29-34)
    rgb_local = k.crossray_decode_sharded(feature_local, sp, weights, 2, xchg, count)
    if rgb_local is None:
        rgb_local = torch.zeros(3, 0, device=dev)
    if exchange is not None and check_exchange:
        n = exchange.__dict__["_decode_calls"] = exchange.__dict__.get("_decode_calls", 0) + 1
        if check_exchange is True or n == 1 or n % int(check_exchange) == 0:
            exchange.check()
    if not gather:
        return rgb_local
    if equal_shards:
        full = torch.empty(ws * 3, n_local, device=dev)
        _timed("allgather_rgb_12B_per_pixel", lambda: dist.all_gather_into_tensor(full, rgb_local.contiguous(), group=group), dev)
        return full.view(ws, 3, n_local).permute(1, 0, 2).reshape(3, ws * n_local)
    sizes = [torch.zeros(1, dtype=torch.long, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.long, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    pad = torch.zeros(3, width, device=dev)
    pad[:, :n_local] = rgb_local
    parts = [torch.empty(3, width, device=dev) for _ in range(ws)]
    _timed("allgather_rgb_12B_per_pixel", lambda: dist.all_gather(parts, pad, group=group), dev)
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
    _timed("allreduce_gradients_flat_%dMB" % round(flat.numel() * 4 / 2 ** 20), lambda: dist.all_reduce(flat, group=group), flat.device)
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
        _timed("allgather_feature_rows", lambda: dist.all_gather(parts, mine, group=group), mine.device)
        return torch.cat([p[:shard_bounds(n_total, ws, r)[1] - shard_bounds(n_total, ws, r)[0]] for r, p in enumerate(parts)])

    @staticmethod
    def backward(ctx, g):
        lo, hi = ctx.block
        return g[lo:hi], None, None


def gather_rays(x, n_total, group=None):
    return GatherRays.apply(x, n_total, group)


ENCODER_BAND_HALO = 12     # rows: the seven layers reach 10 rows beyond a pixel (3x3, 3x3, pool, 3x3, 3x3, pool, 3x3); a multiple of 4 keeps the pools aligned


def encoder_band_plan(H, W, world_size, rank):
    """(H, row0, rows, o0, o1, world_size) for `rank`'s band of an H x W image, or None when the image does not split evenly: the ranks own H /
    world_size rows each (a multiple of 4: two 2 x 2 max-pools) and 32 / world_size rows of the 32 x 32 style grid, whose pooling windows then lie
    inside the owned rows; the band adds ENCODER_BAND_HALO rows on every side that is cut inside the image."""
    if world_size < 2 or 32 % world_size or H % world_size or (H // world_size) % 4 or H // world_size < 8 or W < 8:
        return None
    per = H // world_size
    r0, r1 = rank * per, (rank + 1) * per
    row0, end = max(0, r0 - ENCODER_BAND_HALO), min(H, r1 + ENCODER_BAND_HALO)
    return (H, row0, end - row0, rank * (32 // world_size), (rank + 1) * (32 // world_size), world_size)


def encode_banded(enc, image, group=None):
    """enc(image) for a replicated [1,3,H,W] image in grad mode, the work split into row bands over the ranks of `group` (autograd.BandEncoderFn);
    None when the image does not split (the caller then runs the replicated pass).  Every rank must call it for the same image."""
    if not (torch.is_grad_enabled() and image.dim() == 4 and image.shape[0] == 1 and image.shape[1] == 3):
        return None
    plan = encoder_band_plan(int(image.shape[2]), int(image.shape[3]), dist.get_world_size(group), dist.get_rank(group))
    if plan is None:
        return None
    from .autograd import BandEncoderFn
    convs = (enc.conv1, enc.conv2, enc.conv3, enc.conv4, enc.conv5, enc.conv6, enc.conv7)
    grid = BandEncoderFn.apply(image.to(torch.float32), plan, group, *[t for c in convs for t in (c.weight, c.bias)])
    return grid.view(1, 32, 32, 64).permute(0, 3, 1, 2)


def decode_sharded_train(net, feature_local, style_feature, n_total, group=None, content_only=False):
    """style_net.forward for training on a ray-sharded feature grid (autograd.DecodeShardedFn / ContentDecodeShardedFn): feature_local [n_local,64]
    (this rank's block, every rank the same n_local), style_feature [1,64,h,w] replicated (ignored with content_only) -> RGB planar [3, n_total]
    on every rank, differentiable; the feature gradient stays local."""
    from .autograd import ContentDecodeShardedFn, DecodeShardedFn
    xp = feature_local.contiguous()
    if content_only:
        w, b = net.decoder.rgb_tensors()
        return ContentDecodeShardedFn.apply(xp, w, b, group, net.decoder_tensors())
    sp = style_feature.permute(0, 2, 3, 1).reshape(-1, style_feature.shape[1]).contiguous()
    return DecodeShardedFn.apply(xp, sp, group, int(n_total), *net.decoder_tensors())


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
