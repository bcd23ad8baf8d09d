"""The optimiser step of the training loop as one HIP launch.

The reference builds `torch.optim.Adam(get_parameters(models), lr=hparams.lr, eps=1e-8, weight_decay=hparams.weight_decay)`
(utils/__init__.py:10-33) and Lightning steps it once per batch (train_mask_grid_sample.py:249-252).  At the reference's
1,024-ray batch the step is launch- and host-bound, and torch's multi-tensor Adam costs seven launches plus ~0.5 ms of host
work for the ~190 tensors of the five trained modules.  `FlatAdam` moves the parameters (and keeps both moments) in flat fp32
buffers that share offsets -- `p.data` becomes a view, nothing else about the module changes -- and updates them with
`crnerf_adam_step_f32`: one launch per <= 448 tensors, gradients read where autograd left them.

Scope: the optimiser STEP is on the training step's critical path (SURVEY 8f N4 / configs[3]); the reference's factories around it
(`get_optimizer` / `get_scheduler` / `get_learning_rate`, utils/__init__.py:24-67) are control plane -- SURVEY section 2 marks them out of scope --
and are not mirrored: `FlatAdam(get_parameters(models), lr=hparams.lr, eps=1e-8, weight_decay=hparams.weight_decay)` is the reference's
'adam' branch, and any torch.optim.lr_scheduler drives it like any other optimiser.

Same update as `torch.optim.Adam` (amsgrad / maximize / capturable off): tests/test_gpu_optim.py steps both on the same
gradients.  A parameter whose `.grad` is None is skipped; its step count is the group's (torch keeps one per parameter --
they only differ for a parameter that misses steps, which no module of this pipeline does; `load_state_dict` therefore REFUSES an
optimiser state whose parameters of one group hold different step counts instead of resuming it with the wrong bias correction).
"""
import ctypes
import math

import torch

from . import _lib

__all__ = ["FlatAdam", "get_parameters"]

_ALIGN = 64            # elements: every tensor starts on a 256-byte boundary of the flat buffers
_BLOCK = 4096          # elements one workgroup updates (csrc/kernels.h ADAM_BLOCK_ELEMS)


class _Chunk:
    """<= crnerf_adam_max_tensors() parameters: their block table on the device and a reusable pointer array."""

    def __init__(self, params, offsets, device):
        self.params = params
        rows = []
        for t, (p, off) in enumerate(zip(params, offsets)):
            n = p.numel()
            for c in range(0, n, _BLOCK):
                rows.append((t, off + c, min(_BLOCK, n - c), c))
        self.n_blocks = len(rows)
        self.blocks = torch.tensor(rows, dtype=torch.int32).reshape(-1, 4).to(device)
        self.grads = (ctypes.c_void_p * len(params))()


class _FlatGroup:
    def __init__(self, params, max_tensors):
        if not params:
            raise ValueError("crnerf_amd.FlatAdam: a parameter group is empty")
        dev = params[0].device
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or p.device != dev:
                raise TypeError("crnerf_amd.FlatAdam: parameters must be fp32 tensors of one GPU (got %s on %s); there is no CPU path" % (p.dtype, p.device))
        if len({id(p) for p in params}) != len(params):
            raise ValueError("crnerf_amd.FlatAdam: a parameter appears more than once in a group (torch.optim.Optimizer warns about the same)")
        self.params = list(params)
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += -(-p.numel() // _ALIGN) * _ALIGN
        if total >= 2 ** 31:
            raise ValueError("crnerf_amd.FlatAdam: %d elements do not fit the kernel's 32-bit offsets" % total)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        with torch.no_grad():
            for p, off in zip(params, self.offsets):
                view = self.flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.detach())
                p.data = view
        self.chunks = [_Chunk(self.params[i:i + max_tensors], self.offsets[i:i + max_tensors], dev) for i in range(0, len(params), max_tensors)]
        self.t = 0
        self.step_tensor = torch.zeros((), dtype=torch.float32)       # what state_dict() reports as every parameter's 'step'

    def moments(self, i):
        p, off = self.params[i], self.offsets[i]
        return self.exp_avg[off:off + p.numel()].view(p.shape), self.exp_avg_sq[off:off + p.numel()].view(p.shape)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) with the update in one HIP launch (see the module docstring).

    Construct it AFTER the modules are on the GPU and BEFORE anything captures `p.data` (it re-points every parameter into
    a flat buffer).  Learning-rate schedulers work as with any optimiser (`param_groups[i]['lr']` is read at every step);
    `state_dict()` / `load_state_dict()` use torch.optim.Adam's layout, so a reference checkpoint's optimiser state resumes."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("crnerf_amd.FlatAdam: invalid hyper-parameters lr=%r betas=%r eps=%r weight_decay=%r" % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        max_tensors = int(_lib.load().crnerf_adam_max_tensors())
        self._groups = [_FlatGroup(g["params"], max_tensors) for g in self.param_groups]
        self._point_state_at_the_flat_buffers()

    def _point_state_at_the_flat_buffers(self):
        for fg in self._groups:
            for i, p in enumerate(fg.params):
                m, v = fg.moments(i)
                self.state[p] = {"step": fg.step_tensor, "exp_avg": m, "exp_avg_sq": v}

    def state_dict(self):
        sd = super().state_dict()
        # every parameter gets its own 'step' tensor, as torch.optim.Adam keeps them (it increments each one: a shared tensor would count n times)
        sd["state"] = {k: dict(v, step=v["step"].clone()) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)            # torch re-creates the moments as free-standing tensors: take them back into the flat buffers
        with torch.no_grad():
            for fg in self._groups:
                steps = set()
                for i, p in enumerate(fg.params):
                    st = self.state.get(p)
                    if not st:
                        continue
                    m, v = fg.moments(i)
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    steps.add(int(float(st["step"])))
                if len(steps) > 1:
                    raise ValueError("crnerf_amd.FlatAdam: the loaded state holds different step counts inside one group (%s); "
                                     "this optimiser keeps one per group" % sorted(steps))
                fg.t = steps.pop() if steps else 0
                fg.step_tensor.fill_(fg.t)
        self._point_state_at_the_flat_buffers()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, stream = _lib.load(), _lib.stream_ptr()
        for group, fg in zip(self.param_groups, self._groups):
            if fg.flat.device.index != torch.cuda.current_device():      # stream_ptr() is the CURRENT device's stream
                raise RuntimeError("crnerf_amd.FlatAdam: the parameters live on %s but the current device is cuda:%d; step() enqueues on the "
                                   "current device's stream -- wrap the call in torch.cuda.device(...)" % (fg.flat.device, torch.cuda.current_device()))
            fg.t += 1
            beta1, beta2 = group["betas"]
            step_size = group["lr"] / (1.0 - beta1 ** fg.t)
            bias_correction2_sqrt = math.sqrt(1.0 - beta2 ** fg.t)
            base = fg.flat.data_ptr()
            keep = []                                   # contiguous copies of strided gradients live until the launch is enqueued
            i = 0
            for chunk in fg.chunks:
                arr = chunk.grads
                for k, p in enumerate(chunk.params):
                    if p.data_ptr() != base + 4 * fg.offsets[i]:
                        raise RuntimeError("crnerf_amd.FlatAdam: a parameter no longer lives in the optimiser's flat buffer (its .data was "
                                           "replaced after the optimiser was built -- module.to(), p.data = ...); build the optimiser last")
                    i += 1
                    g = p.grad
                    if g is None:
                        arr[k] = None
                        continue
                    if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                        raise TypeError("crnerf_amd.FlatAdam: gradients must be dense fp32 on the parameter's GPU")
                    if not g.is_contiguous():
                        g = g.contiguous()
                        keep.append(g)
                    arr[k] = g.data_ptr()
                _lib.check(lib.crnerf_adam_step_f32(fg.flat.data_ptr(), fg.exp_avg.data_ptr(), fg.exp_avg_sq.data_ptr(), chunk.blocks.data_ptr(),
                                                    chunk.n_blocks, arr, len(chunk.params), step_size, beta1, beta2, group["eps"],
                                                    group["weight_decay"], bias_correction2_sqrt, stream), "crnerf_adam_step_f32")
            fg.step_tensor.fill_(fg.t)
            torch.autograd.graph.increment_version(fg.params)      # the kernel wrote through raw pointers: caches keyed on p._version must see it
        return loss


def get_parameters(models):
    """utils/__init__.py:10-22: every parameter of a module, a list of modules or a dict of modules."""
    if isinstance(models, (list, tuple)):
        return [p for m in models for p in get_parameters(m)]
    if isinstance(models, dict):
        return [p for m in models.values() for p in get_parameters(m)]
    return list(models.parameters())
