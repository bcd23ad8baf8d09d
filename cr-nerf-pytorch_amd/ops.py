"""Tensor-level wrappers over the C ABI: allocate outputs with torch, enqueue the HIP kernels on the
current stream.  Every function here runs on the GPU or raises."""
import ctypes
import os

import torch

from . import _lib

MLP_TENSOR_NAMES = (
    [n for i in range(1, 9) for n in ("xyz_encoding_%d.0.weight" % i, "xyz_encoding_%d.0.bias" % i)]
    + ["xyz_encoding_final.weight", "xyz_encoding_final.bias", "static_sigma.0.weight", "static_sigma.0.bias",
       "dir_encoding.0.weight", "dir_encoding.0.bias", "static_rgb.0.weight", "static_rgb.0.bias"])
MLP_TENSOR_SHAPES = (
    [(256, 93), (256,)] + [(256, 256), (256,)] * 3 + [(256, 349), (256,)] + [(256, 256), (256,)] * 3
    + [(256, 256), (256,), (1, 256), (1,), (128, 283), (128,), (64, 128), (64,)])


def _wlist(weights, name):
    """The parameter tensors of a call as they are (fp32, contiguous: checked; no detach -- they are only asked for their address and their shape,
    and a training step passes ~450 of them)."""
    return [t if (t.dtype == torch.float32 and t.is_contiguous()) else _f32c(t.detach(), name) for t in weights]


def _slots_valid(slots):
    for d, name, q in slots:
        if d.get(name) is not q:
            return False
    return True


def cached_params(module):
    """tuple(module.parameters()) looked up once: nn.Module.parameters() walks the module tree on every call, and the grad-mode
    checks of the shim modules ran it ~6,000 generator steps per training step (1-2 ms of host time at the reference's 1,024-ray
    batch, where the step is host-bound).  The cache remembers WHERE each Parameter lives (the owning sub-module's _parameters
    dict and its name) and every lookup re-checks those slots by identity, so anything that swaps a Parameter object --
    load_state_dict(assign=True), `layer.weight = nn.Parameter(...)`, to_empty() / meta materialisation,
    torch.__future__.set_overwrite_module_params_on_conversion -- rebuilds it instead of leaving stale tensors behind."""
    c = module.__dict__.get("_crnerf_pcache")
    if c is None or not _slots_valid(c[0]):
        slots = tuple((m._parameters, name, q) for m in module.modules() for name, q in m._parameters.items() if q is not None)
        c = (slots, tuple(s[2] for s in slots))
        module.__dict__["_crnerf_pcache"] = c
    return c[1]


_LIST_CACHE = os.environ.get("CRNERF_LIST_CACHE", "1") != "0"      # 0: measurement switch (A/B of the host time)


def cached_list(module, key, build):
    """A list of a module's Parameter objects (or sub-modules) in kernel order, looked up once per module and generation of the parameter slots
    (cached_params' identity check: a swapped Parameter rebuilds it).  The training step asked for these lists ~20 times per step through 570
    attribute lookups (round 6: the 1,024-ray step is host-bound).  Parameters and modules only -- a cached VIEW would be one autograd node
    shared by every use in a step, and the uses' gradients would then be summed in the engine's order instead of use by use."""
    if not _LIST_CACHE:
        return build()
    cached_params(module)
    tag = module.__dict__["_crnerf_pcache"]
    store = module.__dict__.setdefault("_crnerf_lists", {})
    c = store.get(key)
    if c is None or c[0] is not tag:
        c = (tag, build())
        store[key] = c
    return c[1]


def any_requires_grad(module):
    for q in cached_params(module):
        if q.requires_grad:
            return True
    return False


def mlp_params(module):
    """The 24 NeRF_sigma parameters in MLP_TENSOR_NAMES order; cached with the same slot-identity check as cached_params."""
    c = module.__dict__.get("_crnerf_mlp_params")
    if c is None or not _slots_valid(c[0]):
        slots = []
        for n in MLP_TENSOR_NAMES:
            path, _, leaf = n.rpartition(".")
            owner = module.get_submodule(path)
            slots.append((owner._parameters, leaf, owner._parameters[leaf]))
        c = (tuple(slots), tuple(s[2] for s in slots))
        module.__dict__["_crnerf_mlp_params"] = c
    return c[1]


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    if not t.is_cuda:
        raise RuntimeError("crnerf_amd: %s is on %s; the HIP path needs GPU tensors and has no CPU fallback" % (name, t.device))
    return t


class AutoPack:
    """The packs of precision="auto": `h2` (None when crnerf_pack_mlp_weights_h2 refused the weights: one of them is >= 255 or not finite) and
    `x3`, the scale-free fallback (always there)."""
    __slots__ = ("h2", "x3")

    def __init__(self, h2, x3):
        self.h2, self.x3 = h2, x3


def pack_mlp_weights_auto(state, check=True):
    """check=False (training: a pack per optimiser step): crnerf_pack_mlp_weights_h2_async -- no wait for the stream; a refused pack carries its verdict
    as a flag word the h2 kernels read on the device (they then leave everything to the f32x3 repair), so `h2` is never None."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    h2 = torch.empty(lib.crnerf_packed_mlp_h2_bytes(), dtype=torch.uint8, device=tensors[0].device)
    if not check:
        _lib.check(lib.crnerf_pack_mlp_weights_h2_async(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(h2.data_ptr()), _lib.stream_ptr()),
                   "crnerf_pack_mlp_weights_h2_async")
        return AutoPack(h2, pack_mlp_weights_x3(state))
    rc = lib.crnerf_pack_mlp_weights_h2(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(h2.data_ptr()), _lib.stream_ptr())
    if rc == _lib.ERR_RANGE:
        h2 = None
    else:
        _lib.check(rc, "crnerf_pack_mlp_weights_h2")
    return AutoPack(h2, pack_mlp_weights_x3(state))


def pack_h2_in_range(packed_h2):
    """True when the h2 / transposed-h2 pack carries no range flag (crnerf_pack_h2_status; waits for the stream)."""
    lib = _lib.load()
    rc = lib.crnerf_pack_h2_status(ctypes.c_void_p(packed_h2.data_ptr()), _lib.stream_ptr())
    if rc == _lib.ERR_RANGE:
        return False
    _lib.check(rc, "crnerf_pack_h2_status")
    return True


def pack_mlp_weights(state, out=None, precision="f32"):
    """state: mapping name -> device tensor with the 24 NeRF_sigma tensors (models/nerf.py:137-154).
    precision "f32" -> buffer for the *_f32 entry points, "bf16" -> for the *_bf16 ones, "f32x3" / "f32h2" -> for the *_f32x3 / *_f32h2 ones
    (pack_mlp_weights_x3 / pack_mlp_weights_h2; different layouts each), "auto" -> an AutoPack (h2 + x3)."""
    if _is_auto(precision):
        if out is not None:
            raise ValueError("crnerf_amd: out= is for the f32 / bf16 packs")
        return pack_mlp_weights_auto(state)
    if _is_h2(precision) or _is_x3(precision):
        if out is not None:
            raise ValueError("crnerf_amd: out= is for the f32 / bf16 packs")
        return pack_mlp_weights_h2(state) if _is_h2(precision) else pack_mlp_weights_x3(state)
    lib = _lib.load()
    tensors = []
    for name, shape in zip(MLP_TENSOR_NAMES, MLP_TENSOR_SHAPES):
        t = state[name]
        if tuple(t.shape) != shape:
            raise ValueError("crnerf_amd: %s has shape %s, the HIP kernels are built for %s "
                             "(D=8, W=256, N_emb_xyz=15, N_emb_dir=4, nerf_out_dim=64)" % (name, tuple(t.shape), shape))
        tensors.append(_f32c(t.detach(), name))
    bf16 = _is_bf16(precision)
    nbytes = lib.crnerf_packed_mlp_bf16_bytes() if bf16 else lib.crnerf_packed_mlp_bytes()
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=tensors[0].device)
    arr = _lib.ptr_array(tensors, "mlp tensor")
    fn = lib.crnerf_pack_mlp_weights_bf16 if bf16 else lib.crnerf_pack_mlp_weights
    _lib.check(fn(arr, ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()), "crnerf_pack_mlp_weights" + ("_bf16" if bf16 else ""))
    return out


def _is_auto(precision):
    return precision in ("auto", "f32auto")


def _is_x3(precision):
    return precision in ("f32x3", "x3")


def _is_h2(precision):
    return precision in ("f32h2", "h2")


def _is_bf16(precision):
    if precision in ("bf16", "bfloat16", torch.bfloat16):
        return True
    if precision in ("f32", "fp32", "float32", torch.float32, None) or _is_x3(precision) or _is_h2(precision) or _is_auto(precision):
        return False
    raise ValueError("crnerf_amd: precision must be 'f32', 'bf16', 'f32x3', 'f32h2' or 'auto', got %r" % (precision,))


def _mlp_tensor_list(state):
    tensors = []
    for name, shape in zip(MLP_TENSOR_NAMES, MLP_TENSOR_SHAPES):
        t = state[name]
        if tuple(t.shape) != shape:
            raise ValueError("crnerf_amd: %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
        tensors.append(t if (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda) else _f32c(t.detach(), name))
    return tensors


def pack_mlp_weights_x3(state):
    """Packed weights for the "f32x3" entry points (include/crnerf.h): every weight as three bf16 pieces, k-step-major fragment stream."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    out = torch.empty(lib.crnerf_packed_mlp_x3_bytes(), dtype=torch.uint8, device=tensors[0].device)
    _lib.check(lib.crnerf_pack_mlp_weights_x3(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()),
               "crnerf_pack_mlp_weights_x3")
    return out


def mlp_forward_x3(packed_x3, x, sigma_only=False):
    """NeRF_sigma.forward in fp32 on the bf16 matrix cores (three-piece splits, six MFMAs per product; crnerf_mlp_forward_f32x3)."""
    lib = _lib.load()
    x = _f32c(x, "x")
    want = 93 if sigma_only else 120
    if x.dim() != 2 or x.shape[1] != want:
        raise ValueError("mlp_forward_x3 expects [n,%d], got %s" % (want, tuple(x.shape)))
    if packed_x3.numel() != lib.crnerf_packed_mlp_x3_bytes():
        raise ValueError("crnerf_amd: packed weights are not an x3 pack (pack_mlp_weights_x3)")
    out = torch.empty(x.shape[0], 1 if sigma_only else 65, dtype=torch.float32, device=x.device)
    _lib.check(lib.crnerf_mlp_forward_f32x3(ctypes.c_void_p(packed_x3.data_ptr()), _lib.dev_ptr(x), _lib.dev_ptr(out), x.shape[0], int(bool(sigma_only)),
                                            _lib.stream_ptr()), "crnerf_mlp_forward_f32x3")
    return out


def pack_mlp_weights_h2(state):
    """Packed weights for the "f32h2" entry points (include/crnerf.h): every weight, scaled by 2^8, as two fp16 pieces."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    out = torch.empty(lib.crnerf_packed_mlp_h2_bytes(), dtype=torch.uint8, device=tensors[0].device)
    _lib.check(lib.crnerf_pack_mlp_weights_h2(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()),
               "crnerf_pack_mlp_weights_h2")
    return out


def mlp_forward_h2(packed_h2, x, sigma_only=False):
    """NeRF_sigma.forward in fp32 on the fp16 matrix cores (two-piece splits, three MFMAs per product; crnerf_mlp_forward_f32h2)."""
    lib = _lib.load()
    x = _f32c(x, "x")
    want = 93 if sigma_only else 120
    if x.dim() != 2 or x.shape[1] != want:
        raise ValueError("mlp_forward_h2 expects [n,%d], got %s" % (want, tuple(x.shape)))
    if packed_h2.numel() != lib.crnerf_packed_mlp_h2_bytes():
        raise ValueError("crnerf_amd: packed weights are not an h2 pack (pack_mlp_weights_h2)")
    out = torch.empty(x.shape[0], 1 if sigma_only else 65, dtype=torch.float32, device=x.device)
    _lib.check(lib.crnerf_mlp_forward_f32h2(ctypes.c_void_p(packed_h2.data_ptr()), _lib.dev_ptr(x), _lib.dev_ptr(out), x.shape[0], int(bool(sigma_only)),
                                            _lib.stream_ptr()), "crnerf_mlp_forward_f32h2")
    return out


def render_rays_x3(packed_coarse, packed_fine, rays, n_samples, n_importance, **kw):
    """render_rays(..., precision="f32x3")."""
    return render_rays(packed_coarse, packed_fine, rays, n_samples, n_importance, precision="f32x3", **kw)


def pack_mlp_weights_t(state):
    """Transposed fragment stream for the backward-data kernel."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    out = torch.empty(lib.crnerf_packed_mlp_t_bytes(), dtype=torch.uint8, device=tensors[0].device)
    _lib.check(lib.crnerf_pack_mlp_weights_t(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()),
               "crnerf_pack_mlp_weights_t")
    return out


def mlp_forward_train(packed, x):
    """Forward that keeps the layer activations.  Returns (out[n,65], acts)."""
    lib = _lib.load()
    x = _f32c(x, "x")
    n = x.shape[0]
    out = torch.empty(n, 65, dtype=torch.float32, device=x.device)
    acts = torch.empty(lib.crnerf_mlp_train_acts_bytes(n), dtype=torch.uint8, device=x.device)
    _lib.check(lib.crnerf_mlp_forward_train_f32(ctypes.c_void_p(packed.data_ptr()), _lib.dev_ptr(x), _lib.dev_ptr(out),
                                                ctypes.c_void_p(acts.data_ptr()), n, _lib.stream_ptr()), "crnerf_mlp_forward_train_f32")
    return out, acts


def pack_mlp_weights_t_x3(state):
    """Transposed x3 fragment stream for the backward-data kernel on the x3 core (crnerf_mlp_backward_x3_f32)."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    out = torch.empty(lib.crnerf_packed_mlp_t_x3_bytes(), dtype=torch.uint8, device=tensors[0].device)
    _lib.check(lib.crnerf_pack_mlp_weights_t_x3(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()),
               "crnerf_pack_mlp_weights_t_x3")
    return out


def pack_mlp_weights_t_h2(state):
    """Transposed h2 fragment stream for the backward-data kernel on the h2 core (crnerf_mlp_backward_h2_f32): two fp16 pieces of 2^8 w.  No range
    check of its own -- pack_mlp_weights_h2 / pack_mlp_weights(..., precision="auto") of the same weights is the check."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    out = torch.empty(lib.crnerf_packed_mlp_t_h2_bytes(), dtype=torch.uint8, device=tensors[0].device)
    _lib.check(lib.crnerf_pack_mlp_weights_t_h2(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()),
               "crnerf_pack_mlp_weights_t_h2")
    return out


BWD_PHASE_DGRAD, BWD_PHASE_WGRAD = 8, 16      # CRNERF_BWD_PHASE_* (include/crnerf.h)


def mlp_backward(packed_t, x, out, d_out, acts, wgrad_bf16=False, dgrad_x3=False, dgrad_h2=False, fallback_t_x3=None, phase=None, scratch=None, grads=None):
    """Gradients of sum(out * d_out) w.r.t. the 24 tensors, in MLP_TENSOR_NAMES order.  wgrad_bf16: True / 1 = CRNERF_BWD_WGRAD_BF16
    (include/crnerf.h) -- the weight gradients of every Linear except static_sigma from bf16-rounded operands; 2 / "x3" =
    CRNERF_BWD_WGRAD_BF16X3 -- fp32-accurate weight gradients of the 256 x 256 blocks from three-piece bf16 splits on the bf16 matrix
    cores; 3 / "f16x2" (dgrad_h2 only) = CRNERF_BWD_WGRAD_F16X2 -- the same with the full 256 x 256 blocks from two-piece fp16 splits (three
    piece products, the h2 core's arithmetic; bf16x3 takes over by itself where an operand leaves fp16's range).  Data gradients as the dgrad_*
    arguments say; biases and everything else exact fp32.
    phase: None = the whole backward; "dgrad" = CRNERF_BWD_PHASE_DGRAD (returns the scratch tensor that holds the deltas; x may be None);
    "wgrad" = CRNERF_BWD_PHASE_WGRAD on `scratch` (the tensor a "dgrad" call returned; packed_t / out / d_out may be None) -- the two halves may
    be enqueued on different streams, ordered by the caller (autograd.FusedRenderFn)."""
    lib = _lib.load()
    do_d, do_w = phase != "wgrad", phase != "dgrad"
    if phase not in (None, "dgrad", "wgrad"):
        raise ValueError("crnerf_amd: mlp_backward phase is None, 'dgrad' or 'wgrad', got %r" % (phase,))
    x = _f32c(x, "x") if (do_w or x is not None) else None
    out, d_out = (_f32c(out, "out"), _f32c(d_out, "d_out")) if do_d else (None, None)
    n = x.shape[0] if x is not None else out.shape[0]
    dev = x.device if x is not None else out.device
    f16x2 = wgrad_bf16 in (3, "f16x2", "h2")
    if f16x2 and not dgrad_h2:
        raise ValueError("crnerf_amd: f16x2 weight gradients take their delta ranges from the h2 data gradient (dgrad_h2=True)")
    if do_w and grads is None:
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in MLP_TENSOR_SHAPES]
    need = lib.crnerf_mlp_train_scratch_bytes(n)
    if scratch is None:
        if phase == "wgrad":
            raise ValueError("crnerf_amd: mlp_backward(phase='wgrad') needs the scratch of the 'dgrad' call")
        scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    elif scratch.numel() < need or not scratch.is_contiguous() or scratch.dtype != torch.uint8:
        raise ValueError("crnerf_amd: mlp_backward scratch must be a contiguous uint8 tensor of >= %d bytes" % need)
    # dgrad_x3 / dgrad_h2: the data gradient on the x3 / h2 core; packed_t is then a pack_mlp_weights_t_x3 / pack_mlp_weights_t_h2 pack
    fn, name, want = ((lib.crnerf_mlp_backward_h2_f32, "crnerf_mlp_backward_h2_f32", lib.crnerf_packed_mlp_t_h2_bytes()) if dgrad_h2 else
                      (lib.crnerf_mlp_backward_x3_f32, "crnerf_mlp_backward_x3_f32", lib.crnerf_packed_mlp_t_x3_bytes()) if dgrad_x3 else
                      (lib.crnerf_mlp_backward_ex_f32, "crnerf_mlp_backward_ex_f32", lib.crnerf_packed_mlp_t_bytes()))
    if do_d and packed_t.numel() != want:
        raise ValueError("crnerf_amd: %s needs a %d-byte transposed pack, got %d" % (name, want, packed_t.numel()))
    if fallback_t_x3 is not None and (not dgrad_h2 or fallback_t_x3.numel() != lib.crnerf_packed_mlp_t_x3_bytes()):
        raise ValueError("crnerf_amd: fallback_t_x3 is the pack_mlp_weights_t_x3 safety net of dgrad_h2")
    nul = ctypes.c_void_p(None)
    head = (ctypes.c_void_p(packed_t.data_ptr()) if do_d else nul,) + ((ctypes.c_void_p(fallback_t_x3.data_ptr() if (fallback_t_x3 is not None and do_d) else None),) if dgrad_h2 else ())
    flags = 4 if f16x2 else (2 if wgrad_bf16 in (2, "x3", "bf16x3") else (1 if wgrad_bf16 else 0))
    flags |= BWD_PHASE_DGRAD if phase == "dgrad" else (BWD_PHASE_WGRAD if phase == "wgrad" else 0)
    _lib.check(fn(*head, _lib.dev_ptr(x) if x is not None else nul, _lib.dev_ptr(out) if do_d else nul, _lib.dev_ptr(d_out) if do_d else nul,
                  ctypes.c_void_p(acts.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                  _lib.ptr_array(grads, "grad") if do_w else None, n, flags, _lib.stream_ptr()), name)
    return scratch if phase == "dgrad" else grads


def pack_mlp_weights_mixed(state):
    """bf16 B-operand fragment streams of every nn.Linear (forward and transposed) for the mixed-precision training twins
    (crnerf_mlp_*_mixed_f32, include/crnerf.h).  Returns (packed, tensors): the twins also read the fp32 biases / sigma head."""
    lib = _lib.load()
    tensors = _mlp_tensor_list(state)
    out = torch.empty(lib.crnerf_packed_mlp_mixed_bytes(), dtype=torch.uint8, device=tensors[0].device)
    _lib.check(lib.crnerf_pack_mlp_weights_mixed(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()),
               "crnerf_pack_mlp_weights_mixed")
    return out, tensors


def mlp_forward_train_mixed(packed_mixed, tensors, x):
    """Mixed-precision forward that keeps the layer activations (bf16 rows, kernel-internal order).  Returns (out[n,65], acts)."""
    lib = _lib.load()
    x = _f32c(x, "x")
    n = x.shape[0]
    out = torch.empty(n, 65, dtype=torch.float32, device=x.device)
    acts = torch.empty(lib.crnerf_mlp_train_mixed_acts_bytes(n), dtype=torch.uint8, device=x.device)
    _lib.check(lib.crnerf_mlp_forward_train_mixed_f32(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(packed_mixed.data_ptr()), _lib.dev_ptr(x),
                                                      _lib.dev_ptr(out), ctypes.c_void_p(acts.data_ptr()), n, _lib.stream_ptr()),
               "crnerf_mlp_forward_train_mixed_f32")
    return out, acts


def mlp_backward_mixed(packed_mixed, tensors, x, out, d_out, acts, fused_acts=False):
    """Gradients of sum(out * d_out) w.r.t. the 24 tensors through the mixed-precision twins (MLP_TENSOR_NAMES order).
    fused_acts: `acts` was written by the fused renderer's training twin (render_rays(..., precision="bf16", train=True)); x is unused then."""
    lib = _lib.load()
    out, d_out = _f32c(out, "out"), _f32c(d_out, "d_out")
    n = out.shape[0]
    grads = [torch.empty(s, dtype=torch.float32, device=out.device) for s in MLP_TENSOR_SHAPES]
    scratch = torch.empty(lib.crnerf_mlp_train_mixed_scratch_bytes(n), dtype=torch.uint8, device=out.device)
    if fused_acts:
        _lib.check(lib.crnerf_mlp_backward_mixed_ex_f32(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(packed_mixed.data_ptr()), _lib.dev_ptr(out),
                                                        _lib.dev_ptr(d_out), ctypes.c_void_p(acts.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                                                        _lib.ptr_array(grads, "grad"), n, 1, _lib.stream_ptr()), "crnerf_mlp_backward_mixed_ex_f32")
        return grads
    x = _f32c(x, "x")
    _lib.check(lib.crnerf_mlp_backward_mixed_f32(_lib.ptr_array(tensors, "mlp tensor"), ctypes.c_void_p(packed_mixed.data_ptr()), _lib.dev_ptr(x),
                                                 _lib.dev_ptr(out), _lib.dev_ptr(d_out), ctypes.c_void_p(acts.data_ptr()),
                                                 ctypes.c_void_p(scratch.data_ptr()), _lib.ptr_array(grads, "grad"), n, _lib.stream_ptr()),
               "crnerf_mlp_backward_mixed_f32")
    return grads


def posenc(x, n_freqs):
    lib = _lib.load()
    x = _f32c(x, "x")
    if x.dim() != 2 or x.shape[1] != 3:
        raise ValueError("posenc expects [n,3], got %s" % (tuple(x.shape),))
    out = torch.empty(x.shape[0], 6 * n_freqs + 3, dtype=torch.float32, device=x.device)
    _lib.check(lib.crnerf_posenc_f32(_lib.dev_ptr(x), _lib.dev_ptr(out), x.shape[0], n_freqs, _lib.stream_ptr()), "crnerf_posenc_f32")
    return out


def embed_points(rays, z, dir_emb):
    """x[R*N,120] = cat(PosEmbedding_xyz(o + d z), dir_emb repeated) in one pass (rendering.py:100-114); dir_emb = posenc(dirs, 4)."""
    lib = _lib.load()
    rays, z, dir_emb = _f32c(rays, "rays"), _f32c(z, "z"), _f32c(dir_emb, "dir_emb")
    R, N = z.shape
    if tuple(rays.shape) != (R, 8) or tuple(dir_emb.shape) != (R, 27):
        raise ValueError("embed_points expects rays [R,8], z [R,N], dir_emb [R,27]; got %s %s %s" % (tuple(rays.shape), tuple(z.shape), tuple(dir_emb.shape)))
    x = torch.empty(R * N, 120, dtype=torch.float32, device=z.device)
    _lib.check(lib.crnerf_embed_points_f32(_lib.dev_ptr(rays), _lib.dev_ptr(z), _lib.dev_ptr(dir_emb), _lib.dev_ptr(x), R, N, _lib.stream_ptr()),
               "crnerf_embed_points_f32")
    return x


def mlp_forward_auto(pack, x, sigma_only=False):
    """precision="auto": the h2 core, then crnerf_mlp_forward_f32x3_repair over the same output -- the 128-point groups in which a point left
    fp16's range are evaluated again on the scale-free x3 core (nothing else is touched; no host round trip).  A refused h2 pack: x3 throughout."""
    if pack.h2 is None:
        return mlp_forward_x3(pack.x3, x, sigma_only=sigma_only)
    lib = _lib.load()
    x = _f32c(x, "x")
    out = mlp_forward_h2(pack.h2, x, sigma_only=sigma_only)
    _lib.check(lib.crnerf_mlp_forward_f32x3_repair(ctypes.c_void_p(pack.x3.data_ptr()), _lib.dev_ptr(x), _lib.dev_ptr(out), x.shape[0], int(bool(sigma_only)),
                                                   _lib.stream_ptr()), "crnerf_mlp_forward_f32x3_repair")
    return out


def mlp_forward(packed, x, sigma_only=False, precision="f32"):
    if _is_auto(precision):
        return mlp_forward_auto(packed, x, sigma_only=sigma_only)
    if _is_h2(precision):
        return mlp_forward_h2(packed, x, sigma_only=sigma_only)
    if _is_x3(precision):
        return mlp_forward_x3(packed, x, sigma_only=sigma_only)
    lib = _lib.load()
    x = _f32c(x, "x")
    want = 93 if sigma_only else 120
    if x.dim() != 2 or x.shape[1] != want:
        raise ValueError("mlp_forward expects [n,%d], got %s" % (want, tuple(x.shape)))
    out = torch.empty(x.shape[0], 1 if sigma_only else 65, dtype=torch.float32, device=x.device)
    bf16 = _is_bf16(precision)
    _check_packed(packed, bf16)
    fn = lib.crnerf_mlp_forward_bf16 if bf16 else lib.crnerf_mlp_forward_f32
    _lib.check(fn(ctypes.c_void_p(packed.data_ptr()), _lib.dev_ptr(x), _lib.dev_ptr(out), x.shape[0], int(bool(sigma_only)), _lib.stream_ptr()),
               "crnerf_mlp_forward_bf16" if bf16 else "crnerf_mlp_forward_f32")
    return out


def _check_packed(packed, bf16):
    """The two packed layouts differ in size, so a mix-up is caught here instead of rendering garbage."""
    lib = _lib.load()
    want = lib.crnerf_packed_mlp_bf16_bytes() if bf16 else lib.crnerf_packed_mlp_bytes()
    if packed is not None and packed.numel() * packed.element_size() != want:
        raise ValueError("crnerf_amd: packed weights are %d bytes, the %s entry points need %d (pack with precision=%r)"
                         % (packed.numel() * packed.element_size(), "bf16" if bf16 else "f32", want, "bf16" if bf16 else "f32"))


def composite(raw, z, noise=None, noise_std=0.0):
    lib = _lib.load()
    raw, z = _f32c(raw, "raw"), _f32c(z, "z")
    R, N = z.shape
    if tuple(raw.shape) != (R, N, 65):
        raise ValueError("composite expects raw [R,N,65], got %s" % (tuple(raw.shape),))
    if noise is not None:
        noise = _f32c(noise, "noise")
    w = torch.empty(R, N, dtype=torch.float32, device=z.device)
    feat = torch.empty(R, 64, dtype=torch.float32, device=z.device)
    depth = torch.empty(R, dtype=torch.float32, device=z.device)
    _lib.check(lib.crnerf_composite_f32(_lib.dev_ptr(raw), _lib.dev_ptr(z), _lib.dev_ptr(noise), float(noise_std), _lib.dev_ptr(w),
                                        _lib.dev_ptr(feat), _lib.dev_ptr(depth), R, N, _lib.stream_ptr()), "crnerf_composite_f32")
    return w, feat, depth


def composite_backward(raw, z, d_feature, d_depth=None, d_weights=None, noise=None, noise_std=0.0):
    lib = _lib.load()
    raw, z, d_feature = _f32c(raw, "raw"), _f32c(z, "z"), _f32c(d_feature, "d_feature")
    R, N = z.shape
    opt = [None if t is None else _f32c(t, n) for t, n in ((d_depth, "d_depth"), (d_weights, "d_weights"), (noise, "noise"))]
    d_raw = torch.empty(R, N, 65, dtype=torch.float32, device=z.device)
    _lib.check(lib.crnerf_composite_backward_f32(_lib.dev_ptr(raw), _lib.dev_ptr(z), _lib.dev_ptr(opt[2]), float(noise_std), _lib.dev_ptr(d_feature),
                                                 _lib.dev_ptr(opt[0]), _lib.dev_ptr(opt[1]), _lib.dev_ptr(d_raw), R, N, _lib.stream_ptr()),
               "crnerf_composite_backward_f32")
    return d_raw


def sample_pdf_merge(z_coarse, weights_coarse, n_importance, u=None, return_samples=False):
    lib = _lib.load()
    z_coarse, weights_coarse = _f32c(z_coarse, "z_coarse"), _f32c(weights_coarse, "weights_coarse")
    R, Nc = z_coarse.shape
    u_stride = 0
    if u is not None:
        u = _f32c(u, "u")
        u_stride = 0 if u.dim() == 1 else n_importance
    zs = torch.empty(R, Nc + n_importance, dtype=torch.float32, device=z_coarse.device)
    smp = torch.empty(R, n_importance, dtype=torch.float32, device=z_coarse.device) if return_samples else None
    _lib.check(lib.crnerf_sample_pdf_merge_f32(_lib.dev_ptr(z_coarse), _lib.dev_ptr(weights_coarse), _lib.dev_ptr(u), u_stride, _lib.dev_ptr(zs),
                                               _lib.dev_ptr(smp), R, Nc, n_importance, _lib.stream_ptr()), "crnerf_sample_pdf_merge_f32")
    return (zs, smp) if return_samples else zs


def render_rays(packed_coarse, packed_fine, rays, n_samples, n_importance, use_disp=False, view_dir=None, z_coarse=None, z_steps=None, u=None,
                noise_coarse=None, noise_fine=None, noise_std=0.0, want_z_fine=False, precision="f32", train=False, launcher=False, rng=None):
    """Fused renderer.  Returns a dict of freshly allocated tensors.  train=True: the training twin
    crnerf_render_rays_train_f32 -- the dict additionally holds what the backward needs: z_coarse (as used), z_fine,
    acts_coarse / acts_fine (crnerf_mlp_forward_train_f32 layout, point = ray * N + sample) and raw_coarse / raw_fine [R,N,65].
    train=True with precision="bf16": crnerf_render_rays_train_bf16, the twin of the opt-in mixed-precision mode (acts_* in the layout
    mlp_backward_mixed(..., fused_acts=True) reads).
    precision="auto": packs from pack_mlp_weights(..., precision="auto"); the h2 core, repaired by the x3 core where it poisoned a ray
    (train=True: crnerf_render_rays_train_f32h2, then crnerf_render_rays_train_f32x3_repair -- saved rows included).
    rng (fp32, f32x3, f32h2, auto): {"seed": int, "ray_offset": int, "perturb": float, "jitter": bool, "u": bool, "noise": bool} -- the stochastic
    steps of rendering.py:125 / :169-176 / :30 drawn INSIDE the kernel (include/crnerf.h CRNERF_RNG_*, csrc/philox.h) instead of
    handed over as tensors; the dict then also holds what was drawn: "z_coarse_used" [R,Nc], and with noise "noise_coarse_used" /
    "noise_fine_used" (standard normal, before noise_std).  rng_fill() returns the same draws as tensors."""
    lib = _lib.load()
    repair = None
    if _is_auto(precision):
        # the h2 core with the x3 core as its safety net: render on h2, then crnerf_render_rays_f32x3_repair re-renders the ray quads that came out NaN
        # (a point's activations left fp16's range).  A refused h2 pack (a weight >= 255): the x3 core throughout.
        packs = [pk for pk in (packed_coarse, packed_fine) if pk is not None]
        if any(not isinstance(pk, AutoPack) for pk in packs):
            raise ValueError("crnerf_amd: precision='auto' needs packs from pack_mlp_weights(..., precision='auto')")
        if any(pk.h2 is None for pk in packs):
            precision = "f32x3"
            packed_coarse, packed_fine = packed_coarse.x3, (packed_fine.x3 if packed_fine is not None else None)
        else:
            precision = "f32h2"
            repair = (packed_coarse.x3, packed_fine.x3 if packed_fine is not None else None)
            packed_coarse, packed_fine = packed_coarse.h2, (packed_fine.h2 if packed_fine is not None else None)
    h2 = _is_h2(precision)                       # fp32 on the fp16 matrix cores (crnerf_render_rays_f32h2; packs from pack_mlp_weights_h2)
    x3 = _is_x3(precision)                       # fp32 on the bf16 matrix cores (crnerf_render_rays_f32x3; packs from pack_mlp_weights_x3)
    bf16 = False if (x3 or h2) else _is_bf16(precision)
    want_z_fine = want_z_fine or train
    if h2:
        for pk in (packed_coarse, packed_fine):
            if pk is not None and pk.numel() != lib.crnerf_packed_mlp_h2_bytes():
                raise ValueError("crnerf_amd: precision='f32h2' needs packs from pack_mlp_weights_h2")
    elif x3:
        for pk in (packed_coarse, packed_fine):
            if pk is not None and pk.numel() != lib.crnerf_packed_mlp_x3_bytes():
                raise ValueError("crnerf_amd: precision='f32x3' needs packs from pack_mlp_weights_x3")
    else:
        _check_packed(packed_coarse, bf16)
        _check_packed(packed_fine, bf16)
    rays = _f32c(rays, "rays")
    if rays.dim() != 2 or rays.shape[1] != 8:
        raise ValueError("rays must be [R,8], got %s" % (tuple(rays.shape),))
    R, dev = rays.shape[0], rays.device
    Nc, Ni = int(n_samples), int(n_importance)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    out = {"weights_coarse": new(R, Nc), "feature_coarse": new(R, 64), "depth_coarse": new(R)}
    if Ni > 0:
        out.update({"weights_fine": new(R, Nc + Ni), "feature_fine": new(R, 64), "depth_fine": new(R)})
        if want_z_fine:
            out["z_fine"] = new(R, Nc + Ni)
    if R == 0:
        return out
    keep = [t if t is None else _f32c(t, n) for t, n in ((view_dir, "view_dir"), (z_coarse, "z_coarse"), (z_steps, "z_steps"), (u, "u"),
                                                         (noise_coarse, "noise_coarse"), (noise_fine, "noise_fine"))]
    a = _lib.RenderArgs()
    a.packed_coarse = packed_coarse.data_ptr()
    a.packed_fine = packed_fine.data_ptr() if packed_fine is not None else None
    a.rays = rays.data_ptr()
    a.u_stride = 0 if (u is None or u.dim() == 1) else Ni
    for field, t in zip(("view_dir", "z_coarse", "z_steps", "u", "noise_coarse", "noise_fine"), keep):
        setattr(a, field, t.data_ptr() if t is not None else None)
    a.noise_std = float(noise_std)
    a.use_disp = int(bool(use_disp))
    a.n_rays, a.n_samples, a.n_importance = R, Nc, Ni
    for k in ("weights_coarse", "feature_coarse", "depth_coarse", "weights_fine", "feature_fine", "depth_fine", "z_fine"):
        setattr(a, k, out[k].data_ptr() if k in out else None)
    if rng is not None:
        if bf16:
            raise ValueError("crnerf_amd: in-kernel random draws exist in the fp32 kernels only")
        flags = (_lib.RNG_JITTER if rng.get("jitter") else 0) | (_lib.RNG_U if (rng.get("u") and Ni > 0) else 0) | (_lib.RNG_NOISE if rng.get("noise") else 0)
        a.rng_seed, a.rng_ray_offset, a.rng_flags, a.perturb = int(rng["seed"]) & (2 ** 64 - 1), int(rng.get("ray_offset", 0)), flags, float(rng.get("perturb", 1.0))
        out["z_coarse_used"] = new(R, Nc)
        a.z_coarse_out = out["z_coarse_used"].data_ptr()
        if flags & _lib.RNG_NOISE:
            out["noise_coarse_used"] = new(R, Nc)
            a.noise_coarse_out = out["noise_coarse_used"].data_ptr()
            if Ni > 0:
                out["noise_fine_used"] = new(R, Nc + Ni)
                a.noise_fine_out = out["noise_fine_used"].data_ptr()
    if launcher:      # measurement helper: re-launch the same call on the same buffers with nothing but the C call on the host side
        if train:
            raise ValueError("crnerf_amd: launcher=True is for the inference entry points")
        fn = lib.crnerf_render_rays_f32h2 if h2 else (lib.crnerf_render_rays_f32x3 if x3 else (lib.crnerf_render_rays_bf16 if bf16 else lib.crnerf_render_rays_f32))
        name = "crnerf_render_rays_f32h2" if h2 else ("crnerf_render_rays_f32x3" if x3 else ("crnerf_render_rays_bf16" if bf16 else "crnerf_render_rays_f32"))
        held = (keep, rays, packed_coarse, packed_fine, out, repair)  # the argument struct holds raw pointers: keep EVERY tensor behind them alive
        # (the outputs too: a caller that drops `out` must not hand their memory back to the caching allocator while launch() can still write it)
        a2 = _repair_args(a, repair)

        def launch(_held=held):
            _lib.check(fn(ctypes.byref(a), _lib.stream_ptr()), name)
            if a2 is not None:
                _lib.check(lib.crnerf_render_rays_f32x3_repair(ctypes.byref(a2), _lib.stream_ptr()), "crnerf_render_rays_f32x3_repair")
        return launch, out
    if train:
        Nf = Nc + Ni
        acts_bytes = lib.crnerf_mlp_train_mixed_acts_bytes if bf16 else lib.crnerf_mlp_train_acts_bytes
        out["acts_coarse"] = torch.empty(acts_bytes(R * Nc), dtype=torch.uint8, device=dev)
        out["raw_coarse"] = new(R, Nc, 65)
        if Ni > 0:
            out["acts_fine"] = torch.empty(acts_bytes(R * Nf), dtype=torch.uint8, device=dev)
            out["raw_fine"] = new(R, Nf, 65)
        vp = lambda k: ctypes.c_void_p(out[k].data_ptr()) if k in out else None  # noqa: E731
        fn, name = ((lib.crnerf_render_rays_train_f32h2, "crnerf_render_rays_train_f32h2") if h2 else
                    (lib.crnerf_render_rays_train_f32x3, "crnerf_render_rays_train_f32x3") if x3 else
                    (lib.crnerf_render_rays_train_bf16, "crnerf_render_rays_train_bf16") if bf16 else
                    (lib.crnerf_render_rays_train_f32, "crnerf_render_rays_train_f32"))
        _lib.check(fn(ctypes.byref(a), vp("acts_coarse"), vp("acts_fine"), vp("raw_coarse"), vp("raw_fine"), _lib.stream_ptr()), name)
        a2 = _repair_args(a, repair)
        if a2 is not None:        # precision="auto": the ray quads the h2 twin poisoned, once more on the scale-free core -- outputs AND saved state
            _lib.check(lib.crnerf_render_rays_train_f32x3_repair(ctypes.byref(a2), vp("acts_coarse"), vp("acts_fine"), vp("raw_coarse"), vp("raw_fine"),
                                                                 _lib.stream_ptr()), "crnerf_render_rays_train_f32x3_repair")
        return out
    if h2:
        _lib.check(lib.crnerf_render_rays_f32h2(ctypes.byref(a), _lib.stream_ptr()), "crnerf_render_rays_f32h2")
        a2 = _repair_args(a, repair)
        if a2 is not None:
            _lib.check(lib.crnerf_render_rays_f32x3_repair(ctypes.byref(a2), _lib.stream_ptr()), "crnerf_render_rays_f32x3_repair")
        return out
    fn = lib.crnerf_render_rays_f32x3 if x3 else (lib.crnerf_render_rays_bf16 if bf16 else lib.crnerf_render_rays_f32)
    _lib.check(fn(ctypes.byref(a), _lib.stream_ptr()), "crnerf_render_rays_f32x3" if x3 else ("crnerf_render_rays_bf16" if bf16 else "crnerf_render_rays_f32"))
    return out


def render_rays_bf16_fine(packed_fine_bf16, rays, weights_coarse, n_samples, n_importance, use_disp=False, view_dir=None, z_coarse=None, z_steps=None, u=None,
                          noise_fine=None, noise_std=0.0, want_z_fine=False):
    """crnerf_render_rays_bf16_fine: sample_pdf + z merge on `weights_coarse` [R,Nc] (rendered by another core, e.g. render_rays(..., n_importance=0,
    precision="auto")) and the fine model on the bf16 matrix cores, fused -- {"weights_fine", "feature_fine", "depth_fine"[, "z_fine"]}."""
    lib = _lib.load()
    _check_packed(packed_fine_bf16, True)
    rays, weights_coarse = _f32c(rays, "rays"), _f32c(weights_coarse, "weights_coarse")
    R, dev = rays.shape[0], rays.device
    Nc, Ni = int(n_samples), int(n_importance)
    if rays.dim() != 2 or rays.shape[1] != 8 or tuple(weights_coarse.shape) != (R, Nc) or Ni <= 0:
        raise ValueError("render_rays_bf16_fine: rays [R,8], weights_coarse [R,%d] and n_importance > 0 expected, got %s / %s / %d"
                         % (Nc, tuple(rays.shape), tuple(weights_coarse.shape), Ni))
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    out = {"weights_fine": new(R, Nc + Ni), "feature_fine": new(R, 64), "depth_fine": new(R)}
    if want_z_fine:
        out["z_fine"] = new(R, Nc + Ni)
    if R == 0:
        return out
    keep = [t if t is None else _f32c(t, n) for t, n in ((view_dir, "view_dir"), (z_coarse, "z_coarse"), (z_steps, "z_steps"), (u, "u"), (noise_fine, "noise_fine"))]
    a = _lib.RenderArgs()
    a.packed_fine = packed_fine_bf16.data_ptr()
    a.rays = rays.data_ptr()
    a.u_stride = 0 if (u is None or u.dim() == 1) else Ni
    for field, t in zip(("view_dir", "z_coarse", "z_steps", "u", "noise_fine"), keep):
        setattr(a, field, t.data_ptr() if t is not None else None)
    a.noise_std = float(noise_std)
    a.use_disp = int(bool(use_disp))
    a.n_rays, a.n_samples, a.n_importance = R, Nc, Ni
    a.weights_coarse = weights_coarse.data_ptr()                    # INPUT of this entry point
    for k in ("weights_fine", "feature_fine", "depth_fine", "z_fine"):
        setattr(a, k, out[k].data_ptr() if k in out else None)
    _lib.check(lib.crnerf_render_rays_bf16_fine(ctypes.byref(a), _lib.stream_ptr()), "crnerf_render_rays_bf16_fine")
    return out


def _repair_args(a, repair):
    """The argument block of the h2 render with the x3 packs in place of the h2 ones (crnerf_render_rays_f32x3_repair)."""
    if repair is None:
        return None
    a2 = _lib.RenderArgs()
    ctypes.memmove(ctypes.byref(a2), ctypes.byref(a), ctypes.sizeof(a))
    a2.packed_coarse = repair[0].data_ptr()
    a2.packed_fine = repair[1].data_ptr() if repair[1] is not None else None
    return a2


_IN_KERNEL_RNG = [None]


def set_in_kernel_rng(on=True):
    """True (default): grad-mode renders through the fused fp32 kernel draw their stratified jitter / sample_pdf uniforms / density
    noise inside the kernel (Philox keyed on (seed, ray, sample)); False: torch.rand / torch.randn tensors handed to the kernel, the
    round-1/2 path.  CRNERF_IN_KERNEL_RNG=0 does the same from the environment."""
    _IN_KERNEL_RNG[0] = bool(on)


def in_kernel_rng():
    import os
    if _IN_KERNEL_RNG[0] is not None:
        return _IN_KERNEL_RNG[0]
    return os.environ.get("CRNERF_IN_KERNEL_RNG", "1") not in ("0", "")


def rng_fill(n_rays, n, seed, stream, ray_offset=0, device="cuda"):
    """The renderer's in-kernel draws as a tensor [n_rays, n] (crnerf_rng_fill_f32): stream 0 = jitter uniforms, 1 = sample_pdf
    uniforms, 2 = coarse noise, 3 = fine noise.  Feeding these through render_rays' tensor arguments reproduces the in-kernel
    path bit for bit (tests/test_gpu_rng.py)."""
    lib = _lib.load()
    out = torch.empty(int(n_rays), int(n), dtype=torch.float32, device=device)
    _lib.check(lib.crnerf_rng_fill_f32(ctypes.c_void_p(out.data_ptr()), int(n_rays), int(n), int(seed) & (2 ** 64 - 1), int(stream), int(ray_offset),
                                       _lib.stream_ptr()), "crnerf_rng_fill_f32")
    return out


# ---------------------------------------------------------------- cross-ray decoder pieces
_ws_cache = {}


def crossray_workspace(device):
    # one per device AND stream: pipeline.TrainingSystem runs independent decodes side by side on their own streams (round 6)
    key = (device.type, device.index, int(_lib.stream_ptr().value or 0))
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(_lib.load().crnerf_crossray_workspace_bytes(), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def crossray_chansum(x):
    lib = _lib.load()
    x = _f32c(x, "x")
    out = torch.empty(64, dtype=torch.float32, device=x.device)
    ws = crossray_workspace(x.device)
    _lib.check(lib.crnerf_crossray_chansum_f32(_lib.dev_ptr(x), x.shape[0], _lib.dev_ptr(out), ctypes.c_void_p(ws.data_ptr()),
                                               _lib.stream_ptr()), "crnerf_crossray_chansum_f32")
    return out


def crossray_gram(x, mean, cnn):
    lib = _lib.load()
    x, mean = _f32c(x, "x"), _f32c(mean, "mean")
    cnn = [_f32c(t.detach(), "cnn") for t in cnn]
    out = torch.empty(1024, dtype=torch.float32, device=x.device)
    ws = crossray_workspace(x.device)
    _lib.check(lib.crnerf_crossray_gram_f32(_lib.dev_ptr(x), x.shape[0], _lib.dev_ptr(mean), _lib.ptr_array(cnn, "cnn"), _lib.dev_ptr(out),
                                            ctypes.c_void_p(ws.data_ptr()), _lib.stream_ptr()), "crnerf_crossray_gram_f32")
    return out


def crossray_matrix(gram_sum, count, fc_w, fc_b):
    lib = _lib.load()
    fc_w, fc_b = _f32c(fc_w.detach(), "fc_w"), _f32c(fc_b.detach(), "fc_b")
    out = torch.empty(1024, dtype=torch.float32, device=gram_sum.device)
    _lib.check(lib.crnerf_crossray_matrix_f32(_lib.dev_ptr(gram_sum), float(count), _lib.dev_ptr(fc_w), _lib.dev_ptr(fc_b),
                                              _lib.dev_ptr(out), _lib.stream_ptr()), "crnerf_crossray_matrix_f32")
    return out


def crossray_fold(s_matrix, c_matrix, c_mean, s_mean, lin):
    lib = _lib.load()
    lin = [_f32c(t.detach(), "lin") for t in lin]
    out = torch.empty(195, dtype=torch.float32, device=lin[0].device)
    _lib.check(lib.crnerf_crossray_fold_f32(_lib.dev_ptr(s_matrix), _lib.dev_ptr(c_matrix), _lib.dev_ptr(c_mean), _lib.dev_ptr(s_mean),
                                            _lib.ptr_array(lin, "lin"), _lib.dev_ptr(out), _lib.stream_ptr()), "crnerf_crossray_fold_f32")
    return out


def crossray_apply(x, affine, out=None):
    lib = _lib.load()
    x = _f32c(x, "x")
    HW = x.shape[0]
    if out is None:
        out = torch.empty(3, HW, dtype=torch.float32, device=x.device)
    _lib.check(lib.crnerf_crossray_apply_f32(_lib.dev_ptr(x), HW, _lib.dev_ptr(affine), _lib.dev_ptr(out), out.stride(0), _lib.stream_ptr()),
               "crnerf_crossray_apply_f32")
    return out


def crossray_decode(content_pm, style_pm, weights, out=None):
    """Whole style_net.forward from one host call.  content_pm [HW,64], style_pm [HWs,64] or None,
    weights: the 22 parameter tensors in state_dict order.  Returns planar rgb [3,HW]."""
    lib = _lib.load()
    x = _f32c(content_pm, "content")
    s = _f32c(style_pm, "style") if style_pm is not None else None
    ws = crossray_workspace(x.device)
    HW = x.shape[0]
    if out is None:
        out = torch.empty(3, HW, dtype=torch.float32, device=x.device)
    arr = _lib.ptr_array(_wlist(weights, "decoder weight"), "decoder weight")
    _lib.check(lib.crnerf_crossray_decode_f32(_lib.dev_ptr(x), HW, _lib.dev_ptr(s), s.shape[0] if s is not None else 0, arr,
                                              ctypes.c_void_p(ws.data_ptr()), _lib.dev_ptr(out), out.stride(0), _lib.stream_ptr()),
               "crnerf_crossray_decode_f32")
    return out


def encoder_forward(image, weights):
    """encoder_sameoutputsize.forward: image [1,3,H,W] or [3,H,W] -> pixel-major style grid [1024,64]."""
    lib = _lib.load()
    img = _f32c(image.reshape(3, image.shape[-2], image.shape[-1]), "image")
    H, W = img.shape[-2], img.shape[-1]
    ws = torch.empty(lib.crnerf_encoder_workspace_bytes(H, W), dtype=torch.uint8, device=img.device)
    out = torch.empty(1024, 64, dtype=torch.float32, device=img.device)
    arr = _lib.ptr_array(_wlist(weights, "encoder weight"), "encoder weight")
    _lib.check(lib.crnerf_encoder_forward_f32(_lib.dev_ptr(img), H, W, arr, ctypes.c_void_p(ws.data_ptr()), _lib.dev_ptr(out), _lib.stream_ptr()),
               "crnerf_encoder_forward_f32")
    return out


def decoder_content_backward(content_pm, rgb_w, rgb, d_rgb):
    """Backward of the decoder-only ('content') call: returns (d_content[HW,64], d_w[3,64], d_b[3])."""
    lib = _lib.load()
    x, w = _f32c(content_pm, "content"), _f32c(rgb_w.detach(), "rgb_w").reshape(3, 64)
    rgb, d_rgb = _f32c(rgb, "rgb"), _f32c(d_rgb, "d_rgb")
    HW = x.shape[0]
    dx, dw, db = torch.empty_like(x), torch.empty(3, 64, device=x.device), torch.empty(3, device=x.device)
    work = torch.empty(lib.crnerf_decoder_content_backward_workspace_bytes(HW), dtype=torch.uint8, device=x.device)
    _lib.check(lib.crnerf_decoder_content_backward_f32(_lib.dev_ptr(x), HW, _lib.dev_ptr(w), _lib.dev_ptr(rgb), rgb.stride(0), _lib.dev_ptr(d_rgb),
                                                       d_rgb.stride(0), ctypes.c_void_p(work.data_ptr()), _lib.dev_ptr(dx), _lib.dev_ptr(dw),
                                                       _lib.dev_ptr(db), _lib.stream_ptr()), "crnerf_decoder_content_backward_f32")
    return dx, dw, db


def encoder_forward_train(image, weights):
    """Forward of the appearance encoder that keeps the layer outputs: returns (grid [1024,64], saved, (H, W))."""
    lib = _lib.load()
    img = _f32c(image, "image")
    if img.dim() == 4:
        img = img[0]
    _, H, W = img.shape
    ws = _wlist(weights, "encoder weight")
    saved = torch.empty(lib.crnerf_encoder_train_saved_bytes(H, W), dtype=torch.uint8, device=img.device)
    out = torch.empty(1024, 64, dtype=torch.float32, device=img.device)
    _lib.check(lib.crnerf_encoder_forward_train_f32(_lib.dev_ptr(img), H, W, _lib.ptr_array(ws, "encoder weight"), ctypes.c_void_p(saved.data_ptr()),
                                                    _lib.dev_ptr(out), _lib.stream_ptr()), "crnerf_encoder_forward_train_f32")
    return out, saved, (H, W)


def encoder_backward(weights, saved, hw, out, d_out, want_d_image=True):
    lib = _lib.load()
    H, W = hw
    ws = _wlist(weights, "encoder weight")
    grads = [torch.empty_like(t) for t in ws]
    d_img = torch.empty(3, H, W, dtype=torch.float32, device=out.device) if want_d_image else None
    scratch = torch.empty(lib.crnerf_encoder_train_scratch_bytes(H, W), dtype=torch.uint8, device=out.device)
    _lib.check(lib.crnerf_encoder_backward_f32(H, W, _lib.ptr_array(ws, "encoder weight"), ctypes.c_void_p(saved.data_ptr()), _lib.dev_ptr(out),
                                               _lib.dev_ptr(_f32c(d_out, "d_out")), ctypes.c_void_p(scratch.data_ptr()),
                                               _lib.ptr_array(grads, "encoder grad"), _lib.dev_ptr(d_img), _lib.stream_ptr()),
               "crnerf_encoder_backward_f32")
    return grads, d_img


def encoder_forward_train_band(image_rows, h_image, row0, o0, o1, weights):
    """crnerf_encoder_forward_train_band_f32: the training forward of the appearance encoder over rows [row0, row0 + H) of an image of `h_image` rows
    (image_rows [3,H,W] contiguous) -> (rows [o0, o1) of the 32 x 32 grid as [(o1 - o0) * 32, 64], saved, (H, W))."""
    lib = _lib.load()
    img = _f32c(image_rows, "image_rows")
    _, H, W = img.shape
    ws = _wlist(weights, "encoder weight")
    saved = torch.empty(lib.crnerf_encoder_train_band_saved_bytes(H, W, o1 - o0), dtype=torch.uint8, device=img.device)
    out = torch.empty((o1 - o0) * 32, 64, dtype=torch.float32, device=img.device)
    _lib.check(lib.crnerf_encoder_forward_train_band_f32(_lib.dev_ptr(img), H, W, int(h_image), int(row0), int(o0), int(o1), _lib.ptr_array(ws, "encoder weight"),
                                                         ctypes.c_void_p(saved.data_ptr()), _lib.dev_ptr(out), _lib.stream_ptr()), "crnerf_encoder_forward_train_band_f32")
    return out, saved, (H, W)


def encoder_backward_band(weights, saved, hw, h_image, row0, o0, o1, out, d_out, want_d_image=True):
    """crnerf_encoder_backward_band_f32 -> (this band's part of the 14 weight gradients, d_image_rows [3,H,W] or None)."""
    lib = _lib.load()
    H, W = hw
    ws = _wlist(weights, "encoder weight")
    grads = [torch.empty_like(t) for t in ws]
    d_img = torch.empty(3, H, W, dtype=torch.float32, device=out.device) if want_d_image else None
    scratch = torch.empty(lib.crnerf_encoder_train_band_scratch_bytes(H, W, o1 - o0), dtype=torch.uint8, device=out.device)
    _lib.check(lib.crnerf_encoder_backward_band_f32(H, W, int(h_image), int(row0), int(o0), int(o1), _lib.ptr_array(ws, "encoder weight"),
                                                    ctypes.c_void_p(saved.data_ptr()), _lib.dev_ptr(out), _lib.dev_ptr(_f32c(d_out, "d_out")),
                                                    ctypes.c_void_p(scratch.data_ptr()), _lib.ptr_array(grads, "encoder grad"), _lib.dev_ptr(d_img), _lib.stream_ptr()),
               "crnerf_encoder_backward_band_f32")
    return grads, d_img


def crossray_decode_backward(content_pm, style_pm, weights, d_rgb):
    """Backward of crossray_decode: returns (d_content[HW,64], d_style[HWs,64], [22 weight gradients])."""
    lib = _lib.load()
    x, s, d_rgb = _f32c(content_pm, "content"), _f32c(style_pm, "style"), _f32c(d_rgb, "d_rgb")
    HW, HWs = x.shape[0], s.shape[0]
    ws_t = _wlist(weights, "decoder weight")
    grads = [torch.empty_like(t) for t in ws_t]
    dx, ds = torch.empty_like(x), torch.empty_like(s)
    work = torch.empty(lib.crnerf_crossray_backward_workspace_bytes(HW, HWs), dtype=torch.uint8, device=x.device)
    _lib.check(lib.crnerf_crossray_decode_backward_f32(_lib.dev_ptr(x), HW, _lib.dev_ptr(s), HWs, _lib.ptr_array(ws_t, "decoder weight"),
                                                       _lib.dev_ptr(d_rgb), d_rgb.stride(0), ctypes.c_void_p(work.data_ptr()), _lib.dev_ptr(dx),
                                                       _lib.dev_ptr(ds), _lib.ptr_array(grads, "decoder grad"), _lib.stream_ptr()),
               "crnerf_crossray_decode_backward_f32")
    return dx, ds, grads


def crossray_decode_sharded(content_pm, style_pm, weights, phase, xchg, count_global, rgb=None):
    """One phase (0, 1, 2) of the ray-sharded decode; the caller all-reduces xchg[0:64] after phase 0 and
    xchg[64:1088] after phase 1.  Returns rgb [3, HW_local] after phase 2."""
    lib = _lib.load()
    s = _f32c(style_pm, "style")
    n = content_pm.shape[0]
    x = _f32c(content_pm, "content") if n else None
    ws = crossray_workspace(s.device)
    arr = _lib.ptr_array(_wlist(weights, "decoder weight"), "decoder weight")
    if phase == 2 and rgb is None:
        rgb = torch.empty(3, n, dtype=torch.float32, device=s.device)
    _lib.check(lib.crnerf_crossray_decode_sharded_f32(_lib.dev_ptr(x), n, _lib.dev_ptr(s), s.shape[0], arr, int(phase), _lib.dev_ptr(xchg),
                                                      float(count_global), ctypes.c_void_p(ws.data_ptr()),
                                                      _lib.dev_ptr(rgb) if (rgb is not None and n) else None,
                                                      rgb.stride(0) if (rgb is not None and n) else 0, _lib.stream_ptr()),
               "crnerf_crossray_decode_sharded_f32")
    return rgb


def crossray_decode_backward_sharded(content_pm, style_pm, weights, d_rgb_local, phase, fwd_xchg, count_global, xb, state=None):
    """One phase (0, 1, 2) of the ray-sharded decoder backward (crnerf_crossray_decode_backward_sharded_f32).  state: None for phase 0 -- the call
    allocates (workspace, d_content, d_style, grads) and returns it; pass it back for phases 1 and 2.  The caller all-reduces xb[0:320] after
    phase 0 and xb[320:384] after phase 1.  After phase 2: d_content [HW_local,64], d_style [HWs,64], grads (22; [8..13] are this rank's part)."""
    lib = _lib.load()
    x, s, d_rgb = _f32c(content_pm, "content"), _f32c(style_pm, "style"), _f32c(d_rgb_local, "d_rgb")
    HW, HWs = x.shape[0], s.shape[0]
    ws_t = _wlist(weights, "decoder weight")
    if state is None:
        state = (torch.empty(lib.crnerf_crossray_backward_workspace_bytes(HW, HWs), dtype=torch.uint8, device=x.device), torch.empty_like(x),
                 torch.empty_like(s), [torch.empty_like(t) for t in ws_t])
    work, dx, ds, grads = state
    _lib.check(lib.crnerf_crossray_decode_backward_sharded_f32(_lib.dev_ptr(x), HW, _lib.dev_ptr(s), HWs, _lib.ptr_array(ws_t, "decoder weight"),
                                                               _lib.dev_ptr(d_rgb), d_rgb.stride(0), ctypes.c_void_p(work.data_ptr()), _lib.dev_ptr(dx),
                                                               _lib.dev_ptr(ds), _lib.ptr_array(grads, "decoder grad"), int(phase), _lib.dev_ptr(fwd_xchg),
                                                               float(count_global), _lib.dev_ptr(xb), _lib.stream_ptr()),
               "crnerf_crossray_decode_backward_sharded_f32")
    return state


# ---------------------------------------------------------------- training-side neighbours
LOSS_KEYS = ("kl_a", "rec_a_random", "c_l", "content_constraint", "r_ms", "r_md", "f_l")


def _rgb2d(t, name):
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError("crnerf_amd: %s must be [R,3], got %s" % (name, tuple(t.shape)))
    if not t.is_cuda:
        raise RuntimeError("crnerf_amd: %s is on %s; the HIP path needs GPU tensors and has no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("crnerf_amd: %s must be float32" % name)
    return t


def _dense_flat(t):
    """1-D view of t's elements in MEMORY order when t is a permutation of a contiguous tensor (non-overlapping, no gaps), else None."""
    if t.is_contiguous():
        return t.reshape(-1)
    expect = 1
    for d in sorted(range(t.dim()), key=lambda d: t.stride(d)):
        if t.shape[d] == 1:
            continue
        if t.stride(d) != expect:
            return None
        expect *= t.shape[d]
    return t.as_strided((t.numel(),), (1,))


def loss_args(rgb_coarse, targets, rgb_fine=None, mask=None, a_embedded=None, a_embedded_random=None, a_embedded_random_rec=None,
              content_wo=None, content_with=None, mse_on_appearance=False, coef=1.0, weight_kl=0.0, weight_rec_a=0.0,
              weight_content=0.0, mask_size_weight=0.0, mask_digit_weight=0.0):
    """Fill struct crnerf_loss_args.  rgb tensors may be any strided [R,3] view (e.g. the decoder's planar output
    rearranged the reference's way); everything else is made contiguous.  Returns (struct, keep-alive list)."""
    a = _lib.LossArgs()
    keep = []
    R = rgb_coarse.shape[0]
    for name, t in (("rgb_coarse", rgb_coarse), ("rgb_fine", rgb_fine), ("targets", targets)):
        if t is None:
            setattr(a, name, None)
            continue
        t = _rgb2d(t.detach(), name)
        if t.shape[0] != R:
            raise ValueError("crnerf_amd: %s has %d rows, rgb_coarse %d" % (name, t.shape[0], R))
        keep.append(t)
        setattr(a, name, t.data_ptr())
        setattr(a, name + "_row_stride", t.stride(0))
        setattr(a, name + "_chan_stride", t.stride(1))
    flat = lambda t, n: None if t is None else _f32c(t.detach(), n).reshape(-1)  # noqa: E731
    # The embedding / content terms are full reductions over one tensor or over a same-shaped pair: the element ORDER does not matter as long
    # as both members of a pair are walked in the same one.  The encoders' outputs are NCHW *views* of pixel-major memory; read in memory order
    # they need no transposing copy here (and their gradients, written in the same order, none in the encoder's backward).
    def memory_order(*ts):
        if any(t is None for t in ts):
            return None
        ts = [t.detach() for t in ts]
        if any(t.dtype != torch.float32 or not t.is_cuda or t.shape != ts[0].shape or t.stride() != ts[0].stride() for t in ts):
            return None
        views = [_dense_flat(t) for t in ts]
        return None if any(v is None for v in views) else views
    layouts = {}
    m = flat(mask, "out_mask")
    got = memory_order(a_embedded)
    ae = got[0] if got else flat(a_embedded, "a_embedded")
    if got and not a_embedded.is_contiguous():
        layouts["a_embedded"] = a_embedded.stride()
    got = memory_order(a_embedded_random, a_embedded_random_rec)
    ar, arr = got if got else (flat(a_embedded_random, "a_embedded_random"), flat(a_embedded_random_rec, "a_embedded_random_rec"))
    if got and not a_embedded_random_rec.is_contiguous():
        layouts["a_embedded_random_rec"] = a_embedded_random_rec.stride()
    got = memory_order(content_wo, content_with)
    cw, cwi = got if got else (flat(content_wo, "content_wo_a_embed"), flat(content_with, "content_with_a_embed"))
    if got and not content_wo.is_contiguous():
        layouts["content_wo_a_embed"] = layouts["content_with_a_embed"] = content_wo.stride()
    if m is not None and m.numel() != R:
        raise ValueError("crnerf_amd: out_mask must have one value per ray")
    if arr is not None and (ar is None or ar.numel() != arr.numel()):
        raise ValueError("crnerf_amd: a_embedded_random and a_embedded_random_rec must have the same size")
    if (cw is None) != (cwi is None) or (cw is not None and cw.numel() != cwi.numel()):
        raise ValueError("crnerf_amd: content_wo_a_embed and content_with_a_embed come as a same-sized pair")
    keep += [t for t in (m, ae, ar, arr, cw, cwi) if t is not None]
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    a.mask, a.a_embedded, a.a_embedded_random, a.a_embedded_random_rec, a.content_wo, a.content_with = p(m), p(ae), p(ar), p(arr), p(cw), p(cwi)
    a.n_a = ae.numel() if ae is not None else 0
    a.n_rec = arr.numel() if arr is not None else 0
    a.n_content = cw.numel() if cw is not None else 0
    a.n_rays = R
    a.mse_on_appearance = int(bool(mse_on_appearance))
    a.coef, a.weight_kl, a.weight_rec_a, a.weight_content = float(coef), float(weight_kl), float(weight_rec_a), float(weight_content)
    a.mask_size_weight, a.mask_digit_weight = float(mask_size_weight), float(mask_digit_weight)
    a._memory_order = layouts       # inputs handed over in memory order (name -> strides): their gradients must be allocated with the same strides
    return a, keep


_loss_ws = {}


def loss_forward(args):
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    if dev not in _loss_ws:
        _loss_ws[dev] = torch.empty(lib.crnerf_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
    losses = torch.empty(7, dtype=torch.float32, device=dev)
    _lib.check(lib.crnerf_loss_f32(ctypes.byref(args), _lib.dev_ptr(losses), ctypes.c_void_p(_loss_ws[dev].data_ptr()), _lib.stream_ptr()),
               "crnerf_loss_f32")
    return losses


def loss_backward(args, upstream, want, strides=None):
    """want: dict name -> shape of the gradients to produce (names of struct crnerf_loss_grads without the d_ prefix); strides: name -> strides
    for the inputs loss_args handed over in memory order (their gradients are written in that order too)."""
    lib = _lib.load()
    g = _lib.LossGrads()
    out = {}
    for name, shape in want.items():
        st = (strides or {}).get(name)
        out[name] = (torch.empty_strided(tuple(shape), tuple(st), dtype=torch.float32, device=upstream.device) if st is not None
                     else torch.empty(shape, dtype=torch.float32, device=upstream.device))
        setattr(g, "d_" + name, out[name].data_ptr())
    _lib.check(lib.crnerf_loss_backward_f32(ctypes.byref(args), _lib.dev_ptr(_f32c(upstream, "upstream")), ctypes.byref(g), _lib.stream_ptr()),
               "crnerf_loss_backward_f32")
    return out


def grid_sample_batch(all_rays, all_rgbs, row_offset, img_w, img_h, side, w_lin, h_lin, scale, h_offset, w_offset):
    lib = _lib.load()
    all_rays, all_rgbs = _f32c(all_rays, "all_rays"), _f32c(all_rgbs, "all_rgbs")
    if all_rays.dim() != 2 or all_rays.shape[1] < 9 or all_rgbs.dim() != 2 or all_rgbs.shape[1] != 3:
        raise ValueError("crnerf_amd: all_rays must be [N,>=9] and all_rgbs [N,3]")
    w_lin, h_lin = _f32c(w_lin, "w_lin"), _f32c(h_lin, "h_lin")
    dev, n = all_rays.device, side * side
    out = {"rays": torch.empty(n, 8, device=dev), "ts": torch.empty(n, dtype=torch.int64, device=dev), "rgbs": torch.empty(n, 3, device=dev),
           "rgb_idx": torch.empty(n, dtype=torch.int64, device=dev), "uv_sample": torch.empty(n, 2, device=dev)}
    a = _lib.BatchArgs()
    a.all_rays, a.ray_stride, a.all_rgbs, a.row_offset = all_rays.data_ptr(), all_rays.stride(0), all_rgbs.data_ptr(), int(row_offset)
    a.img_w, a.img_h, a.side = int(img_w), int(img_h), int(side)
    a.w_lin, a.h_lin = w_lin.data_ptr(), h_lin.data_ptr()
    a.scale, a.h_offset, a.w_offset = float(scale), float(h_offset), float(w_offset)
    a.rays, a.ts, a.rgbs, a.rgb_idx, a.uv_sample = (out[k].data_ptr() for k in ("rays", "ts", "rgbs", "rgb_idx", "uv_sample"))
    _lib.check(lib.crnerf_grid_sample_batch_f32(ctypes.byref(a), _lib.stream_ptr()), "crnerf_grid_sample_batch_f32")
    return out


# ---------------------------------------------------------------- transient-mask network operators (csrc/cgnet.hip)
def _chw(t, name):
    """[1,C,H,W] or [C,H,W] fp32 device tensor -> contiguous, with its (C, H, W)."""
    t = _f32c(t, name)
    if t.dim() == 4:
        if t.shape[0] != 1:
            raise ValueError("crnerf_amd: %s: the mask network runs on one image (batch 1), got batch %d" % (name, t.shape[0]))
    elif t.dim() != 3:
        raise ValueError("crnerf_amd: %s must be [1,C,H,W] or [C,H,W]" % name)
    return t, tuple(t.shape[-3:])


def _geom(x_shape, w, stride, padding, dilation, groups):
    C, H, W = x_shape
    cout, cpg, k, k2 = w.shape
    if k != k2:
        raise ValueError("crnerf_amd: conv2d: square kernels only")
    if groups == 1:
        if cpg != C:
            raise ValueError("crnerf_amd: conv2d: weight expects %d input channels, got %d" % (cpg, C))
    elif not (groups == C == cout and cpg == 1):
        raise ValueError("crnerf_amd: conv2d: groups must be 1 or depth-wise (groups = cin = cout)")
    g = _lib.ConvGeom(C, cout, H, W, k, stride, padding, dilation, int(groups != 1))
    Ho = (H + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    return g, Ho, Wo


def conv2d(x, w, stride=1, padding=0, dilation=1, groups=1):
    """F.conv2d(x, w, None, stride, padding, dilation, groups) for batch 1 (groups = 1 or depth-wise)."""
    lib = _lib.load()
    x, shp = _chw(x, "x")
    w = _f32c(w, "weight")
    g, Ho, Wo = _geom(shp, w, stride, padding, dilation, groups)
    y = torch.empty(1, w.shape[0], Ho, Wo, device=x.device)
    _lib.check(lib.crnerf_conv2d_f32(ctypes.byref(g), _lib.dev_ptr(x), _lib.dev_ptr(w), _lib.dev_ptr(y), _lib.stream_ptr()), "crnerf_conv2d_f32")
    return y


def conv2d_backward(x, w, d_y, stride=1, padding=0, dilation=1, groups=1, want_dx=True):
    lib = _lib.load()
    x, shp = _chw(x, "x")
    w, d_y = _f32c(w, "weight"), _f32c(d_y, "d_y")
    g, _, _ = _geom(shp, w, stride, padding, dilation, groups)
    dx = torch.empty_like(x) if want_dx else None
    dw = torch.empty_like(w)
    _lib.check(lib.crnerf_conv2d_backward_f32(ctypes.byref(g), _lib.dev_ptr(x), _lib.dev_ptr(w), _lib.dev_ptr(d_y), _lib.dev_ptr(dx), _lib.dev_ptr(dw),
                                              _lib.stream_ptr()), "crnerf_conv2d_backward_f32")
    return dx, dw


def bn_prelu(x, gamma, beta, alpha, eps, training, running_mean=None, running_var=None, update=None):
    """BatchNorm2d + PReLU.  Returns (y, mean, invstd, var_unbiased); eval mode reads the running statistics.
    update=(momentum, num_batches_tracked or None) in training mode: the momentum update of running_mean / running_var (and the step
    counter) happens in the SAME launch (crnerf_bn_prelu_train_f32) instead of five element-wise launches per layer."""
    lib = _lib.load()
    x, (C, H, W) = _chw(x, "x")
    y = torch.empty_like(x)
    if training and update is not None:
        mean, invstd, var_u = (torch.empty(C, device=x.device) for _ in range(3))
        momentum, nbt = update
        for t, n in ((running_mean, "running_mean"), (running_var, "running_var")):
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise ValueError("crnerf_amd: %s must be a contiguous float32 GPU buffer" % n)
        _lib.check(lib.crnerf_bn_prelu_train_f32(_lib.dev_ptr(x), _lib.dev_ptr(_f32c(gamma, "bn.weight")), _lib.dev_ptr(_f32c(beta, "bn.bias")),
                                                 _lib.dev_ptr(_f32c(alpha, "act.weight")), _lib.dev_ptr(mean), _lib.dev_ptr(invstd), _lib.dev_ptr(var_u),
                                                 _lib.dev_ptr(y), _lib.dev_ptr(running_mean), _lib.dev_ptr(running_var),
                                                 ctypes.c_void_p(nbt.data_ptr()) if nbt is not None else None, float(momentum), C, H * W, float(eps),
                                                 _lib.stream_ptr()), "crnerf_bn_prelu_train_f32")
        return y, mean, invstd, var_u
    if training:
        mean, invstd, var_u = (torch.empty(C, device=x.device) for _ in range(3))
    else:
        mean, invstd, var_u = _f32c(running_mean, "running_mean"), torch.rsqrt(_f32c(running_var, "running_var") + eps), None
    _lib.check(lib.crnerf_bn_prelu_f32(_lib.dev_ptr(x), _lib.dev_ptr(_f32c(gamma, "bn.weight")), _lib.dev_ptr(_f32c(beta, "bn.bias")),
                                       _lib.dev_ptr(_f32c(alpha, "act.weight")), _lib.dev_ptr(mean), _lib.dev_ptr(invstd), _lib.dev_ptr(var_u),
                                       _lib.dev_ptr(y), C, H * W, float(eps), int(bool(training)), _lib.stream_ptr()), "crnerf_bn_prelu_f32")
    return y, mean, invstd, var_u


def bn_prelu_backward(x, gamma, beta, alpha, mean, invstd, d_y, training):
    lib = _lib.load()
    x, (C, H, W) = _chw(x, "x")
    d_y = _f32c(d_y, "d_y")
    dx = torch.empty_like(x)
    dg, db, da = (torch.empty(C, device=x.device) for _ in range(3))
    _lib.check(lib.crnerf_bn_prelu_backward_f32(_lib.dev_ptr(x), _lib.dev_ptr(_f32c(gamma, "bn.weight")), _lib.dev_ptr(_f32c(beta, "bn.bias")),
                                                _lib.dev_ptr(_f32c(alpha, "act.weight")), _lib.dev_ptr(mean), _lib.dev_ptr(invstd), _lib.dev_ptr(d_y),
                                                _lib.dev_ptr(dx), _lib.dev_ptr(dg), _lib.dev_ptr(db), _lib.dev_ptr(da), C, H * W, int(bool(training)),
                                                _lib.stream_ptr()), "crnerf_bn_prelu_backward_f32")
    return dx, dg, db, da


def avgpool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1)."""
    lib = _lib.load()
    x, (C, H, W) = _chw(x, "x")
    y = torch.empty(1, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=x.device)
    _lib.check(lib.crnerf_avgpool3s2_f32(_lib.dev_ptr(x), _lib.dev_ptr(y), C, H, W, 0, _lib.stream_ptr()), "crnerf_avgpool3s2_f32")
    return y


def avgpool3s2_backward(d_y, in_shape):
    lib = _lib.load()
    d_y = _f32c(d_y, "d_y")
    C, H, W = in_shape
    dx = torch.empty(1, C, H, W, device=d_y.device)
    _lib.check(lib.crnerf_avgpool3s2_f32(_lib.dev_ptr(d_y), _lib.dev_ptr(dx), C, H, W, 1, _lib.stream_ptr()), "crnerf_avgpool3s2_f32")
    return dx


def fglo(x, w1, b1, w2, b2):
    """FGlo: x * sigmoid(fc2(relu(fc1(mean_hw(x))))).  Returns (y, stats) -- stats is what the backward needs."""
    lib = _lib.load()
    x, (C, H, W) = _chw(x, "x")
    R = w1.shape[0]
    y, stats = torch.empty_like(x), torch.empty(2 * C + R, device=x.device)
    _lib.check(lib.crnerf_fglo_f32(_lib.dev_ptr(x), _lib.dev_ptr(_f32c(w1, "fc.0.weight")), _lib.dev_ptr(_f32c(b1, "fc.0.bias")),
                                   _lib.dev_ptr(_f32c(w2, "fc.2.weight")), _lib.dev_ptr(_f32c(b2, "fc.2.bias")), _lib.dev_ptr(stats), _lib.dev_ptr(y),
                                   C, R, H * W, _lib.stream_ptr()), "crnerf_fglo_f32")
    return y, stats


def fglo_backward(x, w1, w2, stats, d_y):
    lib = _lib.load()
    x, (C, H, W) = _chw(x, "x")
    d_y = _f32c(d_y, "d_y")
    R = w1.shape[0]
    dev = x.device
    dx, dw1, db1, dw2, db2 = torch.empty_like(x), torch.empty(R, C, device=dev), torch.empty(R, device=dev), torch.empty(C, R, device=dev), \
        torch.empty(C, device=dev)
    scratch = torch.empty(2 * C, device=dev)
    _lib.check(lib.crnerf_fglo_backward_f32(_lib.dev_ptr(x), _lib.dev_ptr(_f32c(w1, "fc.0.weight")), _lib.dev_ptr(_f32c(w2, "fc.2.weight")),
                                            _lib.dev_ptr(stats), _lib.dev_ptr(d_y), _lib.dev_ptr(scratch), _lib.dev_ptr(dx), _lib.dev_ptr(dw1),
                                            _lib.dev_ptr(db1), _lib.dev_ptr(dw2), _lib.dev_ptr(db2), C, R, H * W, _lib.stream_ptr()),
               "crnerf_fglo_backward_f32")
    return dx, dw1, db1, dw2, db2


def bilinear_gather(x, size, idx=None, sigmoid=False):
    """F.interpolate(x[1,1,h,w], size, mode='bilinear', align_corners=False) (then sigmoid), at pixels idx (None: all, as [1,1,Ho,Wo])."""
    lib = _lib.load()
    x, (C, h, w) = _chw(x, "x")
    if C != 1:
        raise ValueError("crnerf_amd: bilinear_gather: one channel expected (the mask), got %d" % C)
    Ho, Wo = int(size[0]), int(size[1])
    if idx is None:
        out, n = torch.empty(1, 1, Ho, Wo, device=x.device), Ho * Wo
    else:
        idx = idx.contiguous()
        out, n = torch.empty(idx.numel(), device=x.device), idx.numel()
    _lib.check(lib.crnerf_bilinear_gather_f32(_lib.dev_ptr(x), h, w, Ho, Wo, _lib.dev_ptr(idx, "idx", torch.int64), n, int(bool(sigmoid)),
                                              _lib.dev_ptr(out), _lib.stream_ptr()), "crnerf_bilinear_gather_f32")
    return out


def bilinear_gather_backward(out, d_out, in_hw, size, idx=None, sigmoid=False):
    lib = _lib.load()
    h, w = in_hw
    Ho, Wo = int(size[0]), int(size[1])
    d_out = _f32c(d_out, "d_out")
    n = d_out.numel()
    d_in = torch.empty(1, 1, h, w, device=d_out.device)
    _lib.check(lib.crnerf_bilinear_gather_backward_f32(_lib.dev_ptr(out) if sigmoid else None, _lib.dev_ptr(d_out), h, w, Ho, Wo,
                                                       _lib.dev_ptr(idx.contiguous(), "idx", torch.int64) if idx is not None else None, n,
                                                       int(bool(sigmoid)), _lib.dev_ptr(d_in), _lib.stream_ptr()), "crnerf_bilinear_gather_backward_f32")
    return d_in


# ---------------------------------------------------------------- the mask network of a training step as two calls (csrc/cgnet_chain.hip)
def cgnet_forward_train(image, params, running_mean, running_var, num_batches_tracked, momentum, eps):
    """Context_Guided_Network(classes=1, M=2, N=2).forward in train mode (lightweight_seg.py:274-368): image[1,C,H,W] -> (mask[1,1,H,W], saved).
    params: the 76 tensors in state_dict order; running_mean / running_var / num_batches_tracked: the 14 BatchNorm layers' buffers (updated)."""
    lib = _lib.load()
    image, (C, H, W) = _chw(image, "image")
    if len(params) != lib.crnerf_cgnet_param_count() or len(running_mean) != lib.crnerf_cgnet_bn_count():
        raise ValueError("crnerf_amd: cgnet_forward_train takes %d parameters and %d BatchNorm layers" % (lib.crnerf_cgnet_param_count(), lib.crnerf_cgnet_bn_count()))
    saved = torch.empty(lib.crnerf_cgnet_arena_bytes(C, H, W) // 4, device=image.device)
    mask = torch.empty(1, 1, H, W, device=image.device)
    nbt = (ctypes.c_void_p * len(running_mean))()
    for i, t in enumerate(num_batches_tracked):
        if t is not None:
            if not t.is_cuda or t.dtype != torch.int64:
                raise ValueError("crnerf_amd: num_batches_tracked must be an int64 GPU buffer")
            nbt[i] = t.data_ptr()
    _lib.check(lib.crnerf_cgnet_forward_train_f32(_lib.dev_ptr(image), C, H, W, _lib.ptr_array(params, "params"), _lib.ptr_array(running_mean, "running_mean"),
                                                  _lib.ptr_array(running_var, "running_var"), nbt, float(momentum), float(eps), _lib.dev_ptr(saved),
                                                  _lib.dev_ptr(mask), _lib.stream_ptr()), "crnerf_cgnet_forward_train_f32")
    return mask, saved


def cgnet_backward(image, params, saved, mask, d_mask):
    """-> the 76 parameter gradients (views of one buffer, shapes of `params`)."""
    lib = _lib.load()
    image, (C, H, W) = _chw(image, "image")
    d_mask = _f32c(d_mask, "d_mask")
    scratch = torch.empty_like(saved)
    sizes = [p.numel() for p in params]
    grads = [g.view(p.shape) for g, p in zip(torch.empty(sum(sizes), device=image.device).split(sizes), params)]
    _lib.check(lib.crnerf_cgnet_backward_f32(_lib.dev_ptr(image), C, H, W, _lib.ptr_array(params, "params"), _lib.dev_ptr(saved), _lib.dev_ptr(mask),
                                             _lib.dev_ptr(d_mask), _lib.dev_ptr(scratch), _lib.ptr_array(grads, "grads"), _lib.stream_ptr()),
               "crnerf_cgnet_backward_f32")
    return grads
