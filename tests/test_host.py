"""CPU-only: the C-ABI library loads and exports every symbol include/crnerf.h declares, the drop-in
modules carry the reference's constructor signatures / attributes / state_dict keys, the product fails
loudly without a GPU (no silent fallback), and the weight-pack layout helpers are self-consistent."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import crnerf_amd
import crnerf_amd.synth as synth
from crnerf_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Args:
    nerf_out_dim, img_wh, pertubeCord = 64, [40, 24], False


def header_symbols():
    text = open(os.path.join(ROOT, "include", "crnerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crnerf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_expects():
    assert header_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libcrnerf_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.crnerf_abi_version.restype = ctypes.c_int
    assert lib.crnerf_abi_version() == 2
    lib.crnerf_packed_mlp_bytes.restype = ctypes.c_size_t
    assert lib.crnerf_packed_mlp_bytes() == 11264 + 2416 * 1024
    # the split-precision packs: three bf16 pieces per weight (f32x3; dir_encoding padded to whole stages and queue turns), two fp16 pieces (f32h2)
    for fn, frags in (("crnerf_packed_mlp_x3_bytes", 3648), ("crnerf_packed_mlp_h2_bytes", 2416), ("crnerf_packed_mlp_t_x3_bytes", 3312)):
        getattr(lib, fn).restype = ctypes.c_size_t
        assert getattr(lib, fn)() == 11264 + frags * 1024, fn
    # NULL arguments of the split-precision entry points are rejected before any device work
    for fn in ("crnerf_mlp_forward_f32x3", "crnerf_mlp_forward_f32h2"):
        f = getattr(lib, fn)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        assert f(None, None, None, 4, 0, None) == -1, fn
        assert f(None, None, None, 0, 0, None) == 0, fn
    for fn in ("crnerf_pack_mlp_weights_x3", "crnerf_pack_mlp_weights_h2"):
        f = getattr(lib, fn)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        assert f(None, None, None) == -1, fn
    # error path without touching a device: NULL arguments are rejected with a message
    lib.crnerf_posenc_f32.restype = ctypes.c_int
    lib.crnerf_posenc_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    assert lib.crnerf_posenc_f32(None, None, 4, 15, None) == -1
    lib.crnerf_last_error.restype = ctypes.c_char_p
    assert b"NULL" in lib.crnerf_last_error()
    assert lib.crnerf_posenc_f32(None, None, 0, 15, None) == 0      # empty input is a no-op


def test_render_args_struct_matches_header_field_order():
    text = open(os.path.join(ROOT, "include", "crnerf.h")).read()
    body = text[text.index("typedef struct crnerf_render_args {"):text.index("} crnerf_render_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([a-z_0-9]+)\s*;", body)
    assert fields == [f[0] for f in _lib.RenderArgs._fields_]


def test_nerf_sigma_mirrors_reference_module():
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    m = NeRF_sigma('fine', Args(), in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=48, encode_random=True)
    assert list(m.state_dict().keys()) == ops.MLP_TENSOR_NAMES
    assert [tuple(v.shape) for v in m.state_dict().values()] == list(ops.MLP_TENSOR_SHAPES)
    assert sum(p.numel() for p in m.parameters()) == 619073
    assert (m.typ, m.encode_random, m.encode_appearance, m.in_channels_a) == ('fine', True, True, 48)
    c = NeRF_sigma('coarse', Args(), in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, encode_random=True)
    assert (c.typ, c.encode_random, c.encode_appearance) == ('coarse', False, False)  # nerf.py:132-134
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(1).items()})  # strict
    with pytest.raises(NotImplementedError):
        NeRF_sigma('coarse', Args())                                              # default in_channels_xyz=63 is not the shipped net
    e = PosEmbedding(14, 15)
    assert torch.equal(e.freqs, 2 ** torch.arange(15, dtype=torch.float32))
    with pytest.raises(NotImplementedError):
        PosEmbedding(14, 15, logscale=False)


def test_style_net_mirrors_reference_module():
    from crnerf_amd.models.linearStyleTransfer import style_net
    net = style_net(Args())
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == synth.DECODER_SHAPES
    assert list(net.state_dict().keys()) == list(synth.DECODER_SHAPES.keys())
    assert sum(p.numel() for p in net.parameters()) == 2140899
    assert net.decoder.n_blocks == 0
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})


def test_no_silent_cpu_fallback():
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    from crnerf_amd.models.rendering import render_rays_cross_ray
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            PosEmbedding(3, 4)(torch.zeros(5, 3))
        m = NeRF_sigma('coarse', Args(), in_channels_xyz=93, in_channels_dir=27)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(torch.zeros(4, 120))
        emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            render_rays_cross_ray({"coarse": m}, emb, torch.from_numpy(synth.rays(4)), None, 64, False, 0, 0, 0, 1024, False, args=Args())


def test_render_shim_validates_like_the_boundary_says():
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    from crnerf_amd.models.rendering import render_rays_cross_ray
    m = NeRF_sigma('coarse', Args(), in_channels_xyz=93, in_channels_dir=27)
    rays = torch.from_numpy(synth.rays(4))
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # grad mode takes the training path -- still GPU only
        render_rays_cross_ray({"coarse": m}, emb, rays, None, 64, False, 0, 0, 0, 1024, False, args=Args())
    with torch.no_grad():
        with pytest.raises(NotImplementedError, match="PosEmbedding"):
            render_rays_cross_ray({"coarse": m}, {"xyz": PosEmbedding(9, 10), "dir": emb["dir"]}, rays, None, 64, False, 0, 0, 0, 1024, False, args=Args())
        with pytest.raises(KeyError):                                # reference reads kwargs['args'] unconditionally
            render_rays_cross_ray({"coarse": m}, emb, rays, None)


def test_posenc_slot_order_is_a_permutation_of_the_reference_columns():
    """layout.h posenc_slot_to_col restated: every reference column appears exactly once, pads elsewhere."""
    def slot_to_col(k, F):
        q, h, p, sc = k >> 3, (k >> 2) & 1, (k >> 1) & 1, k & 1
        a = 4 * q + 2 * h + p
        if a < 3 * F:
            return 3 + 6 * (a // 3) + 3 * sc + a % 3
        if a == 3 * F:
            return sc
        if a == 3 * F + 1:
            return 2 if sc == 0 else -1
        return -1
    for F, pad in ((15, 96), (4, 32)):
        cols = [slot_to_col(k, F) for k in range(pad)]
        assert sorted(c for c in cols if c >= 0) == list(range(6 * F + 3))
        assert cols.count(-1) == pad - (6 * F + 3)


def test_synth_is_deterministic_and_shaped():
    a, b = synth.mlp_state(5, 3.0), synth.mlp_state(5, 3.0)
    assert all(np.array_equal(a[k], b[k]) for k in a) and list(a) == ops.MLP_TENSOR_NAMES
    r = synth.rays(1000, seed=1)
    assert r.shape == (1000, 8) and r.dtype == np.float32
    assert np.allclose(np.linalg.norm(r[:, 3:6], axis=1), 1, atol=1e-6) and (r[:, 6] < r[:, 7]).all()


def test_shard_bounds_cover_without_overlap():
    from crnerf_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 1024, 65537):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_kernels_with_asm_lds_dma_do_not_use_m0_indexing(tmp_path):
    """glds16 (csrc/mlp_core.h) rewrites M0 inside inline asm, which the compiler cannot be told (M0 is reserved).  That is
    safe as long as the compiler itself never parks state in M0 in those kernels: dynamic register indexing
    (s_set_gpr_idx_* / v_movrel*) is the one construct that would.  Checked on the generated gfx950 ISA."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "cr-nerf-pytorch_amd", "csrc")
    # (the x3 / h2 units too: xcore_pipe.h's pieces 1..3 of a stage reuse the M0 piece 0 wrote, any number of instructions apart)
    units = ["render_fused16.hip", "mlp_forward16.hip", "mlp_train16.hip", "render_fused_bf16.hip", "mlp_forward_bf16.hip", "render_fused_x3.hip",
             "render_fused_h2.hip", "mlp_backward_x3.hip"]
    procs = [(u, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-S",
                                   "--cuda-device-only", "-I", csrc, os.path.join(csrc, u), "-o", str(tmp_path / (u + ".s"))],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT)) for u in units]
    for u, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out.decode(errors="replace")[-2000:]
        isa = open(tmp_path / (u + ".s")).read()
        assert "global_load_lds_dwordx4" in isa, u
        for bad in ("s_set_gpr_idx", "v_movrel"):
            assert bad not in isa, "%s: %s found -- a dynamic register index would collide with the LDS-DMA asm's use of M0" % (u, bad)
        in_asm, outside = False, []
        for line in isa.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm and "m0" in line.split(";")[0].replace("vm0", ""):
                outside.append(line)
        assert not outside, "%s: the compiler itself touches M0: %s" % (u, outside[:3])


def test_parameter_caches_follow_replaced_parameters():
    """ops.mlp_params / cached_params cached Parameter OBJECTS forever; flows that swap the objects
    (load_state_dict(assign=True), attribute assignment) must be seen by the next lookup."""
    from crnerf_amd.models.nerf import NeRF_sigma
    m = NeRF_sigma('coarse', Args(), in_channels_xyz=93, in_channels_dir=27)
    first = ops.mlp_params(m)
    allp = ops.cached_params(m)
    assert len(first) == 24 and [tuple(p.shape) for p in first] == list(ops.MLP_TENSOR_SHAPES)
    assert ops.mlp_params(m) is first and ops.cached_params(m) is allp           # steady state: no rebuild
    fresh = {k: torch.from_numpy(v) for k, v in synth.mlp_state(3).items()}
    m.load_state_dict(fresh, assign=True)
    second = ops.mlp_params(m)
    assert all(a is not b for a, b in zip(first, second))
    named = dict(m.named_parameters())
    assert all(named[n] is p for n, p in zip(ops.MLP_TENSOR_NAMES, second))
    assert set(map(id, ops.cached_params(m))) == set(map(id, m.parameters()))
    m.xyz_encoding_final.weight = torch.nn.Parameter(torch.zeros(256, 256), requires_grad=False)
    assert ops.mlp_params(m)[16] is m.xyz_encoding_final.weight
    for p in m.parameters():
        p.requires_grad_(False)
    assert not ops.any_requires_grad(m)
    m.static_rgb[0].bias = torch.nn.Parameter(torch.zeros(64))
    assert ops.any_requires_grad(m)


class _Opaque:            # module-level so that pickle can find it
    pass


def test_checkpoint_loader_is_restricted_unless_trusted(tmp_path):
    """load_ckpt must not unpickle arbitrary objects by default.  A Lightning-style file (state_dict + optimizer state +
    argparse.Namespace hyper-parameters, what the reference's training writes) loads under the restricted unpickler; a file with any
    other Python object needs trust_checkpoint=True."""
    import argparse
    from crnerf_amd import pipeline
    lin = torch.nn.Linear(4, 3)
    opt = torch.optim.Adam(lin.parameters(), lr=1e-3)
    lin(torch.ones(2, 4)).sum().backward()
    opt.step()
    ck = {"epoch": 1, "global_step": 7, "pytorch-lightning_version": "1.1.5", "state_dict": {"nerf_coarse." + k: v for k, v in lin.state_dict().items()},
          "optimizer_states": [opt.state_dict()], "lr_schedulers": [{"T_max": 20, "eta_min": 1e-8, "last_epoch": 1}],
          "hparams_name": "hparams_", "hyper_parameters": {"hparams_": argparse.Namespace(lr=5e-4, N_samples=64, img_wh=[32, 32])}}
    path = str(tmp_path / "last.ckpt")
    torch.save(ck, path)
    got = pipeline.extract_model_state_dict(path, "nerf_coarse")
    assert sorted(got) == ["bias", "weight"] and torch.equal(got["weight"], lin.weight)
    fresh = torch.nn.Linear(4, 3)
    pipeline.load_ckpt(fresh, path, model_name="nerf_coarse")
    assert torch.equal(fresh.weight, lin.weight)
    ck["callbacks"] = {"something": _Opaque()}
    torch.save(ck, path)
    with pytest.raises(RuntimeError, match="trust_checkpoint"):
        pipeline.extract_model_state_dict(path, "nerf_coarse")
    assert torch.equal(pipeline.extract_model_state_dict(path, "nerf_coarse", trust_checkpoint=True)["bias"], lin.bias)


def test_flat_adam_has_no_cpu_path():
    """optim.FlatAdam drives crnerf_adam_step_f32 only: CPU parameters are refused at construction, before anything is re-pointed."""
    import torch
    from crnerf_amd import optim
    lin = torch.nn.Linear(3, 2)
    with pytest.raises(TypeError, match="no CPU path"):
        optim.FlatAdam(lin.parameters(), lr=1e-3)
    assert lin.weight.data.storage_offset() == 0 and lin.weight.numel() == 6
    with pytest.raises(ValueError, match="invalid hyper-parameters"):
        optim.FlatAdam(lin.parameters(), lr=-1.0)


def test_cgnet_chain_parameter_order_is_the_state_dict_order():
    """csrc/cgnet_chain.hip indexes the 76 parameters / 14 BatchNorm layers by position: the module's list must be the module tree's own
    order (the reference's state_dict order, lightweight_seg.py:274-368), and as long as crnerf_cgnet_param_count() says."""
    import torch
    from crnerf_amd.models.lightweight_seg import Context_Guided_Network
    net = Context_Guided_Network(classes=1, M=2, N=2, input_channel=3)
    params, bns = net._chain_modules()
    assert [id(p) for p in params] == [id(p) for p in net.parameters()]
    assert [id(b) for b in bns] == [id(m) for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    lib = _lib.load()
    assert len(params) == lib.crnerf_cgnet_param_count() == 76 and len(bns) == lib.crnerf_cgnet_bn_count() == 14
    assert lib.crnerf_cgnet_arena_bytes(3, 48, 64) > 4 * 32 * 24 * 32 and lib.crnerf_cgnet_arena_bytes(0, 48, 64) == 0
    assert not net._chain_applies(torch.zeros(1, 3, 48, 64))          # CPU parameters: never the chain (the module path then raises, as before)


def test_get_scheduler_mirrors_the_reference_factory():
    """optim.get_scheduler = utils/__init__.py:45-63 ('steplr', 'cosine', 'poly'); the warm-up wrapper is refused loudly, not silently dropped."""
    import types
    import torch
    from crnerf_amd import optim
    lin = torch.nn.Linear(2, 2)
    hp = types.SimpleNamespace(optimizer="sgd", lr=0.1, momentum=0.9, weight_decay=0.0, lr_scheduler="steplr", decay_step=[1], decay_gamma=0.5,
                               num_epochs=4, poly_exp=0.9, warmup_epochs=0)
    opt = optim.get_optimizer(hp, [lin])
    sch = optim.get_scheduler(hp, opt)
    opt.step(); sch.step()
    assert abs(optim.get_learning_rate(opt) - 0.05) < 1e-12
    for name, want in (("cosine", torch.optim.lr_scheduler.CosineAnnealingLR), ("poly", torch.optim.lr_scheduler.LambdaLR)):
        hp.lr_scheduler = name
        assert isinstance(optim.get_scheduler(hp, optim.get_optimizer(hp, [lin])), want)
    hp.lr_scheduler = "exp"
    with pytest.raises(ValueError, match="scheduler not recognized"):
        optim.get_scheduler(hp, opt)
    hp.lr_scheduler, hp.warmup_epochs = "cosine", 2
    with pytest.raises(NotImplementedError, match="GradualWarmupScheduler"):
        optim.get_scheduler(hp, opt)


class _DeferringNode(torch.autograd.Function):
    """A multi-use node in the style of autograd.DecoderFn, on CPU: hands its parameter gradient to the deferral."""

    @staticmethod
    def forward(ctx, x, w):
        from crnerf_amd import autograd as AG
        ctx.save_for_backward(x, w)
        ctx.defer, ctx.leaves = AG._DEFER_ON[0], ([AG._leaf_of(w)] if AG._DEFER_ON[0] else None)
        return x * w

    @staticmethod
    def backward(ctx, g):
        from crnerf_amd import autograd as AG
        x, w = ctx.saved_tensors
        return (g * w,) + AG._param_grads(ctx, [w], [g * x])


class _RaisingNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError("HIP error stand-in")


def test_deferred_gradients_survive_a_backward_that_raised():
    """the engine drops the end-of-pass callback of a backward that raises; the next healthy backward must still deliver .grad,
    and nothing of the dead pass may leak into it."""
    from crnerf_amd import autograd as AG
    w = torch.nn.Parameter(torch.full((3,), 2.0))
    x = torch.arange(3.0, requires_grad=True)
    with AG.deferred_param_grads(True):
        y = _DeferringNode.apply(_RaisingNode.apply(x), w)     # backward: the deferring node runs first, then the raising one
    with pytest.raises(RuntimeError, match="stand-in"):
        y.sum().backward()
    assert w.grad is None
    for _ in range(2):                                         # with and without re-entering the context in between
        y = None
        with AG.deferred_param_grads(True):
            y = _DeferringNode.apply(x, w)
        w.grad = None
        y.sum().backward()
        assert torch.equal(w.grad, torch.arange(3.0))          # this pass's gradient only
        assert not AG._DEFERRED["pending"] and AG._DEFERRED["task"] is None
    # a node built before the failure, run again without re-entering the context (retain_graph): the stale state is keyed to the dead pass
    with AG.deferred_param_grads(True):
        y = _DeferringNode.apply(_RaisingNode.apply(x), w)
        y2 = _DeferringNode.apply(x, w)
    with pytest.raises(RuntimeError):
        y.sum().backward()
    w.grad = None
    y2.sum().backward()
    assert torch.equal(w.grad, torch.arange(3.0))


def test_deferred_gradients_leave_frozen_parameters_alone():
    """a frozen Parameter is a leaf; AccumulateGrad never gives it a .grad and neither may the deferral (FlatAdam would train it)."""
    from crnerf_amd import autograd as AG
    frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
    conv = torch.nn.Parameter(torch.ones(3, 1, 1, 1), requires_grad=False)
    assert AG._leaf_of(frozen) is None and AG._leaf_of(conv.view(3, 1)) is None
    x = torch.arange(3.0, requires_grad=True)
    with AG.deferred_param_grads(True):
        y = _DeferringNode.apply(x, frozen)
    y.sum().backward()
    assert frozen.grad is None and torch.equal(x.grad, torch.ones(3))
    live = torch.nn.Parameter(torch.ones(3, 1, 1, 1))
    assert AG._leaf_of(live) is live and AG._leaf_of(live.view(3, 1)) is live
