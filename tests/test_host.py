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


@pytest.fixture(scope="module")
def isa_units(tmp_path_factory):
    """The gfx950 ISA (hipcc -S --cuda-device-only, build.py's flags) of every translation unit that carries a hand-written instruction stream
    (tools/isa_audit.py AUDITED_UNITS): {unit: path of the .s}.  Cross-compiles without a GPU; ~1 min on 8 cores."""
    import shutil
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_audit
    out = tmp_path_factory.mktemp("isa")
    return isa_audit.compile_units(isa_audit.AUDITED_UNITS, str(out))


def test_kernels_with_asm_lds_dma_do_not_use_m0_indexing(isa_units):
    """glds16 (csrc/mlp_core.h) rewrites M0 inside inline asm, which the compiler cannot be told (M0 is reserved).  That is
    safe as long as the compiler itself never parks state in M0 in those kernels: dynamic register indexing
    (s_set_gpr_idx_* / v_movrel*) is the one construct that would.  Checked on the generated gfx950 ISA.
    (The x3 / h2 units too: glds16_more reuses the M0 the stage's first piece wrote, any number of instructions apart.)"""
    for u, path in isa_units.items():
        isa = open(path).read()
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


def test_hand_written_streams_keep_their_wait_states(isa_units):
    """VERDICT r4 #3.  hipcc pads hazards between its own instructions only: what an asm statement writes is not a VALU write to its hazard
    recogniser, what it reads is not a read.  tools/isa_audit.py walks the emitted ISA of every kernel and holds each dependent pair with a side
    inside an asm statement to the gfx950 table (VALU write -> MFMA operand 2; partial-dword write -> reader / same-register RMW 1; MFMA D ->
    reader P + 3 (+1); VALU-written SGPR -> VMEM 5; M0 -> LDS-DMA 1; ...).  Round 5 found two defects with it: the round-4 h2 training form's
    wrong tiles (an MFMA one state behind the asm v_fma_mixhi that completed its B operand) and SGPR spill reloads (v_readlane) two states
    in front of the LDS-DMA that takes them as base in mlp_forward_h2_kernel -- both closed by construction (h2_operands_ready, glds16's SALU
    copy).  This test keeps every later build honest."""
    import isa_audit
    total_defects, report = 0, []
    strict_only = {}
    for u, path in isa_units.items():
        kernels = isa_audit.audit_file(path)
        assert kernels, "%s: no kernel found in the ISA" % u
        for name, found in kernels.items():
            defects, so = isa_audit.split(found)
            total_defects += len(defects)
            for rule, need, have, have_t, w, c, reg in defects[:4]:
                report.append("%s %s: %s need %d have %d (timed %d): L%d %s -> L%d %s" % (u, name, rule, need, have, have_t, w.line,
                                                                                         w.text.split(";")[0].strip(), c.line, c.text.split(";")[0].strip()))
            if so:
                strict_only[(u, name)] = so
    assert total_defects == 0, "\n".join(report)
    # pairs short in LLVM's instruction count only (an MFMA stream in between that the count does not credit) exist in ONE family: the bf16 pair
    # core's epilogue quarters (v_cvt_pk_bf16_f32 of an accumulator pair 10-11 instructions, >= 17 timed states, behind the MFMA that wrote it)
    for (u, name), so in strict_only.items():
        assert "bf16p" in u, (u, name, so[0][0])
        assert all(f[0] == "mfma->any" and f[3] >= f[1] + 4 for f in so), (u, name)
    # the audit has teeth: the round-4 LDS-DMA statement (no SALU copy, base SGPR straight from a v_readlane) and an MFMA one state behind an asm
    # VALU write are both reported
    bad = """
kern:
	v_readlane_b32 s84, v252, 11
	v_readlane_b32 s85, v252, 12
	;;#ASMSTART
	s_mov_b32 m0, s2
	s_nop 0
	global_load_lds_dwordx4 v204, s[84:85] offset:0
	;;#ASMEND
	;;#ASMSTART
	v_fma_mixhi_f16 v15, v211, v249, -v9 op_sel:[0,0,1] op_sel_hi:[0,0,1]
	;;#ASMEND
	v_cndmask_b32_e32 v12, v32, v12, vcc
	v_mfma_f32_32x32x16_f16 a[48:63], v[34:37], v[14:17], a[48:63]
	;;#ASMSTART
	v_cvt_pk_bf16_f32 v82, a48, a49
	;;#ASMEND
	s_endpgm
"""
    p = os.path.join(os.path.dirname(next(iter(isa_units.values()))), "bad.s")
    with open(p, "w") as f:
        f.write(bad)
    rules = sorted({f[0] for f in isa_audit.audit_file(p)["kern"] if f[3] < f[1]})
    assert rules == ["mfma->any", "valu->mfma", "vsgpr->vmem"], rules


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
        assert not AG._DEFERRED                                # this pass's dict is gone; the dead pass's went when the forward started
    # a node built before the failure, run again without re-entering the context (retain_graph): the stale state is keyed to the dead pass
    with AG.deferred_param_grads(True):
        y = _DeferringNode.apply(_RaisingNode.apply(x), w)
        y2 = _DeferringNode.apply(x, w)
    with pytest.raises(RuntimeError):
        y.sum().backward()
    w.grad = None
    y2.sum().backward()
    assert torch.equal(w.grad, torch.arange(3.0))


class _NestedBackwardNode(torch.autograd.Function):
    """Runs a whole inner backward pass (its own graph task, with deferring nodes of its own) inside its backward -- what
    torch.utils.checkpoint(use_reentrant=True) does."""

    @staticmethod
    def forward(ctx, x, w_inner):
        ctx.w_inner = w_inner
        ctx.save_for_backward(x)
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        from crnerf_amd import autograd as AG
        (x,) = ctx.saved_tensors
        with torch.enable_grad():
            xi = x.detach().requires_grad_(True)
            with AG.deferred_param_grads(True):
                yi = _DeferringNode.apply(xi, ctx.w_inner)
            yi.sum().backward()                                 # inner pass: own graph task id, own pending dict, own callback
        return g, None


def test_deferred_gradients_survive_a_nested_backward():
    """a nested / re-entrant backward has its own graph task id; with ONE pending slot it silently discarded what the outer pass had
    already deferred.  Pending gradients are kept per task: both passes deliver."""
    from crnerf_amd import autograd as AG
    w_outer = torch.nn.Parameter(torch.full((3,), 2.0))
    w_inner = torch.nn.Parameter(torch.full((3,), 5.0))
    x = torch.arange(3.0, requires_grad=True)
    with AG.deferred_param_grads(True):
        # backward order: the outer deferring node first (its gradient is pending), then the node that runs the inner pass
        y = _DeferringNode.apply(_NestedBackwardNode.apply(x, w_inner), w_outer)
    y.sum().backward()
    assert torch.equal(w_outer.grad, torch.arange(3.0))        # the outer pass's deferred gradient arrived
    assert torch.equal(w_inner.grad, torch.arange(3.0))        # so did the inner pass's
    assert not AG._DEFERRED


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


def test_cached_lists_follow_replaced_parameters_and_hold_no_views():
    """ops.cached_list (round 6): the decoder's / encoder's kernel-order lists are looked up once per module; a swapped Parameter rebuilds them,
    and the decoder's reshaped [cout, cin] views are made per call (one view node per use: the gradient accumulation order of a step stays
    use by use)."""
    import _procedural_scene as S
    from crnerf_amd.models.linearStyleTransfer import encoder_sameoutputsize, style_net
    net = style_net(args=S.hparams(), residual_blocks=1)
    a, b = net.decoder_tensors(), net.decoder_tensors()
    assert len(a) == 22 and [tuple(t.shape) for t in a[:8]] == [(128, 64), (128,), (64, 128), (64,), (32, 64), (32,), (1024, 1024), (1024,)]
    for x, y in zip(a, b):
        assert x.data_ptr() == y.data_ptr() and ((x is y) == isinstance(x, torch.nn.Parameter))
    named = dict(net.named_parameters())
    assert a[6] is named["multi_net.snet.fc.weight"] and a[21] is named["decoder.feat_2_rgb_list.0.bias"]
    net.multi_net.snet.fc.weight = torch.nn.Parameter(torch.zeros(1024, 1024))
    assert net.decoder_tensors()[6] is net.multi_net.snet.fc.weight
    enc = encoder_sameoutputsize(out_channel=64)
    lst = ops.cached_list(enc, "probe", lambda: [enc.conv1.weight])
    assert ops.cached_list(enc, "probe", lambda: [enc.conv1.weight]) is lst
    enc.load_state_dict({k: v.clone() for k, v in enc.state_dict().items()}, assign=True)
    assert ops.cached_list(enc, "probe", lambda: [enc.conv1.weight])[0] is enc.conv1.weight


def test_hostpin_confines_all_threads_to_one_cache_domain_and_restores():
    """crnerf_amd.hostpin (round 6: the 1,024-ray training step is bound by two host threads handing the GIL over; with both inside one L3 domain it
    sits on its device time).  pin_host_threads(): every thread the process has ends up inside one last-level-cache domain and inside the mask it
    had; pin_step_threads(): the caller (and the device's autograd worker) only; unpin restores the masks; an explicit CPU set is honoured; a host
    without cache topology in sysfs is left alone."""
    import threading
    from crnerf_amd import hostpin
    before = os.sched_getaffinity(0)
    stop = threading.Event()
    th = threading.Thread(target=stop.wait)
    th.start()
    try:
        saved = hostpin.pin_host_threads()
        cpu = hostpin.current_cpu()
        if saved is None:                                        # no cache topology here: nothing may have changed
            assert os.sched_getaffinity(0) == before
        else:
            dom = set(saved["cpus"])
            assert dom and dom <= before and dom <= hostpin.l3_domain(min(dom)) and (cpu is None or hostpin.current_cpu() in dom)
            assert os.sched_getaffinity(0) == dom and os.sched_getaffinity(th.native_id) == dom
            assert hostpin.allowed_before(saved) == before
            hostpin.unpin_host_threads(saved)
            assert os.sched_getaffinity(0) == before and os.sched_getaffinity(th.native_id) == before
        # the recommended form: the caller and the autograd worker of the device only (on "cpu" the backward runs on the caller itself)
        assert hostpin.autograd_thread_id("cpu") == threading.get_native_id()
        saved = hostpin.pin_step_threads("cpu")
        if saved is not None:
            assert os.sched_getaffinity(0) == set(saved["cpus"]) and os.sched_getaffinity(th.native_id) == before
            assert sorted(k for k in saved if k != "cpus") == [threading.get_native_id()]
            hostpin.unpin_host_threads(saved)
            assert os.sched_getaffinity(0) == before
        one = {min(before)}
        saved = hostpin.pin_host_threads(one)
        assert os.sched_getaffinity(0) == one and saved["cpus"] == one
        hostpin.unpin_host_threads(saved)
        assert os.sched_getaffinity(0) == before
        assert hostpin.l3_domain(min(before), sys_cpu="/nonexistent") is None and hostpin._parse_cpu_list("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    finally:
        os.sched_setaffinity(0, before)
        stop.set()
        th.join()
