"""The fused bf16 TRAINING renderer (crnerf_render_rays_train_bf16 + autograd.MixedFusedRenderFn) of the opt-in mixed-precision mode
(no counterpart in the reference, whose autograd is fp32: rendering.py:100-143 under torch autograd).  Checked against
  * the bf16 inference renderer (same outputs, bit for bit: the hooks do not touch the arithmetic),
  * the per-layer GEMM twins crnerf_mlp_forward_train_mixed_f32 / crnerf_mlp_backward_mixed_f32 on the same points -- which
    tests/test_gpu_train_fused.py holds to torch autograd through the oracle's bf16-operand Linear: same saved activations / relu bits /
    embedded input once both storage orders are undone, same gradients from either buffer,
  * its own consistency (relu bits <-> stored rows)."""
import numpy as np
import pytest
import torch

import crnerf_amd.synth as synth
from crnerf_amd import autograd as AG
from crnerf_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def C(a):
    return T(np.ascontiguousarray(a)).to(DEV)


def _inputs(R, Nc, Ni, seed=0):
    rng = np.random.default_rng(seed)
    rays = synth.rays(R, seed=seed)
    z = np.sort(rng.uniform(rays[:, 6:7], rays[:, 7:8], (R, Nc)).astype(np.float32), -1)
    u = rng.uniform(0, 1, (R, max(Ni, 1))).astype(np.float32)
    return C(rays), C(z), C(u), C(rng.normal(size=(R, Nc)).astype(np.float32)), C(rng.normal(size=(R, Nc + Ni)).astype(np.float32))


def _perm32(f):
    return (f & ~31) | (((f >> 2) & 3) << 3) | (((f >> 4) & 1) << 2) | (f & 3)


def _perm_fused(f):
    return (f & ~12) | (((f >> 2) & 1) << 3) | (((f >> 3) & 1) << 2)


def _slot_to_col(k, F):          # csrc/layout.h posenc_slot_to_col_b
    s, hh, p, sc = k >> 4, (k >> 3) & 1, (k >> 1) & 3, k & 1
    a = 8 * s + 4 * hh + p
    if a < 3 * F:
        return 3 + 6 * (a // 3) + 3 * sc + (a % 3)
    if a == 3 * F:
        return sc
    if a == 3 * F + 1:
        return 2 if sc == 0 else -1
    return -1


def _decode(buf, P, fused):
    """(rows[10,P,256] in reference feature order, bits[10,P,32] u8, xb[P,120] in reference column order) of a mixed-precision acts buffer."""
    n_rows = 10 * P * 512
    rows = buf[:n_rows].view(torch.bfloat16).view(10, P, 256).float()
    f = np.arange(256)
    idx = torch.from_numpy(_perm_fused(f) if fused else _perm32(f)).to(buf.device)
    rows = rows[:, :, idx]
    bits = buf[n_rows:n_rows + 10 * P * 32].view(10, P, 32)
    xb = buf[n_rows + 10 * P * 32:n_rows + 10 * P * 32 + P * 256].view(torch.bfloat16).view(P, 128).float()
    x = torch.zeros(P, 120, device=buf.device)
    if fused:
        for k in range(96):
            c = _slot_to_col(k, 15)
            if c >= 0:
                x[:, c] = xb[:, k]
        for k in range(32):
            c = _slot_to_col(k, 4)
            if c >= 0:
                x[:, 93 + c] = xb[:, 96 + k]
    else:
        x[:, :93] = xb[:, :93]
        x[:, 93:] = xb[:, 96:123]
    return rows, bits, x


def _bits_of_rows(rows):
    """The activity-bit record linear_bf16_kernel defines (byte [q4][u], bit 4b + i <-> feature 32u + 16b + 4q4 + i) from rows in reference order."""
    S, P, _ = rows.shape
    act = (rows > 0).view(S, P, 8, 2, 4, 4)                        # [u][b][q4][i]
    w = (1 << (4 * torch.arange(2, device=rows.device).view(1, 1, 1, 2, 1, 1) + torch.arange(4, device=rows.device).view(1, 1, 1, 1, 1, 4)))
    by = (act.long() * w).sum(dim=(3, 5))                          # [S,P,u,q4]
    return by.permute(0, 1, 3, 2).reshape(S, P, 32).to(torch.uint8)


@torch.no_grad()
@pytest.mark.parametrize("R,Nc,Ni", [(37, 64, 64), (5, 33, 20), (16, 64, 0), (9, 128, 128)])
def test_bf16_train_forward_equals_inference_and_gemm_twins(R, Nc, Ni):
    st_c, st_f = {k: C(v) for k, v in synth.mlp_state(5, 2.0, 0.5).items()}, {k: C(v) for k, v in synth.mlp_state(6, 2.0, 0.5).items()}
    pc, pf = ops.pack_mlp_weights(st_c, precision="bf16"), ops.pack_mlp_weights(st_f, precision="bf16")
    rays, z, u, nc, nf = _inputs(R, Nc, Ni)
    kw = dict(z_coarse=z, u=u if Ni else None, noise_coarse=nc, noise_fine=nf if Ni else None, noise_std=0.7, precision="bf16")
    inf = ops.render_rays(pc, pf if Ni else None, rays, Nc, Ni, want_z_fine=True, **kw)
    trn = ops.render_rays(pc, pf if Ni else None, rays, Nc, Ni, train=True, **kw)
    for k in inf:
        assert torch.equal(inf[k], trn[k]), k
    for tag, st, zz, N in (("coarse", st_c, z, Nc),) + ((("fine", st_f, trn["z_fine"], Nc + Ni),) if Ni else ()):
        P = R * N
        x = AG._embed_points(rays, zz, None)
        packed, tensors = ops.pack_mlp_weights_mixed(st)
        out, acts = ops.mlp_forward_train_mixed(packed, tensors, x)
        raw = trn["raw_" + tag].view(-1, 65)
        err = (raw - out).abs()
        # two bf16 evaluations of the same arithmetic: summation order, sin/cos routine, the rare rounding flip of an intermediate activation
        print(tag, "raw max %.3e mean %.3e" % (float(err.max()), float(err.mean())))
        assert float(err.mean()) <= 2e-4 and float(err.max()) <= 6e-2, (tag, float(err.max()), float(err.mean()))
        rf, bf, xf = _decode(trn["acts_" + tag], P, True)
        rg, bg, xg = _decode(acts, P, False)
        # embedded input: bf16 of the same embedding up to the routines' last bits
        assert float((xf - xg).abs().max()) <= 2.0 ** -7 and float(((xf - xg).abs() > 0).float().mean()) <= 0.05, tag
        # rows: equal up to a bf16 ulp wherever no upstream flip cascaded
        for s in range(10):
            w = 128 if s == 9 else 256
            a, b = rf[s, :, :w], rg[s, :, :w]
            close = (a - b).abs() <= 2.0 ** -6 * torch.maximum(a.abs(), b.abs()) + 2e-3
            assert float(close.float().mean()) >= 0.995, (tag, s, float(close.float().mean()))
        # relu bits: the record of the stored rows themselves, exactly; and the two producers agree on all but the flipped few
        for s in (0, 1, 2, 3, 4, 5, 6, 7, 9):
            want = _bits_of_rows(rf[s:s + 1])[0]
            got = bf[s]
            if s == 9:
                want, got = want.view(P, 4, 8)[:, :, :4], got.view(P, 4, 8)[:, :, :4]        # 128 features: groups u = 0..3
            assert torch.equal(want, got), (tag, s)
            agree = float((bf[s] == bg[s]).float().mean()) if s != 9 else float((bf[s].view(P, 4, 8)[:, :, :4] == bg[s].view(P, 4, 8)[:, :, :4]).float().mean())
            assert agree >= 0.99, (tag, s, agree)


@torch.no_grad()
@pytest.mark.parametrize("R,Nc,Ni", [(64, 64, 64), (7, 33, 20)])
def test_bf16_train_backward_from_fused_buffers_matches_gemm_twins(R, Nc, Ni):
    """The same backward (crnerf_mlp_backward_mixed_ex_f32) from the fused renderer's buffers and from the GEMM twins' buffers of the same points."""
    st_c, st_f = {k: C(v) for k, v in synth.mlp_state(5, 1.0, 0.5).items()}, {k: C(v) for k, v in synth.mlp_state(6, 1.0, 0.5).items()}
    pc, pf = ops.pack_mlp_weights(st_c, precision="bf16"), ops.pack_mlp_weights(st_f, precision="bf16")
    rays, z, u, nc, nf = _inputs(R, Nc, Ni, seed=3)
    trn = ops.render_rays(pc, pf, rays, Nc, Ni, train=True, z_coarse=z, u=u, noise_coarse=nc, noise_fine=nf, noise_std=0.3, precision="bf16")
    gen = torch.Generator().manual_seed(7)
    for tag, st, zz, N in (("coarse", st_c, z, Nc), ("fine", st_f, trn["z_fine"], Nc + Ni)):
        P = R * N
        d_out = torch.randn(P, 65, generator=gen).to(DEV)
        x = AG._embed_points(rays, zz, None)
        packed, tensors = ops.pack_mlp_weights_mixed(st)
        out, acts = ops.mlp_forward_train_mixed(packed, tensors, x)
        g_gemm = ops.mlp_backward_mixed(packed, tensors, x, out, d_out, acts)
        g_fused = ops.mlp_backward_mixed(packed, tensors, None, trn["raw_" + tag].view(-1, 65), d_out, trn["acts_" + tag], fused_acts=True)
        for name, a, b in zip(ops.MLP_TENSOR_NAMES, g_fused, g_gemm):
            scale = float(b.abs().max()) + 1e-12
            rel = float((a - b).norm() / (b.norm() + 1e-30))
            print("%-7s %-28s max|d| %.3e of %.3e  rel-L2 %.3e" % (tag, name, float((a - b).abs().max()), scale, rel))
            assert torch.isfinite(a).all() and rel <= 2e-2 and float((a - b).abs().max()) <= 4e-2 * scale, (tag, name, rel)


def test_bf16_train_fused_autograd_matches_unfused_mixed_path():
    """MixedFusedRenderFn end to end (render -> loss -> backward) against the un-fused mixed path (_render_unfused with the GEMM twins)."""
    from crnerf_amd.models import rendering
    from test_gpu_train_fused import _grads, _modules
    models, emb, args = _modules(gain=2.45, sigma_bias=-1.0, band_limit=4)
    R = 128
    rays, z, u, nc, nf = _inputs(R, 64, 64, seed=5)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)

    def loss_of(out):
        return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum() + 0.1 * out["depth_fine"].sum()
    AG.set_training_precision("bf16")
    try:
        g_fu = _grads(models, lambda: loss_of(AG.fused_render_with_grad(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0)))
        g_un = _grads(models, lambda: loss_of(rendering._render_unfused(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0,
                                                                         False, 1 << 20, train=True)))
    finally:
        AG.set_training_precision("f32")
    for k in g_un:
        rel = float((g_fu[k] - g_un[k]).norm() / (g_un[k].norm() + 1e-30))
        cos = float((g_fu[k] * g_un[k]).sum() / (g_fu[k].norm() * g_un[k].norm() + 1e-30))
        # two bf16 evaluations whose relu masks / hierarchical depths differ in a few places: a few per cent
        assert rel <= 0.15 and cos >= 0.985, (k, rel, cos)


@torch.no_grad()
def test_bf16_train_twin_cold_l2_is_deterministic_and_equals_inference():
    """Guard for the training twin's weight-ring waits (mlp_core_bf16p.h: the stage barrier waits vmcnt(4 + the stores issued since the pieces
    it needs) -- a window that is too wide by one store would let a wave read a weight stage before its LDS-DMA has landed, and only a cold L2
    makes a piece late enough to show it; tests/test_gpu_bf16.py::test_render_cold_l2_is_deterministic is the inference kernel's guard).
    65,536 rays x (64 + 64) in 5,460-ray chunks (the chunk size of the training step), L2 thrashed by a 1 GiB copy before every chunk, four
    passes: the outputs must be bit-identical to the INFERENCE renderer's every time, and the saved rows identical between passes."""
    st_c = {k: C(v) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
    st_f = {k: C(v) for k, v in synth.mlp_state(2, 2.0, 0.5).items()}
    pc, pf = ops.pack_mlp_weights(st_c, precision="bf16"), ops.pack_mlp_weights(st_f, precision="bf16")
    R, step = 65536, 5460
    rays = C(synth.rays(R, seed=0, H=256, W=256))
    z_steps, u = torch.linspace(0, 1, 64, device=DEV), torch.linspace(0, 1, 64, device=DEV)
    junk_a, junk_b = torch.empty(1 << 28, device=DEV), torch.zeros(1 << 28, device=DEV)
    keys = ("weights_coarse", "feature_coarse", "depth_coarse", "weights_fine", "feature_fine", "depth_fine", "z_fine")
    ref = {k: torch.cat([ops.render_rays(pc, pf, rays[i:i + step], 64, 64, z_steps=z_steps, u=u, want_z_fine=True, precision="bf16")[k]
                         for i in range(0, R, step)]) for k in keys}
    first = None
    for it in range(4):
        sums = []
        for i in range(0, R, step):
            junk_a.copy_(junk_b)
            out = ops.render_rays(pc, pf, rays[i:i + step], 64, 64, z_steps=z_steps, u=u, precision="bf16", train=True)
            for k in keys:
                assert torch.equal(out[k], ref[k][i:i + step]), (it, i, k)
            # the saved state, folded to a few numbers per chunk (exact integer sums over the regions the kernel writes)
            for tag in ("coarse", "fine"):
                raw = out["raw_" + tag]
                P = raw.shape[0] * raw.shape[1]
                buf = out["acts_" + tag]
                rows = buf[:10 * P * 512].view(10, P, 512)
                bits = buf[10 * P * 512:10 * P * 544].view(10, P, 32)
                xb = buf[10 * P * 544:10 * P * 544 + P * 256]
                isum = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum())  # noqa: E731
                # (slot 9 holds 128 features = the first 256 bytes of a row; slot 8 is linear: no bits; the rest of the buffer is never written)
                sums += [isum(rows[:9]), isum(rows[9, :, :256]), isum(bits[:8]), isum(xb), float(raw.double().sum())]
            del out
        if first is None:
            first = sums
        assert sums == first, it


def test_bf16_recompute_mode_gradients_are_bit_identical_to_the_stored_mode():
    """set_training_recompute(True) in the mixed-precision mode: forward = the bf16 inference renderer, backward re-runs the chunk through the
    training twin -- whose outputs and saved rows are bit-identical to what the stored mode kept, so the two modes' gradients are EQUAL."""
    from test_gpu_train_fused import _grads, _modules
    models, emb, args = _modules(gain=2.45, sigma_bias=-1.0, band_limit=4)
    R = 96
    rays, z, u, nc, nf = _inputs(R, 64, 64, seed=9)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run():
        out = AG.fused_render_with_grad(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0)
        return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum() + 0.1 * (out["weights_fine"] ** 2).sum()
    AG.set_training_precision("bf16")
    try:
        g_st = _grads(models, run)
        AG.set_training_recompute(True)
        g_rc = _grads(models, run)
    finally:
        AG.set_training_recompute(False)
        AG.set_training_precision("f32")
    for k in g_st:
        assert torch.equal(g_st[k], g_rc[k]), k
