"""GPU parity of the transient-mask network: csrc/cgnet.hip behind crnerf_amd.models.lightweight_seg.

* each operator (conv2d dense / depth-wise / dilated / strided, BatchNorm+PReLU train & eval, AvgPool, FGlo, bilinear
  gather) forward and backward against float64 closed forms on the same inputs;
* the whole network against the imported reference's outputs (golden g13): masks in train and eval mode, running
  statistics, the mask read at the batch's full-resolution pixels, and every parameter gradient (L2 norm + 48 probes);
* against the numpy oracle at a size no golden covers.
Tolerances (fp32 network, ~40 layers): masks 5e-6 absolute (values in (0,1)); gradients 2e-4 of the tensor's
gradient norm against the reference's float64 run (measured: 9e-6)."""
import numpy as np
import pytest
import torch

from _cgnet_fixture import probe_positions, seeded_state
from crnerf_amd import ops
from crnerf_amd.autograd import AvgPool3s2Fn, BilinearGatherFn, BNPReLUFn, Conv2dFn, FGloFn
from crnerf_amd.models import lightweight_seg as LS
from crnerf_amd.models.lightweight_seg import Context_Guided_Network, mask_at_pixels
from oracle import cgnet_ref as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("cfg", [
    dict(cin=3, cout=32, k=3, stride=2, dil=1, groups=1, hw=(44, 60)),     # level1_0
    dict(cin=35, cout=64, k=3, stride=2, dil=1, groups=1, hw=(21, 29)),    # level2_0.conv1x1 (odd size)
    dict(cin=64, cout=64, k=3, stride=1, dil=1, groups=64, hw=(11, 15)),   # F_loc
    dict(cin=64, cout=64, k=3, stride=1, dil=2, groups=64, hw=(11, 15)),   # F_sur, stage 2
    dict(cin=128, cout=128, k=3, stride=1, dil=4, groups=128, hw=(6, 8)),  # F_sur, stage 3 (dilation > half the map)
    dict(cin=128, cout=64, k=1, stride=1, dil=1, groups=1, hw=(11, 15)),   # reduce
    dict(cin=256, cout=1, k=1, stride=1, dil=1, groups=1, hw=(6, 8)),      # classifier
])
def test_conv2d_forward_backward(cfg):
    g = torch.Generator().manual_seed(1)
    H, W = cfg["hw"]
    x = torch.randn(1, cfg["cin"], H, W, generator=g)
    w = torch.randn(cfg["cout"], cfg["cin"] // cfg["groups"], cfg["k"], cfg["k"], generator=g) * 0.2
    pad = (cfg["k"] - 1) // 2 * cfg["dil"]
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, None, cfg["stride"], pad, cfg["dil"], cfg["groups"])
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = Conv2dFn.apply(xd, wd, cfg["stride"], pad, cfg["dil"], cfg["groups"])
    y.backward(dy.to(DEV))
    assert y.shape == yr.shape
    assert _rel(y, yr) < 2e-6 and _rel(xd.grad, xr.grad) < 2e-6 and _rel(wd.grad, wr.grad) < 2e-6


@pytest.mark.parametrize("training", [True, False])
def test_bn_prelu_forward_backward(training):
    g = torch.Generator().manual_seed(2)
    C_, H, W = 35, 22, 30
    x = torch.randn(1, C_, H, W, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C_, eps=1e-3)
    act = torch.nn.PReLU(C_)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.normal_(0, 0.3, generator=g); act.weight.uniform_(0.1, 0.4, generator=g)
        bn.running_mean.normal_(0, 0.3, generator=g); bn.running_var.uniform_(0.5, 2.0, generator=g)
    import copy
    bn_r, act_r = copy.deepcopy(bn).double(), copy.deepcopy(act).double()
    bn.to(DEV); act.to(DEV)
    for m in (bn, act, bn_r, act_r):
        m.train(training)
    xr = x.double().requires_grad_(True)
    yr = act_r(bn_r(xr))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd = x.to(DEV).requires_grad_(True)
    y = BNPReLUFn.apply(xd, bn.weight, bn.bias, act.weight, bn)
    y.backward(dy.to(DEV))
    assert _rel(y, yr) < 2e-6 and _rel(xd.grad, xr.grad) < 5e-6
    assert _rel(bn.weight.grad, bn_r.weight.grad) < 5e-6 and _rel(bn.bias.grad, bn_r.bias.grad) < 5e-6 and _rel(act.weight.grad, act_r.weight.grad) < 5e-6
    assert _rel(bn.running_mean, bn_r.running_mean) < 2e-6 and _rel(bn.running_var, bn_r.running_var) < 2e-6
    assert int(bn.num_batches_tracked) == int(bn_r.num_batches_tracked) == (1 if training else 0)


@pytest.mark.parametrize("hw", [(44, 60), (33, 47), (5, 7)])
def test_avgpool_forward_backward(hw):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, *hw, generator=g)
    xr = x.double().requires_grad_(True)
    yr = torch.nn.functional.avg_pool2d(xr, 3, stride=2, padding=1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd = x.to(DEV).requires_grad_(True)
    y = AvgPool3s2Fn.apply(xd)
    y.backward(dy.to(DEV))
    assert y.shape == yr.shape and _rel(y, yr) < 1e-6 and _rel(xd.grad, xr.grad) < 1e-6


@pytest.mark.parametrize("C_,R", [(64, 8), (128, 8)])
def test_fglo_forward_backward(C_, R):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, C_, 11, 15, generator=g)
    ps = [torch.randn(R, C_, generator=g) * 0.3, torch.randn(R, generator=g) * 0.2, torch.randn(C_, R, generator=g) * 0.5, torch.randn(C_, generator=g) * 0.2]
    leaves_r = [x.double().requires_grad_(True)] + [p.double().requires_grad_(True) for p in ps]
    xr, w1, b1, w2, b2 = leaves_r
    s = torch.sigmoid(torch.relu(xr.mean(dim=(2, 3)) @ w1.t() + b1) @ w2.t() + b2)
    yr = xr * s[:, :, None, None]
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    leaves = [x.to(DEV).requires_grad_(True)] + [p.to(DEV).requires_grad_(True) for p in ps]
    y = FGloFn.apply(*leaves)
    y.backward(dy.to(DEV))
    assert _rel(y, yr) < 2e-6
    for a, b in zip(leaves, leaves_r):
        assert _rel(a.grad, b.grad) < 5e-6


@pytest.mark.parametrize("sigmoid", [False, True])
@pytest.mark.parametrize("picked", [False, True])
def test_bilinear_gather_forward_backward(sigmoid, picked):
    g = torch.Generator().manual_seed(5)
    h, w, Ho, Wo = 6, 8, 44, 60
    x = torch.randn(1, 1, h, w, generator=g)
    idx = torch.randint(0, Ho * Wo, (300,), generator=g) if picked else None
    xr = x.requires_grad_(True)           # fp32 reference: the source-index arithmetic is fp32 in ATen and in the kernel
    full = torch.nn.functional.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False)
    full = torch.sigmoid(full) if sigmoid else full
    yr = full.reshape(-1)[idx] if picked else full
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd = x.detach().to(DEV).requires_grad_(True)
    y = BilinearGatherFn.apply(xd, (Ho, Wo), idx.to(DEV) if picked else None, sigmoid)
    y.backward(dy.to(DEV))
    assert y.shape == yr.shape and _rel(y, yr) < 1e-6 and _rel(xd.grad, xr.grad) < 2e-6


def test_operators_reject_bad_input():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))
    with pytest.raises(ValueError, match="batch 1"):
        ops.conv2d(torch.zeros(2, 3, 8, 8, device=DEV), torch.zeros(4, 3, 3, 3, device=DEV))
    with pytest.raises(ValueError, match="groups"):
        ops.conv2d(torch.zeros(1, 4, 8, 8, device=DEV), torch.zeros(4, 2, 3, 3, device=DEV), groups=2)


def _net(seed):
    net = Context_Guided_Network(classes=1, M=2, N=2, input_channel=3)
    seeded_state(net, seed)
    return net.to(DEV)


@pytest.fixture
def chain_mode():
    """Restores the default (training forward as one autograd node, csrc/cgnet_chain.hip) after a test that switches it."""
    yield LS.set_chain
    LS.set_chain(True)


@pytest.mark.parametrize("chain", [True, False])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_cgnet_matches_reference(golden, tag, chain, chain_mode):
    """chain=True: the training-mode forward / backward are crnerf_cgnet_forward_train_f32 / crnerf_cgnet_backward_f32 (one autograd node);
    chain=False: one node per module (the path eval mode always takes).  Same fixture, same bars."""
    chain_mode(chain)
    g = golden("g13_cgnet")
    net = _net(int(g[tag + "_seed"])).train()
    assert net._chain_applies(torch.from_numpy(g[tag + "_img"]).to(DEV)) == chain
    img = torch.from_numpy(g[tag + "_img"]).to(DEV)
    mask = net(img)
    np.testing.assert_allclose(mask.detach().cpu().numpy(), g[tag + "_mask_train"], atol=5e-6, rtol=0)
    picked = mask_at_pixels(mask, g[tag + "_hw_whole"], torch.from_numpy(g[tag + "_idx"]).to(DEV))
    np.testing.assert_allclose(picked.detach().cpu().numpy(), g[tag + "_picked"], atol=5e-6, rtol=0)
    (picked * torch.from_numpy(g[tag + "_G"]).to(DEV)).sum().backward()
    for k, v in net.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g[tag + "_stat/" + k], atol=1e-5, rtol=1e-5, err_msg=k)
        elif "num_batches" in k:
            assert int(v) == int(g[tag + "_stat/" + k]) == 1
    worst = 0.0
    for k, p in net.named_parameters():
        norm = float(g[tag + "_gnorm/" + k])
        assert p.grad is not None, k
        got = p.grad.detach().double().cpu()
        assert abs(float(got.norm()) - norm) <= 2e-4 * norm + 1e-12, (k, float(got.norm()), norm)
        err = np.abs(got.reshape(-1).numpy()[probe_positions(p.shape, k)] - g[tag + "_gprobe/" + k]).max()
        assert err <= 2e-4 * norm + 1e-12, (k, err, norm)
        worst = max(worst, err / max(norm, 1e-30))
    print("worst gradient probe error / norm:", worst)
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(img).cpu().numpy(), g[tag + "_mask_eval"], atol=5e-6, rtol=0)


def test_cgnet_matches_oracle_at_another_size():
    """1/8-scale Brandenburg-Gate-like aspect (not a golden size), train and eval mode, against the numpy oracle."""
    net = _net(11).train()
    g = torch.Generator().manual_seed(11)
    img = torch.rand(1, 3, 66, 97, generator=g)
    p = {k: v.cpu().numpy().astype(np.float64) for k, v in net.state_dict().items()}
    stats = {}
    want = C.cgnet_forward(img[0].numpy().astype(np.float64), p, True, stats_out=stats)
    with torch.no_grad():
        got = net(img.to(DEV))
    np.testing.assert_allclose(got[0, 0].cpu().numpy(), want, atol=5e-6, rtol=0)
    for k, v in stats.items():
        np.testing.assert_allclose(net.state_dict()[k].cpu().numpy(), v, atol=1e-5, rtol=1e-5)
    p.update(stats)
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(img.to(DEV))[0, 0].cpu().numpy(), C.cgnet_forward(img[0].numpy().astype(np.float64), p, False), atol=5e-6, rtol=0)
    idx = torch.randint(0, 530 * 780, (1024,), generator=g)
    np.testing.assert_allclose(mask_at_pixels(got, (530, 780), idx.to(DEV)).cpu().numpy(), C.mask_at_pixels(want, (530, 780), idx.numpy()), atol=5e-6, rtol=0)


@pytest.mark.parametrize("hw", [(44, 60), (33, 47), (9, 13)])
def test_cgnet_chain_equals_the_module_by_module_path(hw, chain_mode):
    """The one-node training path enqueues the same kernels in the same order: masks and BatchNorm buffers bit for bit; parameter gradients
    differ only where a fan-out's contributions are summed in another order (1e-5 of the tensor's norm; measured ~1e-7)."""
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(1, 3, *hw, generator=g) * 2 - 1).to(DEV)
    G = torch.randn(1, 1, *hw, generator=g).to(DEV)
    out = {}
    for chain in (True, False):
        chain_mode(chain)
        net = _net(21).train()
        assert net._chain_applies(img) == chain
        mask = net(img)
        assert (mask.grad_fn.name() == "CGNetFnBackward") == chain
        (mask * G).sum().backward()
        out[chain] = (mask.detach(), {k: v.clone() for k, v in net.state_dict().items()}, {k: p.grad.clone() for k, p in net.named_parameters()})
    assert torch.equal(out[True][0], out[False][0])
    for k, v in out[True][1].items():
        assert torch.equal(v, out[False][1][k]), k
    worst = 0.0
    for k, gr in out[True][2].items():
        ref = out[False][2][k]
        assert gr.shape == ref.shape and gr.is_contiguous(), k
        err = float((gr - ref).norm() / ref.norm().clamp_min(1e-30))
        assert err <= 1e-5, (k, err)
        worst = max(worst, err)
    print("worst gradient difference / norm:", worst)
    # the conditions that hand the network back to the module-by-module path
    chain_mode(True)
    net = _net(21).train()
    assert not net.eval()._chain_applies(img) and net.train()._chain_applies(img)
    with torch.no_grad():
        assert not net._chain_applies(img)
    assert not net._chain_applies(img.clone().requires_grad_(True))
    net.b1.bn.momentum = 0.2
    assert not net._chain_applies(img)
