"""FlatAdam (crnerf_amd/optim.py, crnerf_adam_step_f32) against the optimiser the reference builds: torch.optim.Adam(parameters, lr,
eps=1e-8, weight_decay) (utils/__init__.py:24-33).  Both are stepped on the SAME gradients; the update is fp32 element-wise arithmetic
in the same order as torch's, so the parameters may differ by rounding of the fused multiply-adds only: 2e-7 absolute + 2e-6 relative
after ten steps (lr 5e-4: one step moves a parameter by <= 5e-4)."""
import copy

import numpy as np
import pytest
import torch

from crnerf_amd import optim

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(1,), (3,), (5, 7), (4096,), (4097,), (64, 129), (256, 256), (2, 3, 3, 3), (12289,)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in SHAPES]


def _close(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        err = (x.detach() - y.detach()).abs()
        bound = 2e-7 + 2e-6 * y.detach().abs()
        assert bool((err <= bound).all()), (what, i, float(err.max()))


@pytest.mark.parametrize("weight_decay", [0.0, 1e-2])
def test_flat_adam_takes_the_steps_torch_adam_takes(weight_decay):
    ours, theirs = _params(1), _params(1)
    before = [p.detach().clone() for p in ours]
    a = optim.FlatAdam(ours, lr=5e-4, eps=1e-8, weight_decay=weight_decay)
    b = torch.optim.Adam(theirs, lr=5e-4, eps=1e-8, weight_decay=weight_decay)
    for p, q in zip(ours, before):
        assert torch.equal(p.detach(), q)                     # moving the parameters into the flat buffer changes no value
    g = torch.Generator().manual_seed(2)
    for step in range(10):
        for k, (p, q) in enumerate(zip(ours, theirs)):
            if k == 2:
                continue                                      # a parameter that never receives a gradient: skipped by both
            grad = (torch.randn(*p.shape, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))).to(DEV)
            if k == 5:                                        # a strided gradient (what an expand / transpose backward can leave)
                grad = grad.t().contiguous().t()
                assert not grad.is_contiguous()
            if k == 3 and step >= 5:
                grad = torch.zeros_like(grad)                 # exact zeros: m decays, v decays, the quotient stays finite
            p.grad, q.grad = grad, grad.clone()
        a.step()
        b.step()
        _close(ours, theirs, "step %d" % step)
    assert torch.equal(ours[2].detach(), before[2])
    assert all(p._version > 0 for k, p in enumerate(ours))    # caches keyed on p._version (NeRF_sigma's packed weights) see the raw-pointer update


def test_flat_adam_state_moves_to_torch_adam_and_back():
    ours, theirs = _params(3), _params(3)
    a = optim.FlatAdam(ours, lr=1e-3)
    b = torch.optim.Adam(theirs, lr=1e-3)
    g = torch.Generator().manual_seed(4)

    def grads():
        for p, q in zip(ours, theirs):
            p.grad = torch.randn(*p.shape, generator=g).to(DEV)
            q.grad = p.grad.clone()

    for _ in range(3):
        grads()
        a.step()
    with torch.no_grad():
        for p, q in zip(ours, theirs):
            q.copy_(p)
    b.load_state_dict(copy.deepcopy(a.state_dict()))          # FlatAdam -> torch.optim.Adam: the layout a Lightning checkpoint stores
    grads()
    a.step()
    b.step()
    _close(ours, theirs, "after handing the state to torch")
    ours2 = _params(5)
    a2 = optim.FlatAdam(ours2, lr=1e-3)
    with torch.no_grad():
        for p, q in zip(ours2, theirs):
            p.copy_(q)
    a2.load_state_dict(copy.deepcopy(b.state_dict()))         # and back
    for p2, q in zip(ours2, theirs):
        p2.grad = torch.randn(*p2.shape, generator=g).to(DEV)
        q.grad = p2.grad.clone()
    a2.step()
    b.step()
    _close(ours2, theirs, "after taking torch's state")
    ours3 = _params(8)
    a3 = optim.FlatAdam(ours3, lr=1e-3)
    with torch.no_grad():
        for p, p3 in zip(ours, ours3):
            p3.copy_(p)
    a3.load_state_dict(copy.deepcopy(a.state_dict()))         # FlatAdam -> FlatAdam: a resumed run continues bit for bit
    for p, p3 in zip(ours, ours3):
        p.grad = torch.randn(*p.shape, generator=g).to(DEV)
        p3.grad = p.grad.clone()
    a.step()
    a3.step()
    for p, p3 in zip(ours, ours3):
        assert torch.equal(p.detach(), p3.detach())


def test_flat_adam_follows_a_scheduler_and_refuses_moved_parameters():
    ours, theirs = _params(6), _params(6)
    a = optim.FlatAdam(ours, lr=5e-4)
    b = torch.optim.Adam(theirs, lr=5e-4)
    sa = torch.optim.lr_scheduler.CosineAnnealingLR(a, T_max=4, eta_min=1e-8)     # the reference's 'cosine' schedule (utils/__init__.py:50-51)
    sb = torch.optim.lr_scheduler.CosineAnnealingLR(b, T_max=4, eta_min=1e-8)
    g = torch.Generator().manual_seed(7)
    for _ in range(4):
        for p, q in zip(ours, theirs):
            p.grad = torch.randn(*p.shape, generator=g).to(DEV)
            q.grad = p.grad.clone()
        a.step()
        b.step()
        sa.step()
        sb.step()
    _close(ours, theirs, "cosine schedule")
    ours[0].data = ours[0].data.clone()
    with pytest.raises(RuntimeError, match="flat buffer"):
        a.step()


def test_get_parameters_flattens_modules_lists_and_dicts():
    """optim.get_parameters = utils/__init__.py:10-22 (what the reference hands its optimiser); FlatAdam over it is the reference's 'adam' branch."""
    lin = torch.nn.Linear(4, 3).to(DEV)
    opt = optim.FlatAdam(optim.get_parameters({"a": lin, "b": [torch.nn.Linear(2, 2).to(DEV)]}), lr=5e-4, eps=1e-8, weight_decay=0.0)
    assert len(opt.param_groups[0]["params"]) == 4 and opt.param_groups[0]["eps"] == 1e-8
    with pytest.raises(ValueError, match="more than once"):
        optim.FlatAdam([lin.weight, lin.weight], lr=1e-3)
