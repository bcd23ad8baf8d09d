"""The "f32x3" entry points (include/crnerf.h): NeRF_sigma.forward (models/nerf.py:157-182) in fp32 on the bf16 matrix cores -- every fp32
operand of the eleven nn.Linear split into three bf16 pieces, a product = the six leading piece products, fp32 accumulation.  Held to the SAME
goldens and tolerances as the fp32 entry points (tests/test_gpu_parity.py), and against a float64 evaluation: the split path must sit where the
fp32 matrix cores sit."""
import numpy as np
import pytest
import torch

import crnerf_amd.synth as synth
from crnerf_amd import ops
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def C(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, atol, rtol=0.0):
    torch.testing.assert_close(got.detach().float().cpu(), torch.as_tensor(want).float(), atol=atol, rtol=rtol)


def test_mlp_x3_golden(golden):
    g = golden("g2_mlp")
    x = C(g["x"])
    for tag, atol, rtol in (("default", 1e-6, 0.0), ("peaky", 3e-5, 1e-5)):      # the fp32 entry point's bars (test_mlp_golden)
        pk = ops.pack_mlp_weights_x3({k: C(v) for k, v in synth.mlp_state(int(g["seed_" + tag]), float(g["gain_" + tag])).items()})
        close(ops.mlp_forward_x3(pk, x), g["out_" + tag], atol=atol, rtol=rtol)
        close(ops.mlp_forward_x3(pk, x[:, :93].contiguous(), sigma_only=True), g["sigma_" + tag], atol=atol, rtol=rtol)


@pytest.mark.parametrize("n", [1, 31, 33, 127, 129, 4099])
def test_mlp_x3_vs_oracle_ragged_sizes(n):
    st = synth.mlp_state(11, 2.0, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-3, 3, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    w = {k: torch.from_numpy(v) for k, v in st.items()}
    close(ops.mlp_forward_x3(ops.pack_mlp_weights_x3({k: C(v) for k, v in st.items()}), x.to(DEV)), O.mlp_forward(w, x), atol=2e-5, rtol=1e-5)


def test_mlp_x3_detects_transposed_or_permuted_packing():
    """One-hot input rows reproduce single columns of W1 / of the dir layer: catches any row / column / slot / piece-order error in fragX."""
    st = synth.mlp_state(3, 1.0)
    w = {k: torch.from_numpy(v) for k, v in st.items()}
    x = torch.zeros(120, 120)
    x[torch.arange(120), torch.arange(120)] = 1.0
    close(ops.mlp_forward_x3(ops.pack_mlp_weights_x3({k: C(v) for k, v in st.items()}), x.to(DEV)), O.mlp_forward(w, x), atol=1e-6)


@pytest.mark.parametrize("gain", [1.0, 2.0, 3.0])
def test_mlp_x3_is_as_accurate_as_the_fp32_matrix_cores(gain):
    """Against the oracle's MLP in float64 on the same fp32 inputs and weights: max / mean error of the x3 path vs the fp32-MFMA path."""
    n = 20000
    st = synth.mlp_state(29, gain, 0.5)
    g = torch.Generator().manual_seed(int(gain * 10))
    x = torch.cat([O.posenc(torch.rand(n, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(n, 3, generator=g) * 2 - 1, 4)], 1).to(DEV)
    dev = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        ref = O.mlp_forward({k: v.double() for k, v in dev.items()}, x.double())
        o32 = ops.mlp_forward(ops.pack_mlp_weights(dev), x).double()
        ox3 = ops.mlp_forward_x3(ops.pack_mlp_weights_x3(dev), x).double()
    e32, ex3 = (o32 - ref).abs(), (ox3 - ref).abs()
    print("gain %.1f: fp32 MFMA max %.3e mean %.3e | x3 max %.3e mean %.3e | x3 vs fp32 MFMA max %.3e" %
          (gain, float(e32.max()), float(e32.mean()), float(ex3.max()), float(ex3.mean()), float((ox3 - o32).abs().max())))
    assert float(ex3.mean()) <= 1.5 * float(e32.mean()) + 1e-9 and float(ex3.max()) <= 2.0 * float(e32.max()) + 1e-7
