"""Debug aid (GPU box): per-layer deltas of the mixed-precision backward against the oracle's autograd hooks."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crnerf_amd.synth as synth
from crnerf_amd import ops, _lib
from oracle import cpu_ref as O
import torch.nn.functional as F

n, gain = int(sys.argv[1]) if len(sys.argv) > 1 else 512, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
DEV = torch.device("cuda", 0)
st = synth.mlp_state(17, gain, 0.5)
rng = np.random.default_rng(n)
T = lambda a: torch.from_numpy(a)
x = torch.cat([O.posenc(T(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15), O.posenc(T(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
d_out = T(rng.normal(size=(n, 65)).astype(np.float32))
w = {k: T(v).clone().requires_grad_(True) for k, v in st.items()}
hooks = {}
lin = lambda h, name: O._LinearBf16.apply(h, w[name + ".weight"], w[name + ".bias"])
xyz = x[:, :93]; h = xyz; pre = []
for layer in range(1, 9):
    if layer == 5:
        h = torch.cat((xyz, h), dim=1)
    z = lin(h, "xyz_encoding_%d.0" % layer); z.retain_grad(); pre.append(z)
    h = F.relu(z)
sigma = F.softplus(O._SigmaHeadStoredBf16.apply(h, w["static_sigma.0.weight"], w["static_sigma.0.bias"]))
final = lin(h, "xyz_encoding_final"); final.retain_grad()
zd = lin(torch.cat((final, x[:, 93:]), dim=1), "dir_encoding.0"); zd.retain_grad()
feat = torch.sigmoid(lin(F.relu(zd), "static_rgb.0"))
ref = torch.cat((feat, sigma), -1)
(ref * d_out).sum().backward()
lib = _lib.load()
packed, tensors = ops.pack_mlp_weights_mixed({k: T(v).to(DEV) for k, v in st.items()})
xd = x.to(DEV)
out, acts = ops.mlp_forward_train_mixed(packed, tensors, xd)
grads = [torch.empty(s, dtype=torch.float32, device=DEV) for s in ops.MLP_TENSOR_SHAPES]
scratch = torch.zeros(lib.crnerf_mlp_train_mixed_scratch_bytes(n), dtype=torch.uint8, device=DEV)
_lib.check(lib.crnerf_mlp_backward_mixed_f32(_lib.ptr_array(tensors, "t"), ctypes.c_void_p(packed.data_ptr()), _lib.dev_ptr(xd), _lib.dev_ptr(out),
                                             _lib.dev_ptr(d_out.to(DEV)), ctypes.c_void_p(acts.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                                             _lib.ptr_array(grads, "g"), n, _lib.stream_ptr()), "bwd")
torch.cuda.synchronize()
pos = np.arange(256); featidx = (pos & ~31) | (((pos >> 2) & 1) << 4) | (((pos >> 3) & 3) << 2) | (pos & 3)   # position -> feature
def rows(buf, slot):
    a = buf[: 10 * n * 512].view(torch.bfloat16).view(10, n, 256)[slot].float().cpu()
    o = torch.empty_like(a); o[:, featidx] = a
    return o
for slot, z in list(enumerate(pre)) + [(8, final), (9, zd)]:
    want = O.bf16_round(z.grad)
    got = rows(scratch, slot)[:, : want.shape[1]]
    act_w = O.bf16_round(F.relu(z) if slot != 8 else z).detach()
    act_g = rows(acts, slot)[:, : want.shape[1]]
    print("slot %d: delta rel-L2 %.3e (max %.3e of %.3e)   act rel-L2 %.3e" % (slot, float((got - want).norm() / want.norm()), float((got - want).abs().max()),
          float(want.abs().max()), float((act_g - act_w).norm() / act_w.norm())))
for name, g in zip(ops.MLP_TENSOR_NAMES, grads):
    r = w[name].grad
    print("%-28s rel-L2 %.3e" % (name, float((g.cpu() - r).norm() / (r.norm() + 1e-30))))
