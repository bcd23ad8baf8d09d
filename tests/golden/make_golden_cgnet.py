"""Generates tests/golden/g13_cgnet.npz by IMPORTING THE REFERENCE in the build container (the transient
mask network, models/lightweight_seg.py:274-368, and its use in train_mask_grid_sample.py:170-176).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cgnet.py

Data only.  Parameters come from tests/_cgnet_fixture.seeded_state (numpy streams keyed by state_dict name), so the file
holds: the input image, the reference's mask in training mode (batch statistics) and in eval mode, the running
statistics after the training-mode forward, the mask read at sampled full-resolution pixels, and -- for the scalar
sum(mask_at_pixels * G) -- each parameter gradient's L2 norm plus its values at 48 fixed positions.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
OUT = os.path.dirname(os.path.abspath(__file__))

from _cgnet_fixture import probe_positions, seeded_state  # noqa: E402
from models.lightweight_seg import Context_Guided_Network  # noqa: E402  (reference)


def main():
    torch.manual_seed(0)
    arrays = {}
    for tag, (H, W), (Hw, Ww), seed in (("a", (44, 60), (350, 478), 3), ("b", (33, 47), (270, 381), 4)):
        net = Context_Guided_Network(classes=1, M=2, N=2, input_channel=3).double()
        seeded_state(net, seed)
        g = torch.Generator().manual_seed(seed)
        img = torch.rand(1, 3, H, W, generator=g)
        idx = torch.randint(0, Hw * Ww, (400,), generator=g)
        G = torch.randn(400, 1, generator=g)
        # --- float32 run: what the reference computes
        net32 = Context_Guided_Network(classes=1, M=2, N=2, input_channel=3)
        seeded_state(net32, seed)
        net32.train()
        mask = net32(img)
        full = torch.nn.functional.interpolate(mask, size=(Hw, Ww), mode="bilinear", align_corners=False)
        picked = full.permute(0, 2, 3, 1).reshape(-1, 1)[idx]
        (picked * G).sum().backward()
        arrays.update({tag + "_img": img, tag + "_idx": idx, tag + "_G": G, tag + "_hw_whole": np.array([Hw, Ww]), tag + "_seed": np.array(seed),
                       tag + "_mask_train": mask.detach(), tag + "_picked": picked.detach()})
        for k, v in net32.state_dict().items():
            if "running" in k or "num_batches" in k:
                arrays[tag + "_stat/" + k] = v.clone()
        # --- float64 run of the same expression: the gradient truth (fp32 autograd of 40 layers carries ~1e-5 noise)
        net.train()
        mask64 = net(img.double())
        full64 = torch.nn.functional.interpolate(mask64, size=(Hw, Ww), mode="bilinear", align_corners=False)
        (full64.permute(0, 2, 3, 1).reshape(-1, 1)[idx] * G.double()).sum().backward()
        for (k, p), (_, p32) in zip(net.named_parameters(), net32.named_parameters()):
            pos = probe_positions(p.shape, k)
            arrays[tag + "_gnorm/" + k] = p.grad.norm()
            arrays[tag + "_gprobe/" + k] = p.grad.reshape(-1)[pos]
            arrays[tag + "_gnorm32/" + k] = p32.grad.norm()
        arrays[tag + "_mask_train64"] = mask64.detach()
        # --- eval mode (running statistics, now updated once)
        net32.eval()
        with torch.no_grad():
            arrays[tag + "_mask_eval"] = net32(img)
    np.savez_compressed(os.path.join(OUT, "g13_cgnet.npz"), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                               for k, v in arrays.items()})
    print("wrote g13_cgnet", sum(np.asarray(v).size for v in arrays.values()), "values")


if __name__ == "__main__":
    main()
