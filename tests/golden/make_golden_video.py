"""Generates tests/golden/g12_video.npz: the test-time camera path and intrinsics of the reference's video script
(appearance_modification_video.py:104-189), by executing THOSE function definitions (the module itself cannot be
imported here: imageio / torchvision / Lightning-era helpers are absent) on a stub dataset object.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_video.py
"""
import math
import os
import types

import numpy as np

SRC = "/root/reference/appearance_modification_video.py"
OUT = os.path.dirname(os.path.abspath(__file__))

text = open(SRC).read()
body = text[text.index("def eulerAnglesToRotationMatrix"):text.index('if __name__ == "__main__":')]
ns = {"np": np, "math": math, "args": types.SimpleNamespace(img_wh=[320, 240])}
exec(compile(body, SRC, "exec"), ns)      # runs the reference's own definitions; nothing of it is stored

out = {}
for name, fn in (("gate", ns["define_poses_brandenburg_gate"]), ("fountain", ns["define_poses_trevi_fountain"])):
    ds = types.SimpleNamespace()
    fn(ds)
    out["poses_" + name] = ds.poses_test
for wh in ((320, 240), (800, 800)):
    ns["args"].img_wh = list(wh)
    ds = types.SimpleNamespace()
    ns["define_camera"](ds)
    out["K_%dx%d" % wh] = ds.test_K
np.savez_compressed(os.path.join(OUT, "g12_video.npz"), **out)
print({k: v.shape for k, v in out.items()})
