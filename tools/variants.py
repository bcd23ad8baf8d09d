"""Tuning tool (build container): A/B builds of libcrnerf_hip.so that differ in a few translation units.

    python tools/variants.py NAME "-DFLAG ..." unit.hip [unit.hip ...]

compiles the named units with build.py's flags + the extra ones into cr-nerf-pytorch_amd/build_NAME/, links them with the shipped build's other
objects into cr-nerf-pytorch_amd/variants/libcrnerf_NAME.so (git-ignored, travels to the GPU box with the snapshot) -- cross-compiled here, so no
GPU-minute is spent on hipcc.  On the box: CRNERF_LIB_PATH=cr-nerf-pytorch_amd/variants/libcrnerf_NAME.so python tools/<bench>.py"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cr-nerf-pytorch_amd")


def main(name, extra, units):
    spec = importlib.util.spec_from_file_location("crnerf_build_v", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    with open(b.STAMP) as f:                                 # the shipped objects must be current (run build.py first; several variants may then be
        if f.read().strip() != b._digest():                  # built in parallel -- this script never rebuilds the shipped library itself)
            raise SystemExit("cr-nerf-pytorch_amd/build.py first: the shipped library is older than its sources")
    objdir = os.path.join(PKG, "build_" + name)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(PKG, "variants"), exist_ok=True)
    procs = []
    for u in units:
        obj = os.path.join(objdir, u.replace(".hip", ".o"))
        cmd = [b._hipcc()] + b.FLAGS + b.PER_FILE_FLAGS.get(u, []) + extra.split() + ["-I", b.CSRC, "-c", os.path.join(b.CSRC, u), "-o", obj]
        procs.append((u, obj, subprocess.Popen(cmd, cwd=objdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = {}
    for u, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit("hipcc failed on %s:\n%s" % (u, out.decode(errors="replace")[-3000:]))
        objs[u] = obj
    link = [objs.get(s, os.path.join(PKG, "build", s.replace(".hip", ".o"))) for s in b.SOURCES]
    lib = os.path.join(PKG, "variants", "libcrnerf_%s.so" % name)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + link)
    print(lib)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
