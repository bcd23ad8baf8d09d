"""Builds libcrnerf_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain
C-ABI shared object (include/crnerf.h) loaded through ctypes by _lib.py.

    python cr-nerf-pytorch_amd/build.py [--force] [--asm]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrnerf_hip.so")
STAMP = os.path.join(HERE, ".build_stamp")
SOURCES = ["abi.hip", "pack.hip", "mlp_forward16.hip", "render_fused16.hip", "mlp_forward_bf16p.hip", "mlp_forward_x3.hip", "mlp_forward_h2.hip", "render_fused_h2.hip", "mlp_backward_x3.hip", "mlp_backward_h2.hip", "render_fused_bf16p.hip", "render_fused_x3.hip", "mlp_train16.hip", "mlp_gemm_bf16.hip", "train_aux.hip", "ray_kernels.hip", "raygen.hip", "encoder.hip", "encoder_train.hip", "cgnet.hip", "cgnet_chain.hip",
           "crossray.hip", "peer_xchg.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/crnerf.h"]   # every header: any edit rebuilds
# -ffp-contract=off: the reference evaluates o + d*z, near*(1-s) + far*s, ... as separate mul/add;
# the kernels call fmaf() explicitly wherever a fused multiply-add is wanted.
# -fno-honor-nans: lets fmaxf(x, 0) be ONE v_max_f32 (otherwise hipcc canonicalises the MFMA output first).
FLAGS = (["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-honor-nans", "-Wall", "-Wno-unused-function"]
         + os.environ.get("CRNERF_EXTRA_FLAGS", "").split())   # tuning builds only, e.g. -DCRNERF_TIMING


# -fno-slp-vectorize (bf16 units): hipcc otherwise packs the epilogue's scalar adds into v_pk_add_f32 bundles placed
# at the END of a layer -- the hand-interleaved epilogue collapses into a serial VALU burst behind the MFMAs.
PER_FILE_FLAGS = {# pragma-unroll-threshold: the tile loop of the 22-k-step layer is "too large" for `#pragma unroll` at the default 16k,
                  # and every ring constant of the pair core depends on full unrolling
                  "render_fused_bf16p.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"],
                  "mlp_forward_bf16p.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"],
                  # the unit tools/isa_audit.py flags for SGPR spill reloads in front of its LDS-DMA statements (csrc/mlp_core.h "HAZARD")
                  "mlp_forward_h2.hip": ["-DCRNERF_GLDS_SALU_COPY"]}
for _kv in os.environ.get("CRNERF_EXTRA_FILE_FLAGS", "").split(";"):      # tuning builds only: "mlp_train16.hip=-fno-slp-vectorize -DX;other.hip=..."
    if "=" in _kv:
        PER_FILE_FLAGS.setdefault(_kv.split("=", 1)[0].strip(), []).extend(_kv.split("=", 1)[1].split())


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, keep_asm=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link the shared library. Returns its path."""
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == digest:
                return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if keep_asm:
            cmd += ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, cwd=objdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out.strip():
            print(out.decode(errors="replace"))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, keep_asm="--asm" in sys.argv))
