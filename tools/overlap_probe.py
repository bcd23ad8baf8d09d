"""Measurement tool (GPU box): does the write-bound h2 data gradient of one ray chunk run faster beside the read-bound f16x2 weight gradients
of the previous chunk than behind them?  (VERDICT r5 "do this" #1a)  Both kernels take a whole CU per workgroup (512 registers per
wave), so side by side means a partition of the CUs: plain second stream, and CU-share streams (crnerf_stream_create_cu_share) at several splits.

    python tools/overlap_probe.py [P=1048576] [units=6]

Every variant runs the SAME work: `units` backward passes over P points (data gradient + the thirteen weight-gradient jobs + reductions) with two
alternating scratch buffers; gradients of every variant are compared bit for bit with the one-stream run."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import crnerf_amd.synth as synth
from crnerf_amd import _lib, ops

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
U = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
lib = _lib.load()
st = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
packed = ops.pack_mlp_weights(st)
packed_th = ops.pack_mlp_weights_t_h2(st)
torch.manual_seed(0)
x = torch.rand(P, 120, device=dev) * 2 - 1
d_out = torch.randn(P, 65, device=dev) * 1e-3
out, acts = ops.mlp_forward_train(packed, x)
nbytes = lib.crnerf_mlp_train_scratch_bytes(P)
scratch = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
grads = [[torch.empty(s, dtype=torch.float32, device=dev) for s in ops.MLP_TENSOR_SHAPES] for _ in range(2)]
kw = dict(wgrad_bf16="f16x2", dgrad_h2=True)


def make_stream(first=None, count=None):
    if first is None:
        return torch.cuda.Stream(), None
    h = ctypes.c_void_p()
    _lib.check(lib.crnerf_stream_create_cu_share(ctypes.byref(h), first, count), "crnerf_stream_create_cu_share")
    return torch.cuda.ExternalStream(h.value), h


def run_serial(units):
    for u in range(units):
        ops.mlp_backward(packed_th, x, out, d_out, acts, scratch=scratch[u & 1], grads=grads[u & 1], **kw)


def run_phases_one_stream(units):
    for u in range(units):
        ops.mlp_backward(packed_th, None, out, d_out, acts, phase="dgrad", scratch=scratch[u & 1], **kw)
        ops.mlp_backward(None, x, None, None, acts, phase="wgrad", scratch=scratch[u & 1], grads=grads[u & 1], **kw)


def run_overlap(units, sd, sw):
    """data gradient of unit u + 1 on stream sd beside the weight gradients of unit u on stream sw"""
    cur = torch.cuda.current_stream()
    sd.wait_stream(cur)
    sw.wait_stream(cur)
    w_done = [None, None]
    for u in range(units):
        with torch.cuda.stream(sd):
            if w_done[u & 1] is not None:
                sd.wait_event(w_done[u & 1])          # the scratch's previous weight gradients have read it
            ops.mlp_backward(packed_th, None, out, d_out, acts, phase="dgrad", scratch=scratch[u & 1], **kw)
            ev = torch.cuda.Event()
            ev.record(sd)
        with torch.cuda.stream(sw):
            sw.wait_event(ev)
            ops.mlp_backward(None, x, None, None, acts, phase="wgrad", scratch=scratch[u & 1], grads=grads[u & 1], **kw)
            w_done[u & 1] = torch.cuda.Event()
            w_done[u & 1].record(sw)
    cur.wait_stream(sd)
    cur.wait_stream(sw)


def timed(fn, reps=3):
    fn(2)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn(U)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / U)
    return best * 1e3


def only(phase, stream=None, reps=3):
    def fn(units):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for u in range(units):
                if phase == "dgrad":
                    ops.mlp_backward(packed_th, None, out, d_out, acts, phase="dgrad", scratch=scratch[u & 1], **kw)
                else:
                    ops.mlp_backward(None, x, None, None, acts, phase="wgrad", scratch=scratch[u & 1], grads=grads[u & 1], **kw)
        if stream is not None:
            torch.cuda.current_stream().wait_stream(stream)
    return timed(fn, reps)


print("P = %d points, %d units per timing, best of 3; per unit" % (P, U), flush=True)
t_ser = timed(run_serial)
ref = [g.clone() for g in grads[(U - 1) & 1]]
print("one call per unit (today)                     %7.3f ms" % t_ser, flush=True)
t_ph = timed(run_phases_one_stream)
same = all(torch.equal(a, b) for a, b in zip(ref, grads[(U - 1) & 1]))
print("two phase calls, one stream                   %7.3f ms   gradients bit-equal: %s" % (t_ph, same), flush=True)
t_d, t_w = only("dgrad"), only("wgrad")
print("data gradient alone %7.3f ms, weight gradients alone %7.3f ms (sum %7.3f)" % (t_d, t_w, t_d + t_w), flush=True)
s1, _ = make_stream()
s2, _ = make_stream()
t_ov = timed(lambda n: run_overlap(n, s1, s2))
same = all(torch.equal(a, b) for a, b in zip(ref, grads[(U - 1) & 1]))
print("two plain streams                             %7.3f ms   gradients bit-equal: %s" % (t_ov, same), flush=True)
per = lib.crnerf_cus_per_xcd()
print("CU-share streams (CUs per XCD: %d)" % per, flush=True)
for a in (per // 2, per * 3 // 8, per * 5 // 8, per // 4, per * 3 // 4):
    sd, hd = make_stream(0, a)
    sw, hw = make_stream(a, per - a)
    t_da, t_wa = only("dgrad", sd), only("wgrad", sw)
    t = timed(lambda n: run_overlap(n, sd, sw))
    same = all(torch.equal(x_, y_) for x_, y_ in zip(ref, grads[(U - 1) & 1]))
    print("  data gradient on %2d / weight gradients on %2d CUs per XCD: %7.3f ms per unit (alone on their shares: %7.3f / %7.3f)   bit-equal: %s"
          % (a, per - a, t, t_da, t_wa, same), flush=True)
    torch.cuda.synchronize()
    lib.crnerf_stream_destroy(hd)
    lib.crnerf_stream_destroy(hw)
# overlapping CU sets: both streams may use every CU, but each kernel's grid is what its share allows -- the dispatcher decides
sd, hd = make_stream(0, per)
sw, hw = make_stream(0, per)
t = timed(lambda n: run_overlap(n, sd, sw))
print("  both on all CUs (two masked streams, full masks): %7.3f ms" % t, flush=True)
