"""Worker of tests/test_gpu_multiproc.py::test_peer_exchange_*: N ranks sharing cuda:0 (gloo carries the IPC handles),
launched by torch.distributed.run.  Modes: `sums` (many back-to-back reductions of every size against an fp64 reference),
`timeout` (rank 1 skips one reduction: rank 0 must come back with NaN and a named culprit, not hang)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnerf_amd.parallel import PeerExchange  # noqa: E402


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    if mode == "sums":
        ex = PeerExchange(timeout_s=20.0)
        rng = np.random.default_rng(100 + rank)
        sizes = [64, 1024, 1, 65, 1000, 7] * 40          # 240 reductions, both parities, no host sync in between
        mine = [torch.from_numpy((rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)).to(dev) for n in sizes]
        keep = [m.clone() for m in mine]
        for m in mine:
            ex.all_reduce(m)
        ex.check()
        ok = True
        for m, k in zip(mine, keep):
            parts = [torch.zeros(k.numel()) for _ in range(world)]
            dist.all_gather(parts, k.cpu())
            want = parts[0].clone()
            for p in parts[1:]:
                want += p                                  # fp32, rank order: the kernel's order -> bit-exact
            ok = ok and torch.equal(m.cpu(), want)
            alls = [torch.zeros(k.numel()) for _ in range(world)]
            dist.all_gather(alls, m.cpu())
            ok = ok and all(torch.equal(a, alls[0]) for a in alls)   # identical on every rank
        ex.close()
        print("rank %d peer sums exact: %s" % (rank, ok), flush=True)
    elif mode == "latency":     # not a test: same-GPU protocol latency for DESIGN section 4 (no xGMI hop in it)
        ex = PeerExchange(timeout_s=20.0)
        x = torch.ones(1024, device=dev)
        for n in (64, 1024):
            v = x[:n]
            for _ in range(20):
                ex.all_reduce(v)
                v.fill_(1.0)
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                ex.all_reduce(v)
                v.fill_(1.0)
            e1.record()
            torch.cuda.synchronize()
            print("rank %d: %d floats, %d ranks on one GPU: %.1f us per all-reduce + fill" % (rank, n, world, e0.elapsed_time(e1) * 5.0), flush=True)
        ex.check()
        ex.close()
    elif mode == "timeout":
        ex = PeerExchange(timeout_s=0.3)
        x = torch.ones(64, device=dev)
        ex.all_reduce(x)
        ex.check()
        assert float(x[0]) == world
        if rank != 1:
            y = torch.ones(64, device=dev)
            ex.all_reduce(y)                               # rank 1 never joins this one
            try:
                ex.check()
                print("rank %d timeout: NOT detected" % rank, flush=True)
            except RuntimeError as e:
                print("rank %d timeout detected: nan=%s msg=%s" % (rank, bool(torch.isnan(y).all()), "rank 1 did not arrive" in str(e)), flush=True)
        dist.barrier()
        os._exit(0)                                        # the exchange is out of step by design: no collective close
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
