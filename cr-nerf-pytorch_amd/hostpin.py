"""Host-thread placement for launch-bound steps (round 6).

The reference's own training batch (command/train.sh:24: 1,024 rays) is a step of ~340 launches whose device time (~5.2 ms) and host time are
within 10 % of each other; the host side is two Python threads -- the caller and the autograd engine's worker for the device -- that hand the GIL
back and forth some forty times per step.  Left to the scheduler on a 2 x 64-core host they migrate between L3 domains and the hand-overs cross the
fabric: 5.6-6.2 ms per step (7.5 met).  With those two threads inside ONE L3 domain (eight cores and their SMT siblings on this EPYC) the same step
takes 5.3 ms in every run, host enqueue 4.3-4.5 ms: it sits on its device time (profiles/r6/train_1024_host_affinity.txt; which domain, and which
socket, made no difference).  Nothing else measured here moves with it: the headline step enqueues in 0.14 ms against 2.3 ms of device time, the
65,536-ray step in 72 of 186 ms.

    saved = hostpin.pin_step_threads("cuda:0")      # the caller + the autograd worker; everything else stays where it is
    ...training loop...
    hostpin.unpin_host_threads(saved)

Opt-in, because an affinity mask is inherited: threads and DataLoader worker processes the pinned caller starts AFTERWARDS stay in the same domain
unless they widen their own mask (`os.sched_setaffinity(0, hostpin.allowed_before(saved))` in a worker_init_fn) -- start them first.  Pinning
EVERY thread of the process (`pin_host_threads()`, or `taskset -c <domain> python train.py` from outside) gives the same 5.3 ms in a process that
only trains, and 8.3-9.5 ms against 6.3-7.0 in a process that had just run a 16-thread CPU workload (bench.py after its CPU baseline): thread pools
and the step's two threads then share sixteen logical CPUs.  Hence two threads, not all.  Linux only; where the kernel does not say which CPUs
share an L3 the calls change nothing and return None.
"""
import ctypes
import os

_SYS_CPU = "/sys/devices/system/cpu"


def _parse_cpu_list(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def current_cpu():
    """The logical CPU the calling thread runs on (libc's sched_getcpu), or None where there is no such call."""
    try:
        cpu = ctypes.CDLL(None, use_errno=True).sched_getcpu()
    except (OSError, AttributeError):
        return None
    return cpu if cpu >= 0 else None


def l3_domain(cpu, sys_cpu=_SYS_CPU):
    """The logical CPUs that share `cpu`'s last-level cache, or None where sysfs does not say."""
    base = os.path.join(sys_cpu, "cpu%d" % cpu, "cache")
    best = None
    try:
        for idx in os.listdir(base):
            try:
                level = int(open(os.path.join(base, idx, "level")).read())
                shared = _parse_cpu_list(open(os.path.join(base, idx, "shared_cpu_list")).read())
            except (OSError, ValueError):
                continue
            if shared and (best is None or level > best[0]):
                best = (level, shared)
    except OSError:
        return None
    return best[1] if best else None


def _threads():
    try:
        return [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        return [0]


def autograd_thread_id(device):
    """The kernel thread id of the autograd engine's worker for `device` (the thread every backward of a step runs on): found by running a
    one-element backward through a Function that reports where it ran."""
    import threading
    import torch
    seen = []

    class _Where(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            seen.append(threading.get_native_id())
            return g

    x = torch.zeros(1, device=device, requires_grad=True)
    _Where.apply(x).sum().backward()
    return seen[0] if seen else None


def pin_step_threads(device, cpus=None):
    """The recommended call: the calling thread and the autograd worker of `device` into one L3 domain (see the module text)."""
    return pin_host_threads(cpus, threads=[0, autograd_thread_id(device)])


def pin_host_threads(cpus=None, threads=None):
    """Confine threads of this process to one L3 domain: the one the caller runs on, or the given set of logical CPUs.  `threads` None: every
    thread the process has NOW; or a list of kernel thread ids (0 = the caller), e.g. [0, autograd_thread_id(device)] -- the two threads a training
    step alternates between -- which leaves thread pools and the runtime's helpers where they are.  Threads started later by a pinned thread
    inherit its mask.  Returns the state `unpin_host_threads` restores ({thread id: previous mask}, plus the chosen set under the key "cpus"), or
    None when nothing was changed."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    if cpus is None:
        cpu = current_cpu()
        if cpu is None or cpu not in allowed:
            cpu = min(allowed)
        dom = l3_domain(cpu)
        if dom is None:
            return None
        cpus = dom & allowed
    else:
        cpus = set(cpus) & allowed
    if not cpus:
        return None
    saved = {"cpus": frozenset(cpus)}
    import threading
    me = threading.get_native_id()
    for tid in (_threads() if threads is None else [me if t == 0 else int(t) for t in threads if t is not None]):
        try:
            saved[tid] = os.sched_getaffinity(tid)
            os.sched_setaffinity(tid, cpus)
        except OSError:                      # the thread ended in between
            saved.pop(tid, None)
    return saved


def allowed_before(saved):
    """The calling thread's mask before `pin_host_threads` (for worker_init_fn: the union of what the pinned threads were allowed)."""
    out = set()
    for tid, mask in (saved or {}).items():
        if tid != "cpus":
            out |= set(mask)
    return out


def unpin_host_threads(saved):
    """Give every thread `pin_host_threads` touched its previous mask; threads started in between get the union of those masks."""
    if not saved:
        return
    union = allowed_before(saved)
    for tid in _threads():
        try:
            if tid in saved or os.sched_getaffinity(tid) == set(saved["cpus"]):      # pinned by the call, or started since by a pinned thread
                os.sched_setaffinity(tid, saved.get(tid, union))
        except OSError:
            pass
