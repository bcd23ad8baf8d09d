"""Wait-state audit of the gfx950 ISA hipcc emits for the translation units that contain hand-written instruction streams.

Why: hipcc pads hazards only between instructions it emitted itself.  An `asm` statement is one opaque instruction to it: nothing inside the
string is padded, an asm VALU write is not seen as a VALU write by the hazard recogniser (so a builtin MFMA reading the register gets at most the
fixed one-state boundary pad), and an asm instruction that READS an MFMA result is not seen as a reader at all.  Whether such a pair is far
enough apart is then a property of ONE build: register pressure or a compiler update moves instructions.  This tool reads the `.s` of a unit
(`hipcc -S --cuda-device-only`, the build's flags), walks every kernel in program order (fall-through + every backward branch once more, so a
loop's tail is checked against its head) and reports every dependent pair with fewer wait states than the table below -- by default only pairs
with at least one side inside `;;#ASMSTART ... ;;#ASMEND` (the ones hipcc does not cover); `--all` checks compiler code too, which is the
tool's self-test: hipcc's own padding must satisfy the same table.

Wait states are counted twice.  "strict": the way LLVM's GCNHazardRecognizer counts them -- every issued instruction is one state, `s_nop N` is
N + 1.  "timed": the same, except that an MFMA cannot issue before the matrix pipe has taken the previous one, i.e. not earlier than `passes`
states after the previous MFMA of the wave issued (a state and a pass are both four cycles) -- a lower bound of the time that has really
passed.  A pair that is short in the timed count is a defect (exit status 1).  A pair that is short only in the strict count has an MFMA
stream between producer and reader that the instruction count does not credit; it is listed ("strict-only") so that a reader can see how the
stream depends on it, and `--strict` turns it into a failure too.

The table (gfx950; CDNA3/4 ISA "manually inserted wait states" + LLVM's GCNHazardRecognizer for gfx940/gfx950, which the platform guide's
section 5.7 summarises):
  valu->mfma   a non-MFMA VALU writes a VGPR/AGPR -> an MFMA reads it as A, B or C                                  2
  partial      a VALU writes PART of a dword (SDWA dst_sel != DWORD, VOP3 op_sel[3]; in asm statements also v_fma_mixlo/mixhi_f16)
               -> a VALU or MFMA reads the register                                                                   1
  rmw          ... -> another partial-dword write of the SAME register (the second one keeps the first one's half)    1
  mfma->any    an XDL MFMA of P passes writes D -> VALU read or write, VMEM / LDS / FLAT read, MFMA A/B read   P + 3 (+ 1 when P != 2)
                                                    -> MFMA C read of an overlapping, not identical range          P + 1 (+ 1 when P != 2)
               (P: 32x32x16 f16/bf16 = 8, 16x16x32 = 4, 32x32x8 = 16, 16x16x16 = 8, 4x4 = 2)
               an fp32 ("SGEMM") MFMA of P passes (16x16x4 = 8, 32x32x2 = 16) -> VALU read or write, memory read               P + 2
                                                    -> any MFMA operand: interlocked on gfx940+                                  0
               an MFMA -> the next MFMA that takes EXACTLY its D as C (accumulate chain)                                         0
  valu->lane   a VALU writes a VGPR -> v_readfirstlane / v_readlane / v_permlane* reads it                            1
  vsgpr->vmem  a VALU writes an SGPR / VCC (v_readfirstlane, v_cmp, carry-out) -> a VMEM / FLAT / LDS-DMA instruction reads it  5
  m0->ldsdma   an SALU writes M0 -> an LDS-DMA instruction (global_load_lds_*, buffer_load_* ... lds)                   1
  store->write a dwordx3/x4 store -> the next instruction overwrites its data registers                                1

    python tools/isa_audit.py unit.s [more.s ...] [--all] [--verbose]        exit status 1 when a pair is short
    python tools/isa_audit.py --compile render_fused_h2.hip ...              compiles csrc/<unit> with build.py's flags first
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cr-nerf-pytorch_amd", "csrc")

# the units whose kernels carry hand-written streams (inline-asm VALU / LDS-DMA / waits next to builtin MFMAs)
AUDITED_UNITS = ["render_fused_h2.hip", "mlp_forward_h2.hip", "mlp_backward_h2.hip", "render_fused_x3.hip", "mlp_forward_x3.hip",
                 "mlp_backward_x3.hip", "mlp_train16.hip", "render_fused16.hip", "mlp_forward16.hip", "render_fused_bf16p.hip",
                 "mlp_forward_bf16p.hip"]

MFMA_PASSES = {  # opcode prefix -> (passes, is_xdl)
    "v_mfma_f32_32x32x16_f16": (8, True), "v_mfma_f32_32x32x16_bf16": (8, True),
    "v_mfma_f32_16x16x32_f16": (4, True), "v_mfma_f32_16x16x32_bf16": (4, True),
    "v_mfma_f32_32x32x8_f16": (16, True), "v_mfma_f32_32x32x8_bf16_1k": (16, True), "v_mfma_f32_32x32x8bf16_1k": (16, True), "v_mfma_f32_32x32x8f16": (16, True),
    "v_mfma_f32_16x16x16_f16": (8, True), "v_mfma_f32_16x16x16_bf16_1k": (8, True), "v_mfma_f32_16x16x16bf16_1k": (8, True), "v_mfma_f32_16x16x16f16": (8, True),
    "v_mfma_f32_16x16x4_f32": (8, False), "v_mfma_f32_16x16x4f32": (8, False),
    "v_mfma_f32_32x32x2_f32": (16, False), "v_mfma_f32_32x32x2f32": (16, False),
    "v_mfma_f32_4x4x1_f32": (2, False), "v_mfma_f32_4x4x1f32": (2, False),
    "v_mfma_f32_16x16x1_f32": (8, False), "v_mfma_f32_32x32x1_f32": (16, False),
}

_MOD = re.compile(r"\s+(?:[a-z_0-9]+:\[[^\]]*\]|[a-z_0-9]+:\S+|glc|slc|nt|sc0|sc1|off|offen|idxen|lds|clamp|tfe|dlc|gds|mul:\d|div:\d|addr64|vcc_lo|row_\w+(?::\d+)?|quad_perm:\[[^\]]*\]|bound_ctrl:\d|wave_\w+(?::\d+)?)(?=\s|$)")
_REG = re.compile(r"^(v|a|s|ttmp)(?:(\d+)|\[(\d+):(\d+)\])$")


def _regs(tok):
    """operand text -> set of ('v'|'a'|'s', n) / ('vcc',) / ('m0',) / ('exec',)"""
    t = tok.strip().lower()
    t = re.sub(r"^(-|\|)+", "", t).rstrip("|")
    m = re.match(r"^(?:neg|abs|sext)\((.*)\)$", t)
    if m:
        t = m.group(1).strip("|- ")
    m = _REG.match(t)
    if m:
        kind = m.group(1)
        if m.group(2) is not None:
            return {(kind, int(m.group(2)))}
        return {(kind, i) for i in range(int(m.group(3)), int(m.group(4)) + 1)}
    if t in ("vcc", "vcc_lo", "vcc_hi"):
        return {("vcc", 0)}
    if t == "m0":
        return {("m0", 0)}
    if t in ("exec", "exec_lo", "exec_hi"):
        return {("exec", 0)}
    return set()


class Ins:
    __slots__ = ("text", "line", "mn", "ops", "mods", "in_asm", "kind", "dst", "src", "cost", "partial", "passes", "xdl", "dst_is_c", "sdst", "lds_dma", "wide_store", "lane_read")

    def __init__(self, text, line, in_asm):
        self.text, self.line, self.in_asm = text, line, in_asm
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.mn = parts[0].lower()
        rest = parts[1] if len(parts) > 1 else ""
        self.mods = " ".join(m.group(0).strip() for m in _MOD.finditer(" " + rest))
        rest = _MOD.sub("", " " + rest).strip()
        self.ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)] if rest else []
        self.cost = 1
        self.partial = False
        self.passes, self.xdl = 0, False
        self.dst, self.src, self.sdst = set(), set(), set()
        self.lds_dma = self.wide_store = self.lane_read = False
        self._classify()

    def _classify(self):
        mn, ops = self.mn, self.ops
        R = [_regs(o) for o in ops]
        allr = set().union(*R) if R else set()
        if mn == "s_nop":
            self.kind = "nop"
            self.cost = int(ops[0], 0) + 1 if ops else 1
            return
        if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
            self.kind = "mfma"
            key = next((k for k in sorted(MFMA_PASSES, key=len, reverse=True) if mn.startswith(k)), None)
            if key is None:
                raise ValueError("isa_audit: unknown MFMA opcode %r (extend MFMA_PASSES)" % mn)
            self.passes, self.xdl = MFMA_PASSES[key]
            self.dst = R[0]
            self.src = set().union(*R[1:]) if len(R) > 1 else set()
            if mn.startswith("v_smfmac"):
                self.src |= R[0]
            return
        if mn.startswith("v_"):
            self.kind = "valu"
            if mn.startswith("v_cmp"):
                if ops and _regs(ops[0]) and next(iter(_regs(ops[0])))[0] in ("s", "vcc") and not mn.endswith("_e32"):
                    self.sdst, self.src = R[0], set().union(*R[1:]) if len(R) > 1 else set()
                else:
                    self.sdst, self.src = {("vcc", 0)}, allr
                if mn.startswith("v_cmpx"):
                    self.sdst = self.sdst | {("exec", 0)}
                return
            if mn.startswith("v_readfirstlane") or mn.startswith("v_readlane"):
                self.sdst, self.src, self.lane_read = R[0], set().union(*R[1:]), True
                return
            if mn.startswith("v_swap"):
                self.dst, self.src = R[0] | R[1], R[0] | R[1]
                return
            has_sdst = "_co_" in mn or mn.startswith("v_div_scale") or mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32")
            self.dst = R[0] if R else set()
            if has_sdst and len(R) > 1 and not mn.endswith("_e32"):
                self.sdst = R[1]
                self.src = set().union(*R[2:]) if len(R) > 2 else set()
            else:
                if has_sdst:
                    self.sdst = {("vcc", 0)}
                self.src = set().union(*R[1:]) if len(R) > 1 else set()
            if mn.startswith("v_permlane") or mn.startswith("v_writelane"):
                self.lane_read = mn.startswith("v_permlane")
                self.src |= R[0]
            if re.match(r"v_(fmac|mac|dot\d+c|pk_fmac)", mn):
                self.src |= R[0]
            # partial-dword writers
            # (v_fma_mixlo / mixhi: LLVM does not list them as forwarding hazards -- hipcc itself emits `v_fma_mixlo_f16 v16, ... ; v_cvt_f32_f16 v48, v16`
            # back to back -- so compiler code is held to LLVM's rule and only the hand-written statements to the stricter one)
            if self.in_asm and (mn.startswith("v_fma_mixlo") or mn.startswith("v_fma_mixhi") or mn.startswith("v_mad_mixlo") or mn.startswith("v_mad_mixhi")):
                self.partial = True
            m = re.search(r"dst_sel:(\w+)", self.mods)
            if m and m.group(1).upper() != "DWORD":
                self.partial = True
            m = re.search(r"(?<![a-z_])op_sel:\[([^\]]*)\]", self.mods)
            if m and not mn.startswith("v_fma_mix") and not mn.startswith("v_pk_") and not mn.startswith("v_dot"):
                bits = [b.strip() for b in m.group(1).split(",")]
                n_src = len(ops) - 1 - (1 if self.sdst and not mn.endswith("_e32") else 0)
                if len(bits) > n_src and bits[-1] == "1":
                    self.partial = True
            return
        if mn.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_")):
            self.kind = "vmem"
            is_lds = "_lds_" in mn or re.search(r"(^|\s)lds($|\s)", self.mods) is not None
            if "_load" in mn and not is_lds:
                self.dst = R[0] if R else set()
                self.src = set().union(*R[1:]) if len(R) > 1 else set()
            else:
                self.src = allr
                self.lds_dma = is_lds
                if "_store" in mn and re.search(r"(dwordx3|dwordx4|b96|b128)", mn):
                    self.wide_store = True
            if "atomic" in mn:
                self.src = allr
            return
        if mn.startswith("ds_"):
            self.kind = "ds"
            if re.match(r"ds_(read|load|bpermute|permute|swizzle|consume|append|ordered)", mn) or "_rtn" in mn:
                self.dst = R[0] if R else set()
                self.src = set().union(*R[1:]) if len(R) > 1 else set()
            else:
                self.src = allr
            return
        if mn.startswith("s_"):
            self.kind = "salu"
            if mn.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_setprio", "s_sleep", "s_sendmsg", "s_setpc", "s_nop", "s_icache", "s_dcache", "s_set_gpr")):
                self.src = allr
            elif mn.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_getreg")):
                self.sdst = R[0] if R else set()
                self.src = set().union(*R[1:]) if len(R) > 1 else set()
            else:
                self.sdst = R[0] if R else set()
                self.src = set().union(*R[1:]) if len(R) > 1 else set()
            return
        self.kind = "other"
        self.src = allr


def parse(path):
    """-> {kernel name: [Ins | ('label', name)]} in program order"""
    kernels, cur, in_asm, name = {}, None, False, None
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            s = raw.strip()
            if not s:
                continue
            if "#ASMSTART" in s:
                in_asm = True
                continue
            if "#ASMEND" in s:
                in_asm = False
                continue
            if s.startswith(";") or s.startswith("//"):
                continue
            m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
            if m:
                lab = m.group(1)
                if not lab.startswith(".L") and not lab.startswith("$"):
                    name, cur = lab, []
                    kernels[name] = cur
                elif cur is not None:
                    cur.append(("label", lab))
                continue
            if s.startswith("."):
                if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".amdhsa_kernel"):
                    cur = None if s.startswith(".section") else cur
                continue
            if cur is None:
                continue
            body = s.split(";")[0].strip()
            if not body:
                continue
            cur.append(Ins(s, ln, in_asm))
    return {k: v for k, v in kernels.items() if any(isinstance(i, Ins) and i.kind == "salu" and i.mn == "s_endpgm" for i in v)}


class Writer:
    __slots__ = ("clock", "tclock", "ins", "cls")

    def __init__(self, clock, tclock, ins, cls):
        self.clock, self.tclock, self.ins, self.cls = clock, tclock, ins, cls       # cls: 'valu' | 'partial' | 'mfma' | 'store-data'


def _need(w, c, reg, as_what):
    """wait states required between writer w (of reg) and consumer c touching reg `as_what` ('read' | 'read-c' | 'write' | 'c-exact'); (n, rule)"""
    wi = w.ins
    if w.cls == "mfma":
        if as_what == "c-exact":
            return 0, "mfma->c"
        P = wi.passes
        if not wi.xdl:                      # fp32 ("SGEMM") producer: interlocked against MFMA consumers on gfx940+, P + 2 for everything else
            return (0, "") if c.kind == "mfma" else (P + 2, "mfma->any")
        if c.kind == "mfma" and as_what == "read-c":
            return P + 1 + (1 if P != 2 else 0), "mfma->any"
        return P + 3 + (1 if P != 2 else 0), "mfma->any"
    if w.cls == "store-data":
        return (1, "store->write") if (as_what == "write" and c.kind in ("valu", "mfma")) else (0, "")
    if as_what == "write":
        if w.cls == "partial" and c.partial:
            return 1, "rmw"
        return 0, ""
    if c.kind == "mfma":
        return 2, "valu->mfma"          # (covers the partial case: 2 >= 1)
    if c.kind == "valu":
        if c.lane_read:
            return 1, "valu->lane"
        if w.cls == "partial":
            return 1, "partial"
    return 0, ""


def audit_kernel(name, items, report_all=False):
    """-> list of (rule, need, have_strict, have_timed, producer Ins, consumer Ins, reg)"""
    found, seen = [], set()
    labels = {it[1]: i for i, it in enumerate(items) if not isinstance(it, Ins)}

    def walk(start, state, sstate, m0, clock, tclock, pipe_free, limit):
        """state: reg -> Writer; sstate: sgpr -> (clock, tclock, Ins) written by a VALU; m0: (clock, tclock, Ins) of the last SALU write; pipe_free: timed
        clock at which the matrix pipe takes the next MFMA; limit: stop after this many states (None: to the end)"""
        t0 = clock
        i = start
        while i < len(items):
            it = items[i]
            i += 1
            if not isinstance(it, Ins):
                continue
            c = it
            if limit is not None and clock - t0 > limit:
                return
            if c.kind == "mfma":
                tclock = max(tclock, pipe_free)
            if c.kind != "nop":
                def hit(rule, need, w_clock, w_tclock, w_ins, reg):
                    have, have_t = clock - w_clock - 1, tclock - w_tclock - 1
                    if have < need and (report_all or w_ins.in_asm or c.in_asm):
                        key = (rule, w_ins.line, c.line, reg)
                        if key not in seen:
                            seen.add(key)
                            found.append((rule, need, have, have_t, w_ins, c, reg))
                exact_c = None
                if c.kind == "mfma" and len(c.ops) >= 4 and _regs(c.ops[3]) == c.dst:
                    exact_c = c.dst
                for reg in c.src:
                    w = state.get(reg)
                    if w is not None:
                        what = "read"
                        if c.kind == "mfma" and len(c.ops) >= 4 and reg in _regs(c.ops[3]) and reg not in set().union(*[_regs(o) for o in c.ops[1:3]]):
                            what = "read-c"
                        if exact_c is not None and reg in exact_c and w.cls == "mfma" and w.ins.dst == c.dst and reg not in set().union(*[_regs(o) for o in c.ops[1:3]]):
                            what = "c-exact"
                        need, rule = _need(w, c, reg, what)
                        if need:
                            hit(rule, need, w.clock, w.tclock, w.ins, reg)
                    if reg[0] in ("s", "vcc") and c.kind == "vmem":
                        sw = sstate.get(reg)
                        if sw is not None:
                            hit("vsgpr->vmem", 5, sw[0], sw[1], sw[2], reg)
                for reg in c.dst:
                    w = state.get(reg)
                    if w is not None and not (c.kind == "mfma" and w.cls == "mfma" and w.ins.dst == c.dst):
                        need, rule = _need(w, c, reg, "write")
                        if need:
                            hit(rule, need, w.clock, w.tclock, w.ins, reg)
                if c.lds_dma and m0 is not None:
                    hit("m0->ldsdma", 1, m0[0], m0[1], m0[2], ("m0", 0))
                # effects
                if c.kind == "mfma":
                    for reg in c.dst:
                        state[reg] = Writer(clock, tclock, c, "mfma")
                    pipe_free = tclock + c.passes
                elif c.kind == "valu":
                    for reg in c.dst:
                        state[reg] = Writer(clock, tclock, c, "partial" if c.partial else "valu")
                    for reg in c.sdst:
                        sstate[reg] = (clock, tclock, c)
                else:
                    for reg in c.dst:
                        state.pop(reg, None)          # memory results: waited for by counters, not by wait states
                    if c.kind == "salu":
                        for reg in c.sdst:
                            sstate.pop(reg, None)
                            if reg == ("m0", 0):
                                m0 = (clock, tclock, c)
                    if c.wide_store:
                        for reg in c.src:
                            if reg[0] in ("v", "a") and reg not in state:
                                state[reg] = Writer(clock, tclock, c, "store-data")
            clock += c.cost
            tclock += c.cost
            if c.kind == "salu":
                if c.mn in ("s_endpgm", "s_branch", "s_setpc_b64"):
                    tgt = c.ops[0] if c.ops else None
                    if c.mn == "s_branch" and tgt in labels and labels[tgt] < i and limit is None:
                        walk(labels[tgt], dict(state), dict(sstate), m0, clock, tclock, pipe_free, 40)
                    state, sstate, m0 = {}, {}, None       # the fall-through is not reached from here
                elif c.mn.startswith("s_cbranch"):
                    tgt = c.ops[-1] if c.ops else None
                    if tgt in labels and labels[tgt] < i and limit is None:     # a loop's back edge: its tail against its head
                        walk(labels[tgt], dict(state), dict(sstate), m0, clock, tclock, pipe_free, 40)

    walk(0, {}, {}, None, 0, 0, 0, None)
    return found


def compile_unit(unit, outdir, extra=()):
    """hipcc -S --cuda-device-only with the flags build.py uses for this unit -> path of the .s"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("crnerf_build_for_audit", os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out = os.path.join(outdir, unit + ".s")
    flags = [f for f in b.FLAGS if f not in ("-fPIC",)] + b.PER_FILE_FLAGS.get(unit, []) + list(extra)
    return out, subprocess.Popen([b._hipcc()] + flags + ["-S", "--cuda-device-only", "-I", CSRC, os.path.join(CSRC, unit), "-o", out],
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def compile_units(units, outdir, extra=(), jobs=None):
    jobs = jobs or max(1, (os.cpu_count() or 2))
    paths, running, todo = {}, [], list(units)
    while todo or running:
        while todo and len(running) < jobs:
            u = todo.pop(0)
            out, p = compile_unit(u, outdir, extra)
            running.append((u, out, p))
        u, out, p = running.pop(0)
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc -S failed on %s:\n%s" % (u, log.decode(errors="replace")[-3000:]))
        paths[u] = out
    return paths


def audit_file(path, report_all=False):
    res = {}
    for name, items in parse(path).items():
        res[name] = audit_kernel(name, items, report_all)
    return res


def summarize(path):
    """per kernel: number of asm statements' instructions by mnemonic, MFMA count -- what the audit looked at"""
    out = {}
    for name, items in parse(path).items():
        ins = [i for i in items if isinstance(i, Ins)]
        asm = {}
        for i in ins:
            if i.in_asm and i.kind != "nop":
                asm[i.mn] = asm.get(i.mn, 0) + 1
        out[name] = {"instructions": len(ins), "mfma": sum(1 for i in ins if i.kind == "mfma"), "asm": asm}
    return out


def split(found):
    """-> (defects: short in the timed count, strict_only: short in the instruction count only)"""
    return [f for f in found if f[3] < f[1]], [f for f in found if f[3] >= f[1]]


def main(argv):
    report_all, verbose, do_compile, strict = "--all" in argv, "--verbose" in argv, "--compile" in argv, "--strict" in argv
    files = [a for a in argv if not a.startswith("--")]
    tmp = None
    if do_compile:
        tmp = tempfile.mkdtemp(prefix="isa_audit_")
        files = list(compile_units(files or AUDITED_UNITS, tmp).values())
    bad = 0
    for path in files:
        res = audit_file(path, report_all)
        summ = summarize(path)
        for name, found in res.items():
            s = summ[name]
            defects, strict_only = split(found)
            print("%s :: %s: %d instructions, %d MFMA, asm %s -> %d short pair(s), %d strict-only" % (
                os.path.basename(path), name, s["instructions"], s["mfma"], dict(sorted(s["asm"].items())), len(defects), len(strict_only)))
            for rule, need, have, have_t, w, c, reg in (found if verbose else (defects[:8] + strict_only[:4])):
                print("    %-12s need %2d have %2d (timed %3d)  %s%d   L%d%s %s   ->   L%d%s %s" % (
                    rule, need, have, have_t, reg[0], reg[1], w.line, "*" if w.in_asm else " ", w.text.split(";")[0].strip()[:70],
                    c.line, "*" if c.in_asm else " ", c.text.split(";")[0].strip()[:70]))
            bad += len(defects) + (len(strict_only) if strict else 0)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
