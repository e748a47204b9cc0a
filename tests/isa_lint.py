"""ISA lint for the kernels that read fragments / operands with INLINE-ASSEMBLY loads behind hand-counted `s_waitcnt`.

hipcc does not know that the destination of an inline-assembly `ds_read_*` / `global_load_*` is not valid until the counted
wait the source placed behind it: it may copy such a register (phi resolution: `v_mov_b64`, `v_accvgpr_write`), spill it, or hoist
an instruction that reads it in front of that wait.  Every op-level test passes (alone, the read has long landed); in the train
step it is a run-to-run difference (round 4, `wgradr`, profiles/r04_wgradr_soak_bisect.txt).

`lint(text, kernel_regex)` walks the gfx950 assembly of every matching kernel ALONG ITS CONTROL FLOW (a forward dataflow over the basic
blocks; where paths join, a register stays pending if it is pending on either) and models the two in-order counters:

  * lgkmcnt queue: every `ds_*` (and scalar-memory) instruction in program order;
  * vmcnt queue:   every vector-memory instruction (loads, stores, atomics, LDS-DMA) in program order;
  * `s_waitcnt lgkmcnt(N)` / `vmcnt(N)` retires all but the N youngest entries of its queue;
  * the destination registers of an inline-assembly load (between `;;#ASMSTART` / `;;#ASMEND`) are PENDING until their entry retires.

Any instruction that names a pending register — as a source (use before the wait, the compiler copy) or as a destination (clobber
under a load in flight) — is a finding.  Test infrastructure only (CPU, no GPU needed)."""
from __future__ import annotations

import re
from dataclasses import dataclass

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_WAIT = re.compile(r"(vmcnt|lgkmcnt)\((\d+)\)")
_VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load",
         "scratch_store", "flat_load", "flat_store", "flat_atomic")
_LGKM = ("ds_", "s_load_", "s_buffer_load_", "s_store_")
_EMPTY = frozenset()
_VMCAP, _LGCAP = 63, 15                       # the hardware counters: vmcnt has 6 bits, lgkmcnt 4 (issue stalls beyond)


def _regs(s: str) -> frozenset:
    out = set()
    for m in _REG.finditer(s):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return frozenset(out)


@dataclass
class Finding:
    kernel: str
    line: int
    text: str
    regs: tuple

    def __str__(self):
        return f"{self.kernel}: line {self.line}: `{self.text.strip()}` touches in-flight {sorted(self.regs)}"


@dataclass
class _Ins:
    line: int
    text: str
    op: str
    regs: frozenset
    kind: str            # "", "wait", "vm", "lgkm", "branch", "cbranch", "end"
    dest: frozenset      # pending destination (inline-assembly loads only)
    waits: tuple         # ((queue, keep), ...)
    target: str


def _functions(text: str, kernel_regex: str):
    for m in re.finditer(r"\n(_Z\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, flags=re.S):
        if re.search(kernel_regex, m.group(1)):
            yield m.group(1), m.group(2).split("\n")


def _parse(lines):
    ins, labels, in_asm = [], {}, False
    for i, raw in enumerate(lines):
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        m = _LABEL.match(raw)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        code = raw.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op = code.split()[0]
        operands = code[len(op):]
        kind, dest, waits, target = "", _EMPTY, (), ""
        if op == "s_waitcnt":
            kind, waits = "wait", tuple((k, int(n)) for k, n in _WAIT.findall(code))
        elif op == "s_endpgm":
            kind = "end"
        elif op == "s_branch":
            kind, target = "branch", operands.strip()
        elif op.startswith("s_cbranch"):
            kind, target = "cbranch", operands.strip()
        elif op.startswith(_VMEM) or op.startswith(_LGKM):
            kind = "vm" if op.startswith(_VMEM) else "lgkm"
            is_load = ("load" in op or "read" in op) and " lds" not in code and "_lds_" not in op
            if in_asm and is_load and op.startswith(("ds_read", "ds_load", "global_load", "buffer_load")):
                dest = _regs(operands.split(",")[0])
        ins.append(_Ins(i, raw, op, _regs(operands) if kind != "wait" else _EMPTY, kind, dest, waits, target))
    return ins, labels


def _merge(a, b):
    """position-wise union of two queues aligned at their YOUNG end: a register pending at distance d from the end on either path
    stays pending until a wait that retires distance d on both (conservative join of the two paths)"""
    if a == b:
        return a
    n = max(len(a), len(b))
    pa = (_EMPTY,) * (n - len(a)) + tuple(a)
    pb = (_EMPTY,) * (n - len(b)) + tuple(b)
    return _canon(tuple(x | y for x, y in zip(pa, pb)))


def _canon(q):
    k = 0
    while k < len(q) and not q[k]:
        k += 1
    return tuple(q[k:])


def lint(text: str, kernel_regex: str):
    """-> (findings, {kernel name: number of inline-assembly loads in its text})"""
    findings, per_kernel = [], {}
    for name, lines in _functions(text, kernel_regex):
        ins, labels = _parse(lines)
        per_kernel[name] = sum(1 for x in ins if x.dest)
        leaders = set(labels.values()) | {0}
        for k, x in enumerate(ins):
            if x.kind in ("branch", "cbranch", "end"):
                leaders.add(k + 1)
        seen_lines = set()
        state = {0: ((), ())}                   # block leader -> (lgkm queue, vm queue) on entry, joined over its predecessors
        work = [0]

        def flow(to, lg, vm):
            new = (_canon(lg), _canon(vm))
            old = state.get(to)
            if old is not None:
                new = (_merge(old[0], new[0]), _merge(old[1], new[1]))
                if new == old:
                    return
            state[to] = new
            work.append(to)

        while work:
            pc = work.pop()
            lg, vm = (list(q) for q in state[pc])
            while pc < len(ins):
                x = ins[pc]
                if x.kind == "wait":
                    for kq, keep in x.waits:
                        q = vm if kq == "vmcnt" else lg
                        del q[: max(0, len(q) - keep)]
                else:
                    if x.regs and (lg or vm):
                        pend = set()
                        for e in lg:
                            pend |= e
                        for e in vm:
                            pend |= e
                        hit = x.regs & pend
                        if hit and x.line not in seen_lines:
                            seen_lines.add(x.line)
                            findings.append(Finding(name, x.line, x.text, tuple(sorted(hit))))
                    if x.kind == "vm":
                        vm.append(x.dest)
                        del vm[: max(0, len(vm) - _VMCAP)]
                    elif x.kind == "lgkm":
                        lg.append(x.dest)
                        del lg[: max(0, len(lg) - _LGCAP)]
                    elif x.kind == "end":
                        break
                    elif x.kind == "branch":
                        flow(labels[x.target], lg, vm)
                        break
                    elif x.kind == "cbranch":
                        flow(labels[x.target], lg, vm)
                pc += 1
                if pc in leaders:
                    flow(pc, lg, vm)
                    break
    return findings, per_kernel


_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _sregs(text: str) -> set:
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_vmem_sgpr_hazard(text: str, kernel_regex: str, wait_states: int = 5):
    """gfx9: a vector-memory instruction that READS an SGPR needs `wait_states` wait states behind a VALU instruction that WROTE it
    (v_readlane / v_readfirstlane / v_cmp ... — LLVM's VmemSgprWaitStates).  hipcc pads the instructions it emits itself, never the
    inside of an asm string: with SGPRs spilled to VGPR lanes, the scalar base of an inline-assembly load can come out of a
    v_readlane one instruction earlier (round 5: a memory fault on ragged fp8 tiles).  Walks each kernel's text in program order
    (a label does not reset the distances: a fall-through is the short path) and reports every INLINE-ASSEMBLY vector-memory
    instruction whose scalar operand is younger than that.  -> (findings, number of asm vector-memory instructions seen)"""
    findings, seen = [], 0
    for name, lines in _functions(text, kernel_regex):
        age = {}                                   # SGPR -> wait states since its last VALU write
        in_asm = False
        for i, raw in enumerate(lines):
            if ";;#ASMSTART" in raw:
                in_asm = True
                continue
            if ";;#ASMEND" in raw:
                in_asm = False
                continue
            code = raw.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            op = code.split()[0]
            operands = code[len(op):]
            if in_asm and op.startswith(_VMEM):
                seen += 1
                young = sorted(r for r in _sregs(operands) if age.get(r, 99) < wait_states)
                if young:
                    findings.append(Finding(name, i, raw, tuple(("s", r) for r in young)))
            step = 1
            if op == "s_nop":
                try:
                    step = int(operands.strip(), 0) + 1
                except ValueError:
                    step = 1
            for r in list(age):
                age[r] += step
            if op.startswith("v_") and not op.startswith("v_mfma"):
                first = operands.split(",")[0]
                for r in _sregs(first):            # VALU with a scalar destination (v_readlane, v_readfirstlane, v_cmp_* e64)
                    age[r] = 0
    return findings, seen
