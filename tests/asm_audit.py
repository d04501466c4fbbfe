"""Static audit of hand-counted waits in a compiled gfx950 kernel (test infrastructure).

The RHS-block stage kernels (butterfly_amd/csrc/bfhip_stage_mfma.h) issue their fragment loads as `asm volatile` statements and
place every `s_waitcnt` by hand.  hipcc neither counts those loads nor knows that their destination registers are not written
until the data lands: it may schedule a VMEM / LDS instruction of its own between a load and its wait (the count is then off by
one), or read / copy / reuse a destination register early (cdna_hip_programming.md section 5.7 item 1) -- silently wrong
results, on some instantiations only (round 5 met the second: a v_mov of a fragment between its ds_read and the wait).

`audit(asm_text, symbol)` replays the kernel's instruction stream with the hardware's two in-order counters -- vmcnt (vector
loads AND stores on gfx9) and lgkmcnt (LDS) -- and reports every instruction that touches a vector register some load issued
before it may still be writing.  Straight-line model of every innermost MFMA loop: replayed from the last full drain of both
counters ahead of the loop (so that its prologue requests are in the queues), the body three times (a load at the bottom of an
iteration met by a reader at the top of the next).  s_waitcnt retires all but the n youngest of a queue; scalar loads (which
return out of order) are counted conservatively as entries that only lgkmcnt(0) retires."""
import re

_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_BRANCH = re.compile(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def function_body(asm_text, symbol):
    body = asm_text[asm_text.index(symbol + ":"):]
    body = body[:body.index("s_endpgm")]
    lines = []
    for raw in body.split("\n")[1:]:
        ln = raw.split(";")[0].strip()
        if not ln or ln.startswith(".") and not _LABEL.match(ln):
            continue
        lines.append(ln)
    return lines


def innermost_loops(lines, must_contain="v_mfma"):
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [_LABEL.match(ln)] if m}
    loops = []
    for i, ln in enumerate(lines):
        m = _BRANCH.match(ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            if any(must_contain in x for x in lines[a:i + 1]):
                loops.append((a, i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    return inner


class _State:
    def __init__(self):
        self.vm, self.lgkm, self.problems = [], [], []

    def pending(self):
        regs = set()
        for q in (self.vm, self.lgkm):
            for dst, _ in q:
                regs |= dst
        return regs

    def step(self, ln, where):
        op = ln.split()[0]
        if _LABEL.match(ln):
            return
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ln)
            if m:
                n = int(m.group(1))
                del self.vm[:max(0, len(self.vm) - n)]
            m = re.search(r"lgkmcnt\((\d+)\)", ln)
            if m:
                n = int(m.group(1))
                if n == 0:
                    self.lgkm.clear()
                else:          # scalar entries may return out of order: a counted wait retires in-order LDS entries only up to the first scalar one
                    while len(self.lgkm) > n and not self.lgkm[0][1]:
                        self.lgkm.pop(0)
            if not re.search(r"vmcnt|lgkmcnt|expcnt", ln):      # raw immediate: treat as a full wait
                self.vm.clear(); self.lgkm.clear()
            return
        ops = ln[len(op):]
        regs = _vregs(ops)
        busy = self.pending() & regs
        if busy:
            self.problems.append((where, ln, sorted(busy)))
        first = ops.split(",")[0]
        if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
            self.vm.append((set() if re.search(r"\blds\b", ln) else _vregs(first), False))
        elif op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic", "flat_atomic")):
            self.vm.append((set(), False))
        elif op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
            self.lgkm.append((_vregs(first), False))
        elif op.startswith("ds_"):
            self.lgkm.append((set(), False))
        elif op.startswith(("s_load", "s_buffer_load")):
            self.lgkm.append((set(), True))


def audit(asm_text, symbol, loop_must_contain="v_mfma"):
    """[(line index, instruction, registers still pending)] inside the innermost loops of `symbol`; [] = every asm load's
    registers are left alone until waited for.  Each loop is replayed from the last full drain of both counters ahead of it (its
    prologue requests are then in the queues), the body three times."""
    lines = function_body(asm_text, symbol)
    problems = []
    for a, b in innermost_loops(lines, loop_must_contain):
        pv = max([i for i in range(a) if re.search(r"s_waitcnt.*vmcnt\(0\)", lines[i])] or [0])
        pl = max([i for i in range(a) if re.search(r"s_waitcnt.*lgkmcnt\(0\)", lines[i])] or [0])
        st = _State()
        for i in range(min(pv, pl), a):
            st.step(lines[i], i)
        st.problems = []
        for _ in range(3):
            for i in range(a, b + 1):
                st.step(lines[i], i)
        problems += st.problems
    return problems


def loop_vmem(asm_text, symbol, loop_must_contain="v_mfma"):
    """For every innermost MFMA loop: (number of MFMAs, sorted mnemonics of its vector-memory instructions)."""
    lines = function_body(asm_text, symbol)
    out = []
    for a, b in innermost_loops(lines, loop_must_contain):
        seg = lines[a:b + 1]
        vm = sorted(x.split()[0] + (" lds" if re.search(r"\blds\b", x) else "") for x in seg if re.match(r"(buffer_|global_|flat_|scratch_)", x))
        out.append((sum("v_mfma" in x for x in seg), vm))
    return out
