"""Audit of a -save-temps .s: between an inline-asm ds_read and the inline-asm s_waitcnt lgkmcnt(N) that retires it,
no compiler-generated instruction may read or write its destination registers (the data lands asynchronously;
hipcc treats the destination as written at ;;#ASMEND).  The same for inline-asm global_load_* destinations, which stay
in flight until the inline-asm wait marked `; XOP_FENCE` (qkv_attention.hip).

Control flow IS followed (round 4): the in-flight queues travel along every branch edge -- conditional branches are walked
both ways, unconditional ones to their target -- so a load requested at the bottom of a loop and fenced at its head
(qkv_attention.hip's operand prefetch across the back-edge) has the loop-head code checked too.  A walk carried over an
edge ends when its queues are empty or when the same (place, queues) state has been seen.

usage: audit_asm_reads.py file.s [kernel-symbol-substring] [-v]"""
import re
import sys

path = sys.argv[1]
want = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
verbose = "-v" in sys.argv
src = open(path).read().split("\n")

LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
labels = {}
for i, l in enumerate(src):
    m = LABEL.match(l)
    if m:
        labels[m.group(1)] = i

# kernels = function symbols (not .L labels); a walk never leaves the function it started in
func_of = [None] * len(src)
cur = None
for i, l in enumerate(src):
    m = LABEL.match(l)
    if m and not m.group(1).startswith("."):
        cur = m.group(1)
    if l.startswith(".Lfunc_end"):
        func_of[i] = cur
        cur = None
        continue
    func_of[i] = cur


def regs_of(t):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", t))
    for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", t):
        regs.update(range(1000 + int(a), 1000 + int(b) + 1))
    regs.update(1000 + int(a) for a in re.findall(r"\ba(\d+)\b", t))
    return regs


bad_lines = {}
seen = set()
work = []  # (line index, lds queue, global queue, inasm)


def walk(i, queue, gqueue, carried):
    """Linear walk from line i.  queue / gqueue: tuples of (line, frozenset(regs)).  carried: this walk exists only for its queues."""
    fn = func_of[i] if i < len(src) else None
    inasm = False
    while i < len(src):
        if func_of[i] != fn:
            return
        if carried:
            if not queue and not gqueue:
                return
            key = (i, tuple(q[0] for q in queue), tuple(q[0] for q in gqueue))
            if key in seen:
                return
            seen.add(key)
        l = src[i]
        t = l.strip()
        i += 1
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if inasm:
            m = re.match(r"ds_read\w* (v\[(\d+):(\d+)\]|v(\d+)),", t)
            if m:
                regs = frozenset(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else frozenset({int(m.group(4))})
                queue = queue + ((i, regs),)
            m = re.match(r"ds_read\w* a\[(\d+):(\d+)\],", t)  # (accumulator-file destination: registers 1000 + n)
            if m:
                queue = queue + ((i, frozenset(range(1000 + int(m.group(1)), 1000 + int(m.group(2)) + 1))),)
            if re.match(r"ds_write", t):  # an asm store sits in the same in-order queue: counted, nothing lands in registers
                queue = queue + ((i, frozenset()),)
            m = re.match(r"(?:global|buffer)_load\w* (v\[(\d+):(\d+)\]|v(\d+)),", t)
            if m and not t.startswith("global_load_lds") and not t.endswith(" lds"):  # (LDS-DMA: first operand = address, nothing lands in registers)
                regs = frozenset(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else frozenset({int(m.group(4))})
                gqueue = gqueue + ((i, regs),)
            m = re.match(r"(?:global|buffer)_load\w* a\[(\d+):(\d+)\],", t)
            if m:
                gqueue = gqueue + ((i, frozenset(range(1000 + int(m.group(1)), 1000 + int(m.group(2)) + 1))),)
            if "XOP_FENCE" in t:
                gqueue = ()
            mk = re.search(r"XOP_KEEP (\d+)", t)  # a hand-written vmcnt wait that leaves only the youngest n asm loads in flight
            if mk:
                n = int(mk.group(1))
                gqueue = gqueue[len(gqueue) - n:] if n > 0 else ()
            for mm in re.finditer(r"lgkmcnt\((\d+)\)", t):
                n = int(mm.group(1))
                queue = queue[len(queue) - n:] if n > 0 else ()
            continue
        if not t or t[0] in ";." or LABEL.match(l):
            continue
        if "s_waitcnt" in t and "lgkmcnt(0)" in t:
            queue = ()
            continue
        mb = re.match(r"(s_branch|s_cbranch_\w+)\s+([\w.$]+)", t)
        if mb:
            tgt = labels.get(mb.group(2))
            if tgt is not None and (queue or gqueue):
                work.append((tgt, queue, gqueue))
            if mb.group(1) == "s_branch":  # (the fall-through code below an unconditional branch is another path)
                if carried:
                    return
                queue, gqueue = (), ()
            continue
        if t.startswith("s_endpgm") or t.startswith("s_setpc"):
            if carried:
                return
            queue, gqueue = (), ()
            continue
        regs = regs_of(t)
        for ln, rs in queue + gqueue:
            hit = regs & rs
            if hit:
                bad_lines.setdefault(i, (t, sorted(hit), ln))
                break


# base pass: every function from its first line with empty queues (falls through conditional branches; targets are queued)
starts = [i for name, i in labels.items() if not name.startswith(".") and (want in name)]
for s in sorted(starts):
    walk(s, (), (), False)
while work:
    tgt, q, g = work.pop()
    walk(tgt, q, g, True)

for i in sorted(bad_lines)[: (len(bad_lines) if verbose else 20)]:
    t, hit, ln = bad_lines[i]
    print(f"line {i}: `{t}` touches v{hit} of the asm read at line {ln} still in flight")
print("violations:", len(bad_lines))
sys.exit(1 if bad_lines else 0)
