"""Audit of a -save-temps .s: between an inline-asm ds_read and the inline-asm s_waitcnt lgkmcnt(N) that retires it,
no compiler-generated instruction may read or write its destination registers (the data lands asynchronously;
hipcc treats the destination as written at ;;#ASMEND).  The same for inline-asm global_load_* destinations, which stay
in flight until the inline-asm wait marked `; XOP_FENCE` (qkv_attention.hip).
usage: audit_asm_reads.py file.s [kernel-symbol-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
inside = want == ""
queue = []  # [(line, set(regs))]
gqueue = []  # asm global loads in flight
inasm = False
bad = 0
for i, l in enumerate(src):
    t = l.strip()
    lab = re.match(r"^([A-Za-z_][\w.$]*):", l)
    if want and lab:
        inside = want in lab.group(1)
        queue = []
        gqueue = []
    if not inside:
        continue
    if t.startswith(";;#ASMSTART"):
        inasm = True
        continue
    if t.startswith(";;#ASMEND"):
        inasm = False
        continue
    if inasm:
        m = re.match(r"ds_read\w* (v\[(\d+):(\d+)\]|v(\d+)),", t)
        if m:
            regs = set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(4))}
            queue.append((i + 1, regs))
        m = re.match(r"ds_read\w* a\[(\d+):(\d+)\],", t)  # (accumulator-file destination: registers 1000 + n)
        if m:
            queue.append((i + 1, set(range(1000 + int(m.group(1)), 1000 + int(m.group(2)) + 1))))
        if re.match(r"ds_write", t):  # an asm store sits in the same in-order queue: counted, nothing lands in registers
            queue.append((i + 1, set()))
        m = re.match(r"(?:global|buffer)_load\w* (v\[(\d+):(\d+)\]|v(\d+)),", t)
        if m and not t.startswith("global_load_lds") and not t.endswith(" lds"):  # (LDS-DMA: the first operand is the address, nothing lands in registers)
            regs = set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(4))}
            gqueue.append((i + 1, regs))
        m = re.match(r"(?:global|buffer)_load\w* a\[(\d+):(\d+)\],", t)  # (accumulator-file destination: registers 1000 + n below)
        if m:
            gqueue.append((i + 1, set(range(1000 + int(m.group(1)), 1000 + int(m.group(2)) + 1))))
        if "XOP_FENCE" in t:
            gqueue = []
        mk = re.search(r"XOP_KEEP (\d+)", t)  # a hand-written vmcnt wait that leaves only the youngest n asm loads in flight
        if mk:
            n = int(mk.group(1))
            gqueue = gqueue[len(gqueue) - n:] if n > 0 else []
        for mm in re.finditer(r"lgkmcnt\((\d+)\)", t):
            n = int(mm.group(1))
            queue = queue[len(queue) - n:] if n > 0 else []
        continue
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    if "s_waitcnt" in t and "lgkmcnt(0)" in t:
        queue = []
        continue
    if t.startswith("s_branch") or t.startswith("s_endpgm"):  # (control flow is not followed: the fall-through code
        queue = []                                             #  below an unconditional branch is another path)
        gqueue = []
        continue
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", t))
    for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", t):
        regs.update(range(1000 + int(a), 1000 + int(b) + 1))
    regs.update(1000 + int(a) for a in re.findall(r"\ba(\d+)\b", t))
    for ln, rs in queue + gqueue:
        hit = regs & rs
        if hit:
            bad += 1
            if bad <= 20 or "-v" in sys.argv:
                print(f"line {i + 1}: `{t}` touches v{sorted(hit)} of the asm read at line {ln} still in flight")
            break
print("violations:", bad)
sys.exit(1 if bad else 0)
