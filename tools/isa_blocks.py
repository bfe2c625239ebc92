#!/usr/bin/env python3
"""Per-basic-block instruction mix of a kernel in a hipcc -S listing (which blocks hold the MFMAs, how many register-file
copies / nops / waits sit beside them).  usage: isa_blocks.py file.s kernel_substring [min_instructions]"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split("\n")
inside, blocks, cur = False, [], None
for l in lines:
    m = re.match(r"^(_Z\S+):", l)
    if m:
        inside = pat in m.group(1)
        if inside:
            print("==", m.group(1))
            cur = [m.group(1)[:30], {}]
            blocks.append(cur)
        continue
    if not inside:
        continue
    if l.startswith(".Lfunc_end"):
        for b in blocks:
            d = b[1]
            n = sum(d.values())
            if n >= minn:
                c = lambda f: sum(v for k, v in d.items() if f(k))
                print(f"  {b[0]:<14} n={n:5d} mfma={c(lambda k: 'mfma' in k):4d} acc_rd={d.get('v_accvgpr_read_b32', 0):4d} acc_wr={d.get('v_accvgpr_write_b32', 0):4d} "
                      f"mov={c(lambda k: k.startswith('v_mov')):4d} ds_rd={c(lambda k: k.startswith('ds_read')):4d} ds_wr={c(lambda k: k.startswith('ds_write')):3d} "
                      f"vmem={c(lambda k: k.startswith(('buffer_', 'global_', 'flat_'))):4d} nop={d.get('s_nop', 0):3d} wait={d.get('s_waitcnt', 0):3d} "
                      f"scratch={c(lambda k: k.startswith('scratch_')):3d} valu={c(lambda k: k.startswith('v_') and 'mfma' not in k):5d}")
        inside, blocks, cur = False, [], None
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), {}]
        blocks.append(cur)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m and cur is not None and not l.strip().startswith((";", ".")):
        cur[1][m.group(1)] = cur[1].get(m.group(1), 0) + 1
