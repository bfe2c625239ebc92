#!/usr/bin/env python3
"""Build-time check of csrc/qkv_attention.hip's counted wait at the first ring barrier of a work unit (round 6).

`s_waitcnt vmcnt(8)` there means: "everything but the previous work unit's eight output stores has landed" (the next ring unit's DMA pieces and the
operand loads are older: vmcnt is in order).  That holds only while (a) a work unit issues EXACTLY eight output stores (hipcc neither merged nor split
the eight raw_buffer_store_b64), and (b) no vector-memory instruction sits between the operand fence (`; XOP_FENCE` in the listing) and that wait.

    python3 tools/audit_qkv_wait.py hipt_abmil_atec23_amd/csrc/build/qkv_attention.s      (exit 1 on a violation)
"""
import re
import sys


def main(path):
    s = open(path).read()
    m = re.search(r'^(_ZN12_GLOBAL__N_115qkv_attn_kernelILi0ELb0EEEvNS_13QkvAttnParamsE): ;.*?\n(.*?)s_endpgm', s, re.S | re.M)
    if not m:
        print("qkv_attn_kernel<0, false> not found")
        return 1
    lines = [l.strip() for l in m.group(2).split("\n")]
    bad = 0
    vm = lambda l: l.startswith(("global_", "buffer_", "flat_", "scratch_"))
    stores = [l for l in lines if l.startswith("buffer_store")]
    if len(stores) != 8 or any(not l.startswith("buffer_store_dwordx2") for l in stores):
        bad += 1
        print(f"expected exactly eight buffer_store_dwordx2 (the output stores of a work unit), found {len(stores)}: {sorted(set(x.split()[0] for x in stores))}")
    waits = [i for i, l in enumerate(lines) if re.match(r's_waitcnt vmcnt\(8\)', l) and "XOP_FENCE" not in l]
    if len(waits) != 1:
        bad += 1
        print(f"expected ONE `s_waitcnt vmcnt(8)` besides the operand fence, found {len(waits)}")
    for i in waits:
        j = i
        while j >= 0 and "XOP_FENCE" not in lines[j]:
            j -= 1
        between = [l for l in lines[j + 1:i] if vm(l)] if j >= 0 else ["(no XOP_FENCE before the wait)"]
        if between:
            bad += 1
            print(f"vector-memory instructions between the operand fence and vmcnt(8): {between[:4]}")
    print(f"violations: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
