#!/usr/bin/env python3
"""Build-time check of csrc/embed32.hip's hand-counted ring wait (round 6).

A ring phase of embed32_kernel ends in `s_waitcnt vmcnt(NLD)` + `s_barrier`: "my LDS-DMA pieces of the next unit have landed; the NLD pixel loads
issued behind them may still be in flight".  vmcnt counts INSTRUCTIONS in issue order, so NLD must equal the number of global loads hipcc emitted
between the last DMA piece and that wait -- not the number written in the source (hipcc once merged 12 8-byte loads into 8 wider ones: a race on
the weight ring).  For every embed32_kernel instantiation in the listing: every `s_waitcnt vmcnt(N)` that is followed by `s_barrier` with a
`buffer_load ... lds` before it must see exactly N `global_load` instructions (and no stores) since that DMA piece.

    python3 tools/audit_embed32_loads.py hipt_abmil_atec23_amd/csrc/build/embed32.s      (exit 1 on a violation)
"""
import re
import sys


def main(path):
    s = open(path).read()
    bad = 0
    checked = 0
    for m in re.finditer(r'^(_ZN12_GLOBAL__N_114embed32_kernelILi(\d)ELb(\d)EEEv11EmbedParams): ;.*?\n(.*?)s_endpgm', s, re.S | re.M):
        lines = [l.strip() for l in m.group(4).split("\n")]
        since_dma = None  # global loads since the last DMA piece
        for i, l in enumerate(lines):
            if re.match(r'buffer_load_dword\w* .*\blds\b', l):
                since_dma = 0
            elif l.startswith("global_load") or l.startswith("buffer_load"):
                if since_dma is not None:
                    since_dma += 1
            elif l.startswith(("global_store", "buffer_store", "global_atomic", "scratch_")):
                since_dma = None  # a store / spill in between: not a ring wait this check understands
            w = re.match(r's_waitcnt vmcnt\((\d+)\)', l)
            if w and since_dma is not None:
                nxt = next((x for x in lines[i + 1:i + 4] if x and not x.startswith((";", "s_nop"))), "")
                if nxt.startswith("s_barrier"):
                    n = int(w.group(1))
                    checked += 1
                    if n != 0 and n != since_dma:
                        bad += 1
                        print(f"KIND {m.group(2)} LNOUT {m.group(3)}: vmcnt({n}) before a ring barrier, but {since_dma} loads were issued since the last DMA piece")
                    since_dma = None
    print(f"ring waits checked: {checked}")
    print(f"violations: {bad}")
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
