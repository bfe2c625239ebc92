#!/usr/bin/env python3
"""Build-time check of the hand-counted ring waits of the streaming kernels against the instructions hipcc actually emitted (round 6).

A ring phase of these kernels ends in `s_waitcnt vmcnt(N)` + `s_barrier`: "my LDS-DMA pieces of the next ring unit have landed; the N youngest
vector-memory operations may still be in flight".  vmcnt counts INSTRUCTIONS in issue order, so N must be the number hipcc emitted -- not the
number written in the source (it once merged 12 8-byte pixel loads of embed32.hip into 8 wider ones: vmcnt(12) let four DMA pieces pass the
barrier in flight, a race that showed only under stream concurrency).  Per kernel family, for every `s_waitcnt vmcnt(N)` followed by `s_barrier`:

  embed32_kernel        N <= the vector-memory instructions (pixel loads; spill traffic counts too) issued since the last DMA piece
  seqgemm_pipe_kernel   N <= the epilogue stores issued since the last DMA piece
  mlp16_kernel          N = 12 <= the vector-memory instructions issued since the previous ring barrier (this phase's twelve DMA pieces for the
                        unit after next; the first wait behind a row phase also sees the row loads / epilogue stores)
Fewer than N is a VIOLATION (pieces may be in flight behind the barrier); more than N is reported as a note (the wait is stricter than intended).

    python3 tools/audit_ring_waits.py hipt_abmil_atec23_amd/csrc/build/<embed32|seqgemm_pipe|mlp16>.s      (exit 1 on a violation)
"""
import re
import sys

VM = ("global_", "buffer_", "flat_", "scratch_")


def is_dma(l):
    return bool(re.match(r'buffer_load_dword\w* .*\blds\b', l)) or l.startswith("global_load_lds")


def ring_waits(lines):
    """(index, N) of every s_waitcnt vmcnt(N) that is followed by s_barrier"""
    for i, l in enumerate(lines):
        w = re.match(r's_waitcnt vmcnt\((\d+)\)', l)
        if w:
            nxt = next((x for x in lines[i + 1:i + 4] if x and not x.startswith((";", "s_nop"))), "")
            if nxt.startswith("s_barrier"):
                yield i, int(w.group(1))


def main(path):
    s = open(path).read()
    bad = checked = notes = 0
    for m in re.finditer(r'^(_ZN12_GLOBAL__N_1\d+(embed32_kernel|seqgemm_pipe_kernel|mlp16_kernel)I\S*): ;.*?\n(.*?)s_endpgm', s, re.S | re.M):
        name, fam = m.group(1), m.group(2)
        lines = [l.strip() for l in m.group(3).split("\n")]
        waits = dict(ring_waits(lines))
        since = None      # non-DMA vector-memory instructions since the last DMA piece (embed32, seqgemm_pipe)
        kinds = set()
        ndma = nother = 0  # since the previous ring barrier (mlp16)
        for i, l in enumerate(lines):
            if is_dma(l):
                since, kinds = 0, set()
                ndma += 1
            elif l.startswith(VM):
                nother += 1
                if since is not None:
                    since += 1
                    kinds.add("store" if "_store" in l or "_atomic" in l else "load")
            if i in waits:
                n = waits[i]
                checked += 1
                # SAFE iff at least N vector-memory instructions were issued behind the DMA pieces that must have landed: the N youngest are then
                # all younger than those pieces.  Fewer than N: pieces may still be in flight behind the barrier -- a violation.  More than N (a
                # register spill or reload in the window counts too): the wait also covers some of the loads / stores meant to fly -- slower, not wrong.
                if fam == "mlp16_kernel":
                    have, what = ndma + nother, f"{ndma} DMA pieces + {nother} other vector-memory instructions since the previous ring barrier"
                else:
                    have, what = since, f"{since} vector-memory instructions ({sorted(kinds)}) since the last DMA piece"
                if n != 0 and (have is None or have < n):
                    bad += 1
                    print(f"VIOLATION {name[:64]}: vmcnt({n}) before a ring barrier, but only {what}")
                elif n != 0 and have > n:
                    notes += 1
                    if notes <= 6:
                        print(f"note {name[:64]}: vmcnt({n}) with {what}: stricter than intended")
                since, ndma, nother = None, 0, 0
    print(f"ring waits checked: {checked} ({notes} stricter than intended)")
    print(f"violations: {bad}")
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
