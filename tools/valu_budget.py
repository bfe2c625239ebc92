#!/usr/bin/env python3
"""Per-work-unit VALU instruction budget of a kernel from its hipcc listing (round 6, VERDICT r5 #2c: "count first").

    python3 tools/valu_budget.py hipt_abmil_atec23_amd/csrc/build/qkv_attention.s qkv_attn_kernelILi0ELb0 [min_block_instructions]

For every basic block of the kernel with at least that many instructions (default 300: the work-unit loop body and its peers): instructions by class
-- matrix (v_mfma), transcendental (v_exp / v_rcp / v_rsq / v_sqrt / v_log), fused multiply-add, add / sub, mul, max / min / med3,
bf16 packing (v_cvt_pk_bf16 / v_perm / and-or packing), register-file moves (v_mov, v_accvgpr), cross-lane (permlane / dpp / readlane), everything
else -- with the issue cycles MI355X_MICROARCH.md prices them at (4 per vector instruction, 8 per transcendental, 8 per MFMA issue slot) beside the
matrix-pipe cycles (32 per 32x32x16 MFMA, 16 per 16x16x32).  The sum over a wave's loop body x waves x work units is what SQ_INSTS_VALU counts."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 300


def cls(op):
    if "mfma" in op:
        return "mfma"
    if not op.startswith("v_"):
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        return "scalar/other"
    if re.match(r"v_(exp|rcp|rsq|sqrt|log|sin|cos)", op):
        return "transcendental"
    if re.match(r"v_(pk_)?fma", op) or op.startswith("v_fmac"):
        return "fma"
    if re.match(r"v_(pk_)?(add|sub)", op):
        return "add/sub"
    if re.match(r"v_(pk_)?mul", op):
        return "mul"
    if re.match(r"v_(max|min|med3|pk_max|pk_min)", op):
        return "max/min"
    if re.match(r"v_(cvt|perm_b32|and_or|lshl_or|bfi|alignbit|bfe|lshrrev|lshlrev|and_b32|or_b32|or3)", op):
        return "convert/pack"
    if re.match(r"v_(mov|accvgpr)", op):
        return "moves"
    if re.match(r"v_(permlane|readlane|readfirstlane|writelane)", op) or "dpp" in op:
        return "cross-lane"
    if re.match(r"v_(cmp|cndmask)", op):
        return "compare/select"
    return "other valu"


ISSUE = {"transcendental": 8, "mfma": 8}
lines = open(path).read().split("\n")
inside, blocks, cur = False, [], None
for l in lines:
    m = re.match(r"^(_Z\S+):", l)
    if m:
        inside = pat in m.group(1)
        if inside:
            print("==", m.group(1))
            cur = [m.group(1)[:40], collections.Counter(), collections.Counter()]
            blocks.append(cur)
        continue
    if not inside:
        continue
    if l.startswith(".Lfunc_end"):
        tot = collections.Counter()
        for b in blocks:
            n = sum(b[1].values())
            for k, v in b[1].items():
                tot[k] += v
            if n >= minn:
                valu = sum(v for k, v in b[1].items() if k not in ("mfma", "lds", "vmem", "scalar/other"))
                issue = sum(v * ISSUE.get(k, 4) for k, v in b[1].items() if k not in ("lds", "vmem", "scalar/other"))
                pipe = sum(v * (16 if "16x16x32" in k else 32) for k, v in b[2].items())
                print(f"  block {b[0]}: {n} instructions, VALU {valu}, MFMA {b[1]['mfma']} ({pipe} matrix-pipe cycles), vector issue cycles {issue}")
                print("     " + ", ".join(f"{k} {v}" for k, v in sorted(b[1].items(), key=lambda kv: -kv[1])))
        inside, blocks, cur = False, [], None
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), collections.Counter(), collections.Counter()]
        blocks.append(cur)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m and cur is not None and not l.strip().startswith((";", ".")):
        cur[1][cls(m.group(1))] += 1
        if "mfma" in m.group(1):
            cur[2][m.group(1)] += 1
