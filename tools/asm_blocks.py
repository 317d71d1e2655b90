#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a `make asm` listing.
usage: asm_blocks.py <file.s> <substring of the mangled kernel name> [min instructions]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
key = sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = text.index(key + "")
start = text.index("\n", text.index(":", start))
end = text.index("s_endpgm", start)
blocks, cur = [], ["<entry>"]
for line in text[start:end].splitlines():
    if re.match(r"^\.LBB", line):
        blocks.append(cur)
        cur = [line.split(":")[0]]
    elif line.startswith("\t") and line.strip() and not line.strip().startswith((".", ";")):
        cur.append(line.strip().split()[0])
blocks.append(cur)


def kind(op):
    if "mfma" in op: return "mfma"
    if op.startswith("ds_read"): return "ds_read"
    if op.startswith("ds_write"): return "ds_write"
    if op.startswith("global_load_lds"): return "lds_dma"
    if op.startswith("scratch_"): return op
    if op.startswith(("global_", "buffer_")): return "vmem"
    if op == "s_waitcnt": return "waitcnt"
    if op == "s_barrier": return "barrier"
    if op.startswith("v_accvgpr"): return op
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return op


for b in blocks:
    ops = b[1:]
    if len(ops) < floor:
        continue
    print(f"{b[0]:12s} {len(ops):5d}", dict(Counter(kind(o) for o in ops)))
    print("             valu:", Counter(o for o in ops if kind(o) == "valu").most_common(10))
