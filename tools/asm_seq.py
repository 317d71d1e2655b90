#!/usr/bin/env python3
"""Instruction-kind sequence of one basic block of a `make asm` listing (M mfma, R ds_read, D lds-dma,
W[..] waitcnt, B barrier, S scratch, a accvgpr move, v valu, s salu, g other vmem).
usage: asm_seq.py <file.s> <kernel substring> <block label, e.g. .LBB6_7>"""
import sys

text = open(sys.argv[1]).read()
i = text.index(sys.argv[2])
i = text.index(sys.argv[3] + ":", i)
j = text.index("\n.LBB", i + 1)
out = []
for line in text[i:j].splitlines()[1:]:
    line = line.strip()
    if not line or line.startswith((";", ".")):
        continue
    op = line.split()[0]
    if "mfma" in op: out.append("M")
    elif op.startswith("ds_read"): out.append("R")
    elif op.startswith("global_load_lds"): out.append("D")
    elif op == "s_waitcnt": out.append("W[" + line.split(None, 1)[1] + "]")
    elif op == "s_barrier": out.append("B")
    elif op.startswith("scratch"): out.append("S")
    elif "accvgpr" in op: out.append("a")
    elif op.startswith("v_"): out.append("v")
    elif op.startswith(("global_", "buffer_")): out.append("g")
    else: out.append("s")
print("".join(out))
