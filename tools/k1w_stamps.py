#!/usr/bin/env python3
"""Mean per-stage cycle counts of the diagnostic build of k1_planes_wide.hip (-DPSA_K1W_STAMP=1), by wavefront:
    PSA_HIP_LIBRARY=tools/probes/_x/libpsa_hip_xstamp.so python bench.py --k1 wide --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>&1 | python tools/k1w_stamps.py"""
import collections
import re
import sys

rows = collections.defaultdict(list)
for ln in sys.stdin:
    m = re.search(r"block (\d+) wave (\d+) stages (\d+): dma (\d+)  tile0 (\d+)  tiles1-3 (\d+)  tiles4-7 (\d+)  fold\+vmcnt (\d+)  barrier (\d+)", ln)
    if m:
        b, w, n, *v = map(int, m.groups())
        rows[w].append(v)
print("wave   dma  tile0  t1-3  t4-7  fold+vmcnt barrier   total   (mean over the printed workgroups and launches, cycles per stage;")
print("                                                          every interval includes one s_memtime round trip, ~100 cycles)")
for w in sorted(rows):
    v = [sum(x) / len(x) for x in zip(*rows[w])]
    print("%4d %6.0f %6.0f %5.0f %5.0f %8.0f %8.0f %8.0f" % (w, *v, sum(v)))
