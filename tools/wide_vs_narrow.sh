#!/bin/bash
# The 256-row form of the planes kernel (k1_planes_wide.hip, default where it costs no extra rows) against the
# 128-row loader-wavefront form on every configuration, interleaved, two rounds (GPU box).
cd "$(dirname "$0")/.."
out=gpurun_out/${TAG:-wide_vs_narrow}.txt; : > $out
for r in 1 2; do
  for cfg in "--config C3" "--config C3 --k-points 128" "--config C3 --k-points 192" "--config C3 --k-points 96" "--config C3 --summation incoherent" "--config C2" "--config C4" "--config C5"; do
    for k1 in narrow wide; do
      timeout -k 10 200 python bench.py --k1 $k1 --no-cpu-baseline --no-extras --steps 6 --warmup 2 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('round $r  %-36s %-6s K1 %7.3f ms x %d  frac %.3f  step %7.3f ms  %s' % ('$cfg', '$k1', r['avg_launch_ms'], r['launches'] // 6, r['frac'], d['ms_per_step'], r['kernel'][:22]))" | tee -a $out
    done
  done
done
