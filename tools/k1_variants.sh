#!/bin/bash
# Time side builds of the planes kernel (tools/k1_experiments.sh build ...) on several shapes, two
# rounds each, interleaved (GPU box):  tools/k1_variants.sh "<variants>" "<k-point counts>"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/probes/_x
res=$ROOT/gpurun_out/${TAG:-k1_variants}.txt
: > "$res"
for K in $2; do
  for round in 1 2; do
    for x in $1; do
      lib=$ROOT/psa_amd/csrc/libpsa_hip.so
      [ "$x" != 0 ] && lib=$OUT/libpsa_hip_x$x.so
      PSA_HIP_LIBRARY=$lib python "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 6 --warmup 2 --k-points $K $BENCH_EXTRA 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('K=$K round $round variant $x: K1 %.3f ms  frac %.3f (%s)' % (r['avg_launch_ms'], r['frac'], r['bound']))" | tee -a "$res"
    done
  done
done
