#!/bin/bash
# The round's profiling passes (GPU box): rocprofv3 kernel statistics of the timed workload and of the
# default command, then the PMC passes of the projection kernel at K = 256 / 64 / 32.
#   tools/profile_round.sh r3        -> gpurun_out/prof_r3*, gpurun_out/pmc_r3*.json ; then tools/make_profiles.py r3
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$tag" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras \
  > "$ROOT/gpurun_out/prof_${tag}_bench.json" 2> "$ROOT/gpurun_out/prof_${tag}.err"
echo "stats pass 1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${tag}_default" -- python3 "$ROOT/bench.py" \
  > "$ROOT/gpurun_out/prof_${tag}_default_bench.json" 2> "$ROOT/gpurun_out/prof_${tag}_default.err"
echo "stats pass 2 done"
"$ROOT/tools/pmc_k1.sh" $tag > /dev/null
echo "pmc K=256 done"
"$ROOT/tools/pmc_k1.sh" ${tag}k64 --k-points 64 > /dev/null
echo "pmc K=64 done"
"$ROOT/tools/pmc_k1.sh" ${tag}k32 --k-points 32 > /dev/null
echo "pmc K=32 done"
