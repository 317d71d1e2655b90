#!/bin/bash
# FETCH_SIZE of the projection kernel for a side build (tools/k1_experiments.sh build <x>), GPU box:
#   tools/pmc_fetch_variant.sh <x> [bench.py args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
x=$1; shift
out=$ROOT/gpurun_out/pmc_fetch_x$x
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export PSA_HIP_LIBRARY=$ROOT/tools/probes/_x/libpsa_hip_x$x.so
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$out.log" 2>&1
python3 "$ROOT/tools/pmc_summarise.py" "$out" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print('variant $x', k[:40], 'FETCH_SIZE x2 = %.2f GB' % (2*v['FETCH_SIZE']*1024/1e9), 'kernel_ms', round(v.get('kernel_ms',0),3))"
