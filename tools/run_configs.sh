#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${TAG:-r2}_configs.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 280 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step', round(d['ms_per_step'],3), 'value', '%.3e'%d['value'], 'k1_ms', round(r['avg_launch_ms'],3), 'launches', r['launches'], 'bound', r['bound'], 'achieved', round(r['achieved'],1), r['unit'], 'frac', round(r['frac'],3), 'hbm_frac', round(r['hbm_frac_if_bytes_bound'],3), 'kernel', r['kernel'][:16])" >> $out; }
run --config C3
run --config C3 --k1 onthefly
run --config C3 --k1 bf16x3
run --config C3 --k1 mfma32
run --config C3 --k-points 128
run --config C3 --k-points 64
run --config C3 --k-points 32
run --config C3 --k-points 16
run --config C3 --summation incoherent
run --config C2
run --config C4
run --config C5
cat $out
