#!/bin/bash
# Build side copies of libpsa_hip.so with timing / schedule experiments compiled into a projection
# kernel and time configuration 3 with each:
#   tools/k1_experiments.sh build 1 2 4 ...      (in the build container)
#   tools/k1_experiments.sh run 1 2 4 ...        (on the GPU box; writes gpurun_out/k1_experiments.txt;
#                                                 BENCH_ARGS="--k-points 32" selects another workload)
# KERNEL=k1_pair  MACRO=PSA_K1_EXPERIMENT   the on-the-fly "2 x f16" kernel (results are WRONG by construction)
# KERNEL=k1_planes MACRO=PSA_K1P_X          (default) the planes kernel: schedule variants, results stay right
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/psa_amd/csrc
OUT=$ROOT/tools/probes/_x
KERNEL=${KERNEL:-k1_planes}
MACRO=${MACRO:-PSA_K1P_X}
mode=$1; shift
mkdir -p "$OUT"
if [ "$mode" = build ]; then
  make -C "$SRC" >/dev/null
  for x in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I"$SRC" -fno-fast-math \
      -ffp-contract=on -fno-slp-vectorize -D$MACRO=$x $EXTRA_DEFS -c "$SRC/$KERNEL.hip" -o "$OUT/${KERNEL}_x$x.o" 2>/dev/null
    objs=$(sed -n 's/^SRCS := //p' "$SRC/Makefile" | tr ' ' '\n' | grep -v "^$KERNEL.hip" | sed "s#\(.*\)\.hip#$SRC/build/\1.o#")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$OUT/libpsa_hip_x$x.so" $objs "$OUT/${KERNEL}_x$x.o" \
      -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib 2>/dev/null
    [ -n "$TAG" ] && mv "$OUT/libpsa_hip_x$x.so" "$OUT/libpsa_hip_x$TAG.so" && x=$TAG
    echo "built $OUT/libpsa_hip_x$x.so"
  done
else
  mkdir -p "$ROOT/gpurun_out"
  : > "$ROOT/gpurun_out/k1_experiments.txt"
  for x in "$@"; do
    lib=$SRC/libpsa_hip.so
    [ "$x" != 0 ] && lib=$OUT/libpsa_hip_x$x.so
    PSA_HIP_LIBRARY=$lib python "$ROOT/bench.py" --no-cpu-baseline --steps 5 --warmup 2 $BENCH_ARGS 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('experiment $x: K1', round(d['roofline']['avg_launch_ms'],3), 'ms')" \
      | tee -a "$ROOT/gpurun_out/k1_experiments.txt"
  done
fi
