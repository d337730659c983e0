#!/bin/bash
# interleaved A/B of library builds on one box: LIBS="a.so b.so" [ROUNDS=3] [BENCH_ARGS=...] bash tools/experiments/ab_libs.sh
# prints the resample kernel time (HIP events inside bench.py) and ms/step of config 1, resize only
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${ROUNDS:-3}); do for lib in $LIBS; do
  FLGPU_LIB=$PWD/$lib python bench.py --steps ${STEPS:-100} --warmup 5 --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --frontend none ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), flush=True)"
done; done
