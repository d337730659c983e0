#!/bin/bash
# alternates library builds under the DEFAULT bench workload (resize + letterbox + JPEG encode, 400 timed steps), the shape the
# driver times: LIBS="a.so b.so" [ROUNDS=3] bash tools/experiments/ab_bench_default.sh
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${ROUNDS:-3}); do for lib in $LIBS; do
  FLGPU_LIB=$PWD/$lib python bench.py --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'ms/step', round(d['ms_per_step'],4), 'stages', {k: round(v,3) for k,v in d.get('stage_ms_per_step',{}).items()}, flush=True)"
done; done
