#!/bin/bash
# interleaved A/B of kernel variants (tools/build_ablate.sh) on one box: resize-only config 1, kernel time from HIP events
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for lib in tools/libfanlin_gpu_ablate_*.so; do
  FLGPU_LIB=$PWD/$lib python bench.py --steps 30 --warmup 3 --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --frontend none ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3))"
done; done
