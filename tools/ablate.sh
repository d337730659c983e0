#!/bin/bash
# experiment helper: times the streaming kernel of every tools/libfanlin_gpu_ablate_*.so, ROUNDS times interleaved
for r in $(seq 1 ${ROUNDS:-1}); do
for lib in tools/libfanlin_gpu_ablate_*.so; do
  FLGPU_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --cpu-images 0 --latency-requests 0 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
done; done
