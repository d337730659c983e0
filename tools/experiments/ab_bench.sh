# usage: LIBS="a.so b.so" bash tools/experiments/ab_bench.sh  -- interleaved A/B of whole libraries on one box
for r in 1 2 3; do for lib in $LIBS; do
  FLGPU_LIB=$PWD/$lib python bench.py --steps 30 --warmup 3 --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --frontend none ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done; done
