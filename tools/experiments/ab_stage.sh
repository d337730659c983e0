# usage: LIBS="a.so b.so" STAGE=blur BENCH_ARGS="--blur 10 --grayscale" bash tools/experiments/ab_stage.sh -- interleaved A/B of one stage time
for r in 1 2 3; do for lib in $LIBS; do
  FLGPU_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --cpu-images 0 --latency-requests 0 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '${STAGE:-blur}', round(d['stage_ms_per_step']['${STAGE:-blur}'],3), 'value', round(d['value']))"
done; done
