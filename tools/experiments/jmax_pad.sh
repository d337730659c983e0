# experiment: exact vs padded per-lane column counts of the horizontal pass tables
for r in 1 2 3; do for pad in 4 1; do
  FLGPU_JMAX_PAD=$pad python bench.py --steps 10 --warmup 2 --cpu-images 0 --latency-requests 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad kernel_ms', round(d['roofline']['kernel_ms'],3))"
done; done
