# experiment: request-queue shape (lanes x max batch) vs latency / throughput of flgpu_transform with host buffers
for cfg in "1 0" "2 32" "4 16" "4 8" "8 8" "1 0" "2 32" "4 16" "4 8" "8 8"; do
  set -- $cfg
  python bench.py --steps 2 --warmup 1 --cpu-images 0 --latency-requests 512 --latency-threads ${THREADS:-64} --queue-lanes $1 --queue-max-batch $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['latency']; print('lanes $1 max_batch $2', 'p50', round(d['p50_ms'],2), 'p99', round(d['p99_ms'],2), 'img/s', round(d['images_per_s']), 'flushes', d['queue_flushes'])"
done
