# experiment: request-queue shape vs latency / throughput of flgpu_transform, C caller threads (no GIL)
P=tools/latency/latency_probe
for cfg in ${CFGS:-"64 4 16" "64 4 32" "64 3 32" "256 4 16" "256 4 32" "256 3 32" "16 4 16" "16 4 32" "1024 4 32" "64 4 16"}; do
  set -- $cfg
  echo -n "threads $1 lanes $2 max_batch $3: "; $P $1 ${REQ:-1024} 1920 1080 "w=300&h=200" ${FE:-0} $2 $3
done
