# experiment: flgpu_transform latency vs number of concurrent callers (C threads), pixels and JPEG outputs
P=tools/latency/latency_probe
for fe in 0 3; do for t in 1 2 4 8 16 64; do
  echo -n "front_end $fe threads $t: "; $P $t $((t * 64 < 256 ? 256 : t * 16)) 1920 1080 "w=300&h=200" $fe | cut -c1-120
done; done
