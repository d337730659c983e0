#!/bin/bash
# A cap on the source bytes a flush collects, against the pixel-source leg (variants built with
#   ABL_FILE=fl_queue.cpp bash tools/build_ablate.sh "cap0::-DFL_QUEUE_CAP_MB=0" "cap16::-DFL_QUEUE_CAP_MB=16" ...; the probe takes them through LD_PRELOAD)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for v in ${CAPS:-cap0 cap24 cap48}; do
  echo "$v pixels : $(LD_PRELOAD=$R/tools/libfanlin_gpu_ablate_$v.so $R/tools/latency/latency_probe 64 4096 1920 1080 'w=300&h=200' 3 0 0 0 2>&1 | tail -1 | cut -c1-125)"
done
done
