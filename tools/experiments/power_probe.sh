#!/bin/bash
# samples rocm-smi (socket power, sclk) while bench.py runs the resample kernel back to back; prints the mean power and the kernel time
# usage: [FLGPU_LIB=...] bash tools/experiments/power_probe.sh [steps]
cd "$(dirname "$0")/../.."
STEPS=${1:-3000}
python bench.py --steps $STEPS --warmup 5 --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --frontend none > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 3.5
P=""
for i in $(seq 1 4); do
  kill -0 $BP 2>/dev/null || break
  P="$P $(/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr -d '()' | tr '\n' ' ')"
  sleep 0.3
done
wait $BP
python -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('${FLGPU_LIB##*/}', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'samples [sclk, W]:', '$P')"
