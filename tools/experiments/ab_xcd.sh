#!/bin/bash
# same-box A/B: strips of a picture on one XCD (default) vs the plain launch order (FLGPU_NO_XCD_ORDER=1)
cd "$(dirname "$0")/../.."
for r in 1 2 3 4; do for v in 0 1; do
  FLGPU_NO_XCD_ORDER=$v python bench.py --steps 40 --warmup 3 --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --frontend none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no_xcd_order=$v', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done; done
