#!/bin/bash
# experiment helper: times the streaming kernel with parts switched off (FL_ABLATE build)
for lib in ${LIBS:-tools/libfanlin_gpu_ablate.so}; do
export FLGPU_LIB=$PWD/$lib
for ab in ${ABS:-0 1 2 3 4 6 7}; do
  FLGPU_ABLATE=$ab python bench.py --steps 10 --warmup 2 --cpu-images 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib ablate=$ab', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
done; done
