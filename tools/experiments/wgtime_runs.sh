#!/bin/bash
# N fresh processes of the flagship batch with the -DFL_MFMA_TIMING build: where and when every workgroup of launch 100 ran
# (tools/build_ablate.sh "wgtime:0:-DFL_MFMA_TIMING" with ABL_FILE=fl_mfma.hip first).   bash tools/experiments/wgtime_runs.sh [N]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for i in $(seq 1 ${1:-6}); do
  echo "== process $i"
  FLGPU_LIB=$R/tools/libfanlin_gpu_ablate_wgtime.so timeout -k 10 120 python3 tools/experiments/placement_probe4.py 0 0 3 2>&1 | grep -E "mfma wg times|xcc|alloc"
done
