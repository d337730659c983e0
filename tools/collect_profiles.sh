#!/bin/bash
# Runs on the GPU box (through gpurun): collects everything profiles/ is made of into gpurun_out/profiles_raw/.
#   bash tools/collect_profiles.sh
# rocprofv3 needs the program itself after `--` and a writable TMPDIR; counters are collected in separate passes.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_raw
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --config3-share 0 --config4 0"
run() { echo "== $*" >&2; timeout -k 10 300 "$@"; }
# kernel time summaries (the same commands bench.py is judged on, fewer steps)
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/config1 -o c1 -- $B --steps 100 --warmup 3 > $O/config1.log 2>&1   # the default workload: resize + letterbox + JPEG encode
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/config2 -o c2 -- $B --steps 5 --warmup 2 --blur 10 --grayscale --frontend none > $O/config2.log 2>&1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/jpeg -o jp -- $B --steps 5 --warmup 2 --frontend none > $O/jpeg.log 2>&1   # resize only
# the packed arithmetic of rounds 2-3 (narrower than the reference's: an extra of the bench line, not its headline), same workload
export FLGPU_MFMA_ARITH=packed
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/config1_packed -o c1p -- $B --steps 100 --warmup 3 > $O/config1_packed.log 2>&1
unset FLGPU_MFMA_ARITH
# roctx ranges of the runtime (FLGPU_ROCTX=1) next to the kernels
export FLGPU_ROCTX=1
run rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/markers -o mk -- $B --steps 3 --warmup 1 > $O/markers.log 2>&1
unset FLGPU_ROCTX
# HBM traffic and instruction counters of config 1, one pass per counter group
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $B --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- $B --steps 2 --warmup 1 > $O/pmc_sq2.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -o p -- $B --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
# plain bench lines (no profiler attached)
cd $R
for name_args in "config1:" "config1_resize_only:--frontend none" "config1_crop:--crop --frontend none" "config2_gray_blur:--blur 10 --grayscale --frontend none" "config1_jfif444:--frontend jfif444" "config1_webp420:--frontend webp420"; do
  name=${name_args%%:*}; args=${name_args#*:}
  extra="--cpu-images 0 --latency-requests 0 --extra-steps 0 --steps 100 --config3-share 0 --config4 0"; [ "$name" = config1 ] && extra=""
  run python3 bench.py $args $extra > $O/bench_$name.json 2> $O/bench_$name.err
done
ls $O
