#!/bin/bash
# PMC passes for the two JPEG encode kernels (runs on the GPU box through gpurun): bash tools/collect_jpeg_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/jpeg_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --steps 3 --warmup 1"
run() { echo "== $*" >&2; timeout -k 10 300 "$@"; }
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $B > $O/pmc_sq.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- $B > $O/pmc_sq2.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN --output-format csv -d $O/pmc_sq3 -o p -- $B > $O/pmc_sq3.log 2>&1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $B > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -30
