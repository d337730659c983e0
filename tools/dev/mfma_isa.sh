#!/bin/bash
# development aid: compiles fl_mfma.hip to gfx950 ISA, prints the register / scratch use of every instantiation and writes
# the flagship instantiation (Rgb8, letterbox, operands in LDS) to /tmp/k2.s     usage: tools/dev/mfma_isa.sh [extra flags]
cd "$(dirname "$0")/../../fanlin-rs_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -S --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" fl_mfma.hip -o /tmp/fl_mfma2.s 2> /tmp/fl_mfma2.log
grep -E "error" -A6 /tmp/fl_mfma2.log | head -40
grep -E "Function Name|VGPRs:|ScratchSize|SGPRs Spill" /tmp/fl_mfma2.log | paste - - - - | sed 's/fl_mfma.hip:[0-9]*:1: remark://g' | awk '{print $3, $5, $6, $9, $10, $11, $12, $13,$14}' | sed 's/_ZN2fl12_GLOBAL__N_120resample_mfma_kernel//; s/EEEvPKNS_3JobEPKNS_8MfmaItemEPKjjjPj//'
awk '/^_ZN2fl12_GLOBAL__N_120resample_mfma_kernelILi3ELb1ELb1ELb0EEE[A-Za-z0-9_]*:/,/s_endpgm/' /tmp/fl_mfma2.s > /tmp/k2.s
wc -l /tmp/k2.s
