#!/bin/bash
# Builds the variants of tools/lab/race_probe.hip (cross-compiles without a GPU) and, with "run", runs them.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -DCIMRGP_HANDOVER_LATE"
build() { [ -x "race_probe_$1" ] && [ "race_probe_$1" -nt race_probe.hip ] && [ "race_probe_$1" -nt ../../cimrgp_amd/csrc/potrf.hip ] || $HIPCC $F $2 race_probe.hip -o "race_probe_$1"; }
if [ "${1:-build}" = build ]; then
    build late "" &
    build dump "-DCIMRGP_RACE_DUMP" &
    build nop1 "-DRACE_NOPS_AFTER=1" &      # 4 tiles x 16 states = 64 states = 256 cycles
    wait
    build nop4 "-DRACE_NOPS_AFTER=4" &      # 1024 cycles
    build pre4 "-DRACE_PRESLEEP=4" &        # ~256 cycles ahead of the operand reads, behind the zeroing
    build pre16 "-DRACE_PRESLEEP=16" &      # ~1024 cycles
    wait
    build noprio "-DRACE_NOPRIO" &           # the chain kernels without s_setprio(3)
    build rdback "-DRACE_RDBACK" &           # the gathering wave reads its last LDS write back before the barrier
    build pdump "-DCIMRGP_RACE_PDUMP" &      # the pivot wave's gathered columns as read, one store per lane and block
    build detect "-DCIMRGP_RACE_DETECT" &   # every LDS value taken right behind a barrier is read again at the end of the step
    build top4 "-DRACE_TOPSLEEP=4" &        # ~256 cycles right behind the barrier, ahead of the zeroing
    build vnop1 "-DRACE_VNOP=1" &           # 4 x 2 wait states between the zeroing and the multiplies
    build vnop7 "-DRACE_VNOP=7" &           # 4 x 8 wait states
    wait
    ls -la race_probe_*
else
    for v in ${RACE_VARIANTS:-late detect dump nop1 nop4 pre4 pre16 top4 vnop1 vnop7}; do echo "== $v"; ./race_probe_$v ${RACE_REPS:-300}; done
fi
