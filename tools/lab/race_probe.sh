#!/bin/bash
# Builds the variants of tools/lab/race_probe.hip (cross-compiles without a GPU) and, with "run", runs them.
# All variants use the round-3 hand-over (-DCIMRGP_GATHER_LATE: the gather right behind a step's multiplies):
#   late     + -DCIMRGP_RACE_UNSETTLED: the code of rounds 1-3 -- reproduces the wrong results (1.7-3 % of the runs)
#   pdump    the same, and the pivot wave records what it READ as gathered: the wrong words are the stale LDS words
#   nop1/4   256 / 1024 idle cycles between the multiplies and the gather's stores: still wrong (not the matrix cores)
#   noprio   the chain kernels at the default wave priority: still wrong (not s_setprio)
#   settled  with the read-back of the last stored word ahead of the barrier (lds_settle): 0 of 1999
# (round 5 also ran, and retired from the source: sleeps behind the barrier and ahead of the operand reads, wait states
#  between the vector instructions that zero accumulator entries and the multiplies -- all still wrong; a build that read
#  every operand again at the end of the step -- never wrong: gpurun_out/r05_race_probe*.txt, profiles/r05_race_probe.txt)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result -DCIMRGP_GATHER_LATE"
build() { [ -x "race_probe_$1" ] && [ "race_probe_$1" -nt race_probe.hip ] && [ "race_probe_$1" -nt ../../cimrgp_amd/csrc/potrf.hip ] || $HIPCC $F $2 race_probe.hip -o "race_probe_$1"; }
if [ "${1:-build}" = build ]; then
    build late "-DCIMRGP_RACE_UNSETTLED" &
    build pdump "-DCIMRGP_RACE_UNSETTLED -DCIMRGP_RACE_PDUMP" &
    build nop1 "-DCIMRGP_RACE_UNSETTLED -DRACE_NOPS_AFTER=1" &
    wait
    build nop4 "-DCIMRGP_RACE_UNSETTLED -DRACE_NOPS_AFTER=4" &
    build noprio "-DCIMRGP_RACE_UNSETTLED -DRACE_NOPRIO" &
    build settled "" &
    wait
    ls -la race_probe_*
else
    for v in ${RACE_VARIANTS:-late pdump nop1 nop4 noprio settled}; do echo "== $v"; ./race_probe_$v ${RACE_REPS:-1000}; done
fi
