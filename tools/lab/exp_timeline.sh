#!/bin/bash
# kernel-trace timeline of one factorisation: bash tools/exp_timeline.sh TAG [n] [rows]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-x}; N=${2:-8192}; ROWS=${3:-0}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_tr -- python3 tools/potrf_once.py $N 3 $ROWS > gpurun_out/${TAG}_tr.log 2>&1 || { tail -5 gpurun_out/${TAG}_tr.log; exit 1; }
python3 tools/timeline.py gpurun_out/${TAG}_tr > gpurun_out/${TAG}_timeline_n${N}_rows${ROWS}.txt
rm -rf gpurun_out/${TAG}_tr
tail -3 gpurun_out/${TAG}_tr.log
