#!/bin/bash
# round-5 GPU call 36: timeline of two pipelined steps (the early panels of step i+1 beside the last panels of step i)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/steptr -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-launch-events > gpurun_out/steptr.out 2> gpurun_out/steptr.err || { tail -5 gpurun_out/steptr.err; exit 1; }
tail -1 gpurun_out/steptr.out | cut -c1-200
python3 tools/timeline.py gpurun_out/steptr boundary 6 7500 7500 > gpurun_out/r05_timeline_pipelined_steps.txt
rm -rf gpurun_out/steptr
wc -l gpurun_out/r05_timeline_pipelined_steps.txt
