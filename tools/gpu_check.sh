#!/bin/bash
# on the GPU box: the GPU suite, then (only if it is green) the SLAM chain probe N times.  usage: tools/gpu_check.sh [N]
N=${1:-2}
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1
tail -1 gpurun_out/t.log
grep -q " passed" gpurun_out/t.log || exit 1
grep -q "failed\|error" gpurun_out/t.log && exit 1
for i in $(seq $N); do timeout -k 10 120 python tools/slam_probe.py 1048576 || exit 1; done
