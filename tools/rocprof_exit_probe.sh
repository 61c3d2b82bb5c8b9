#!/bin/bash
# does the bench leave rocprofv3 alive at exit? (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exitp
A="--workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-micro"
run() { tag=$1; shift; "$@" > gpurun_out/exitp/$tag.out 2> gpurun_out/exitp/$tag.err; echo "$tag rc=$?"; rm -rf gpurun_out/exitp/tr_$tag; }
run default rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exitp/tr_default -- python3 bench.py $A
QS_EKF_CU_MASK=none run nomask rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exitp/tr_nomask -- python3 bench.py $A
run noekf rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exitp/tr_noekf -- python3 bench.py $A --ekf 0
