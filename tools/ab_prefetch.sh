#!/bin/bash
# A/B: look-ahead wave on/off, chain time per workload (run on the GPU box)
mkdir -p gpurun_out
for wl in c1 adv; do for pf in 1 0; do
  echo "wl=$wl prefetch=$pf"; QS_CHAIN_PREFETCH=$pf timeout -k 10 200 python tools/slam_probe2.py 1048576 3 $wl 2>/dev/null | tail -1
done; done
