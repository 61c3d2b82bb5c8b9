#!/bin/bash
# experiment: the EKF stream confined to a subset of the CUs (QS_EKF_CU_MASK: hex words, lowest CUs first), so that the filter's
# kernels do not share SIMDs with the loop-closure chain's workgroups.  usage: tools/ekf_cu_mask_probe.sh [workload] [steps]
WL=${1:-c3}; ST=${2:-20}
F=ffffffff
for mask in "" "0,$F,$F,$F,$F,$F,$F,$F" "0,0,$F,$F,$F,$F,$F,$F" "$F,$F,$F,$F,$F,$F,$F,0" "0,$F,0,$F,0,$F,0,$F" "ffffff00,ffffff00,ffffff00,ffffff00,ffffff00,ffffff00,ffffff00,ffffff00"; do
  echo "mask=$mask"
  QS_EKF_CU_MASK=$mask timeout -k 10 200 python bench.py --no-cpu-baseline --no-micro --workload $WL --steps $ST 2>/dev/null | python -c "
import sys,json; r=json.load(sys.stdin); print(round(r['ms_per_step'],4), {k:round(v,3) for k,v in r['stages_ms_per_step'].items()})"
done
