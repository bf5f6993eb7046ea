#!/bin/bash
# What each default of the production step buys, one box, alternating with the default: tools/switch_table.sh > gpurun_out/switch_table.txt
#   columns: pairs/s, ms per step, max step ms, GEMM family ms (serialized pass), serialized ms per step
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  echo -n "$1: "
  env $1 python3 $R/bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['step_ms']['max'], r['roofline']['gemm_ms_per_step'], r['roofline']['serialized_ms_per_step'])"
}
for s in "X=default" "ECAMP_OVERLAP_WGRAD=0" "ECAMP_OVERLAP_BRANCHES=0" "X=default" "ECAMP_WGRAD_GROUP=0" "ECAMP_GELU_SAVED_GRAD=0" "X=default" "ECAMP_HOLD_TENSORS=0" "ECAMP_MAX_STEPS_IN_FLIGHT=0" "ECAMP_FUSED_GRAD_NORM=0" "X=default"; do run "$s"; done
