#!/bin/bash
# sweep of k_trace_wq knobs on the GPU box: each argument is one environment, e.g. "RTX_WQ_WALKERS=10 RTX_WQ_PATHS=960"
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "== $v"
  env RTX_TRACE_KERNEL=wq_diag $v timeout -k 10 120 python scripts/kernel_diag.py 100 2>&1 | grep -v amdgpu.ids
done
