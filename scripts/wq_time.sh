#!/bin/bash
# timing (no diag) of k_trace_wq knob settings: each argument is one environment
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "== $v"
  env RTX_TRACE_KERNEL=wq $v timeout -k 10 120 python scripts/kernel_diag.py 100 2>&1 | grep -v amdgpu.ids
done
