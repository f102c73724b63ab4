#!/bin/bash
# Build experiment variants of the library side by side: scripts/ab_builds.sh name1 "-DX=1" name2 "-DY" ...
# Each lands in ray-tracing-series-rust_amd/lib/<name>/librtx_hip.so (travels with gpurun; RTX_LIBRARY=<path> selects it: api.py).
# The product build in lib/librtx_hip.so is rebuilt last, without extra flags.
cd "$(dirname "$0")/.."
L=ray-tracing-series-rust_amd/lib
while [ $# -ge 2 ]; do
  name="$1"; flags="$2"; shift 2
  RTX_EXTRA_HIPFLAGS="$flags" python - <<PY || exit 1
import sys; sys.path.insert(0, "ray-tracing-series-rust_amd")
import build
build.build_library(force=True, verbose=False)
PY
  mkdir -p $L/$name && cp $L/librtx_hip.so $L/$name/librtx_hip.so && echo "built $name [$flags]"
done
python - <<PY
import sys; sys.path.insert(0, "ray-tracing-series-rust_amd")
import build
build.build_library(force=True, verbose=False); build.build_app(); build.build_roofline_tools()
PY
echo "product rebuilt"
