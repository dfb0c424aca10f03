#!/bin/bash
# Local front end of a gpurun call: the library that travels to the GPU box must be the build of the CURRENT sources
# (a failed local `make` once sent a stale libmaskdit_hip.so and the measurements of that call described the old kernels).
#   bash tools/gpu.sh <timeout s> '<command run on the GPU box>'
set -e
cd "$(dirname "$0")/.."
if ! make -C maskdit_amd/csrc -j8 > /tmp/mdt_make.log 2>&1; then
  grep -i "error" /tmp/mdt_make.log | head -20
  echo "tools/gpu.sh: build failed -- not calling gpurun" >&2
  exit 1
fi
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
