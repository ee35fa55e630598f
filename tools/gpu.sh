#!/bin/bash
# build the library here (incremental), then run a command on the GPU box: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import vkn_import; v = vkn_import.load(); v.build()"
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
