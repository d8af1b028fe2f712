#!/bin/bash
# Container-side wrapper: rebuild the in-tree libraries (the GPU box runs whatever .so travels with the snapshot), then hand
# the command to gpurun.  usage: scripts/gpu.sh <timeout_s> '<command>'
cd "$(dirname "$0")/.." || exit 1
python -c "import __graft_entry__ as g; g.build()" || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
