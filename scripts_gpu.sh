#!/bin/bash
# helper: rebuild the .so, then run a command on the GPU box
set -e
python __graft_entry__.py >/dev/null
exec /usr/local/graft/bin/gpurun "$@"
