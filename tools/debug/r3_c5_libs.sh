#!/bin/bash
# c5 bench (DiffSTG) for the default library and variants in build_variants/: tools/debug/r3_c5_libs.sh default name1 name2:ENV=V ...
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  env $envs python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$spec', round(d['value'],3), 'windows/s', round(d['ms_per_step'],1), 'ms')"
done
