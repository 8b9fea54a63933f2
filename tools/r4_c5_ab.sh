# round 4: c5 bench value for (library, env) specs on ONE box: spec = default | <variant>[:ENV=V,...]   (boxes differ by +-10 %: compare only within a call)
set -o pipefail
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  for cfg in c5 nss; do
  extra="--steps 3 --warmup 1"; [ $cfg = nss ] && extra=""
  env $envs python3 bench.py --config $cfg $extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$spec $cfg', round(d['value'],3), 'windows/s')"
  done
done
