# round 4: f(x) microbench + c5 / nss records for (library, env) specs on ONE box
set -o pipefail
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  echo "== $spec"; env $envs python3 tools/nst_microbench.py 181 3 f16x3 2>/dev/null | tail -1
  env $envs python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   c5', round(d['value'],3), 'windows/s')"
  env $envs python3 bench.py --config nss --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   nss', round(d['value'],3), 'windows/s')"
done
