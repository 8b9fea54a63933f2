# c5 bench under a list of environment settings: bash tools/r3_c5_env.sh "A=1" "B=2 C=3" ...
for e in "" "$@"; do
  env $e python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', round(d['value'],3), 'windows/s', round(d['ms_per_step'],1), 'ms')"
done
