# c5 bench for the default library and every variant under tools/vt/
for lib in default tools/vt/*.so; do
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/$lib; fi
  python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],3), 'windows/s', round(d['ms_per_step'],1), 'ms')"
done
