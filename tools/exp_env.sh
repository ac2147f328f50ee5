# usage: bash tools/exp_env.sh "bench args" "ENV=1 ENV2=.." ...   (one bench run per environment string; "-" = none)
ARGS=$1; shift
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then E=""; else E="$cfg"; fi
  env $E python bench.py $ARGS --no-cpu-baseline > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$cfg failed"; tail -3 gpurun_out/sw.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('$cfg', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v['isolated_ms_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
done
