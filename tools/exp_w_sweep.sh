for cfg in "RH_DX_W=2" "RH_DX_W=8" "RH_LIN_W=4" "RH_LIN_W=4 RH_DX_W=2"; do
  env $cfg python bench.py --no-cpu-baseline > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$cfg failed"; tail -3 gpurun_out/sw.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('$cfg', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v['isolated_ms_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
done
