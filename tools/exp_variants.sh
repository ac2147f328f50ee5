# usage: bash tools/exp_variants.sh "bench args" tag1 tag2 ...   (tags of tools/build_variant.py builds; "base" = the default library)
ARGS=$1; shift
for t in "$@"; do
  if [ $t = base ]; then unset RACTIP_HOT_LIB; else export RACTIP_HOT_LIB=$PWD/ractip_amd/libractip_hot_$t.so; fi
  python bench.py $ARGS --no-cpu-baseline > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$t failed"; tail -3 gpurun_out/sw.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('$t', round(d['value'],1), round(d['ms_per_step'],2), round(d.get('device_resident_pairs_per_s',0),1), {k:round(v['isolated_ms_per_step'],2) for k,v in d['roofline']['sweeps'].items()})"
done
