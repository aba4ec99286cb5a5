for c in 0 64 0 32 64 96 128 192; do
  MDNO_LIB=scripts/micro/exp/libmdno_cache.so MDNO_EXP_CACHE_MIB=$c timeout -k 10 120 python bench.py --steps 50 --warmup 5 --single-mode --skip-cpu-baseline --skip-ensemble-leg --skip-config-legs > gpurun_out/sw_$c.json 2> gpurun_out/sw_$c.err || exit 1
  python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.loads(open(f'gpurun_out/sw_{c}.json').read().strip().splitlines()[-1])
pa=d['rooflines']['conv_factored_moment']['per_application']
print('cache MiB',c,'frames/s %.1f'%d['value'],'ms/step %.4f'%d['ms_per_step'],{k:v for k,v in pa.items() if 'ms' in k})
PY
done
