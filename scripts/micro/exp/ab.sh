run() {
  timeout -k 10 120 python bench.py --steps 50 --warmup 5 --single-mode --skip-cpu-baseline --skip-ensemble-leg --skip-config-legs "${@:2}" > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
pa=d['rooflines']['conv_factored_moment']['per_application']
print(sys.argv[1],'frames/s %.1f'%d['value'],'ms/step %.4f'%d['ms_per_step'],pa['k1_k2_k3_ms'])
PY
}
run new-1member
export MDNO_LIB=scripts/micro/exp/libmdno_prev.so; run prev-1member
unset MDNO_LIB; run new-1member
export MDNO_LIB=scripts/micro/exp/libmdno_prev.so; run prev-1member
unset MDNO_LIB; run new-8members --total-members 8
export MDNO_LIB=scripts/micro/exp/libmdno_prev.so; run prev-8members --total-members 8
unset MDNO_LIB; run new-64members --total-members 64 --steps 10
export MDNO_LIB=scripts/micro/exp/libmdno_prev.so; run prev-64members --total-members 64 --steps 10
