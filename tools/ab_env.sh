# A/B of one environment switch on one box: alternates step-only bench runs with VAR=1 and VAR=0.
# usage: bash tools/ab_env.sh VAR [rounds] [steps]
V="$1"; R="${2:-4}"; S="${3:-30}"
for i in $(seq 1 $R); do
  for on in 1 0; do
    env "$V=$on" DEPGAN_BENCH_STEP_ONLY=1 python bench.py --steps $S --warmup 5 2>/dev/null |
      python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$V=$on', d['ms_per_step'], 'ms', d['value'], 'slices/s conv', d['conv_class']['achieved'])"
  done
done
