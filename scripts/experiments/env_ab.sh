#!/bin/bash
# A/B of bench.py under environment switches: env_ab.sh "<label>=<ENV assignments>" ... ; every arm runs posit and fp8 outliers
run() { env $2 python bench.py --no-cpu-baseline --steps 300 ${@:3} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['value'],1), 'TF', round(j['roofline']['kernel_ms']*1000,1), 'us')"; }
for rep in 1 2; do
for arm in "$@"; do
  label="${arm%%=*}"; envs="${arm#*=}"
  run "$label posit" "$envs"
  run "$label fp8" "$envs" --outlier fp8_e4m3
done
done
