# A/B timing of library builds on the Merkle and whole-proof regions: VARIANTS="a b" bash tools/ab_merkle.sh
set -e
for v in $VARIANTS; do
  export STARKHIP_LIB=$PWD/stark_verifier_amd/libstarkhip_$v.so
  echo "== $v"
  timeout -k 5 300 python bench.py --steps 10 --warmup 3 --ntt-batch 4 --cfg2-batch 0 --sharded-log-n 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('merkle %.4f G/s (launch %.1f us)  batch %.0f /s  resident %.0f /s  parity %s' % (d['merkle']['value']/1e9, d['merkle']['roofline']['launch_us'], d['batch_verify']['value'], d['batch_verify']['device_resident_value'], all(d['parity'].values())))"
done
