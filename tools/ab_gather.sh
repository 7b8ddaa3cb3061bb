# A/B of the batch verifier's launch structure. STARKHIP_BATCH_GATHER=split: a branch launch per bucket instead of one
# for all; STARKHIP_BATCH_OVERLAP=0: the scalar checks behind the branch launch instead of beside it.
set -e
for rep in 1 2 3; do
for v in "split 0" "one 0" "one 1"; do
  set -- $v
  echo "== gather $1 overlap $2"
  STARKHIP_BATCH_GATHER=$1 STARKHIP_BATCH_OVERLAP=$2 timeout -k 5 100 python bench.py --steps 10 --warmup 3 --ntt-batch 4 --cfg2-batch 0 --sharded-log-n 0 --no-cpu-baseline --merkle-n 65536 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('batch %.0f /s  resident %.0f /s (%.3f ms)  parity %s' % (d['batch_verify']['value'], d['batch_verify']['device_resident_value'], d['batch_verify']['device_resident_ms_per_step'], all(d['parity'].values())))"
done
done
for rep in 1 2 3; do
for v in twostep onepass; do
  echo "== resident $v"
  STARKHIP_BATCH_RESIDENT=$v timeout -k 5 100 python bench.py --steps 10 --warmup 3 --ntt-batch 4 --cfg2-batch 0 --sharded-log-n 0 --no-cpu-baseline --merkle-n 65536 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('batch %.0f /s  resident %.0f /s (%.3f ms)  parity %s' % (d['batch_verify']['value'], d['batch_verify']['device_resident_value'], d['batch_verify']['device_resident_ms_per_step'], all(d['parity'].values())))"
done
done
