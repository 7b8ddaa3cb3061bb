# Resident batch verification: the branch kernel's residency capped by unused LDS (STARKHIP_GATHER_LDS bytes per
# workgroup of 256 lanes: 0 = 8 workgroups per compute unit, 53248 = 3) and which side the host queues first
# (STARKHIP_BATCH_ORDER=checks: the scalar checks, as before; unset: descriptors and branch launch), alternating in one job
set -e
mkdir -p gpurun_out
out=gpurun_out/gather_lds_ab.txt
: > $out
for n in 1250 10000; do
for rep in 1 2; do
for v in "0 checks" "53248 checks" "53248 branches" "0 branches"; do
  set -- $v
  echo "== proofs=$n STARKHIP_GATHER_LDS=$1 STARKHIP_BATCH_ORDER=$2" >> $out
  STARKHIP_GATHER_LDS=$1 STARKHIP_BATCH_ORDER=$2 timeout -k 10 200 python tools/bench_resident.py $n 30 2>/dev/null >> $out
done
done
done
cat $out
