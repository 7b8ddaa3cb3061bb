# Resident batch verification with and without the check kernels' tables (STARKHIP_CHECK_TABLES=0: every lane computes
# its own powers of g2 and walks the constants' polynomial), alternating in one job
set -e
mkdir -p gpurun_out
out=gpurun_out/check_tables_ab.txt
: > $out
for n in 1250 10000; do
for rep in 1 2; do
for v in 0 1; do
  echo "== proofs=$n STARKHIP_CHECK_TABLES=$v" >> $out
  STARKHIP_CHECK_TABLES=$v timeout -k 10 200 python tools/bench_resident.py $n 30 2>/dev/null >> $out
done
done
done
grep -E "==|ms_per_call" $out | sed -e "s/\"ms_per_call_mean.*accept/ accept/"
