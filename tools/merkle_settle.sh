# The Merkle branch kernel at two launch sizes, measured over 3 + 10, 30 + 100 and 300 + 300 (untimed + timed) launches:
# how much of a short measurement is the clock ramp after idle.
set -e
mkdir -p gpurun_out
out=gpurun_out/merkle_settle.txt
: > $out
for n in 1048576 131072; do
for ws in "3 10" "30 100" "300 300"; do
  set -- $ws
  echo "== n=$n warmup=$1 steps=$2" >> $out
  BENCH_WARMUP=$1 BENCH_STEPS=$2 timeout -k 10 120 python tools/bench_merkle.py $n 20 32 2>/dev/null | grep -v "tree build" >> $out
done
done
cat $out
