# A/B timing of library builds over transform sizes: VARIANTS="a b" bash tools/ab_sizes.sh
set -e
for v in $VARIANTS; do
  export STARKHIP_LIB=$PWD/stark_verifier_amd/libstarkhip_$v.so
  echo "== $v"
  for cfg in "20 64" "16 1024" "24 1" "22 16" "21 32" "18 256" "14 4096" "13 8192" "12 16384" "11 32768" "10 65536" "8 262144" "6 1048576"; do
    set -- $cfg
    timeout -k 5 120 python tools/bench_one.py $1 $2 8 2>/dev/null
    timeout -k 5 120 python tools/bench_one.py $1 $2 8 inv 2>/dev/null
  done
done
