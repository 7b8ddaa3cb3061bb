set -e
for v in $VARIANTS; do
  export STARKHIP_LIB=$PWD/stark_verifier_amd/libstarkhip_$v.so
  echo "== $v"
  timeout -k 5 120 python tools/bench_one.py 20 64 10 2>/dev/null
  timeout -k 5 120 python tools/bench_one.py 12 16384 10 2>/dev/null
done
