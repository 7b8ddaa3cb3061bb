# A/B timing of library builds: VARIANTS="a b" bash tools/ab.sh  (stark_verifier_amd/libstarkhip_<variant>.so)
set -e
for v in $VARIANTS; do
  export STARKHIP_LIB=$PWD/stark_verifier_amd/libstarkhip_$v.so
  echo "== $v"
  timeout -k 5 120 python tools/bench_one.py 20 64 10 2>/dev/null
  if [ -z "$ONLY20" ]; then
    timeout -k 5 120 python tools/bench_one.py 16 1024 10 2>/dev/null
    timeout -k 5 120 python tools/bench_one.py 24 1 10 2>/dev/null
  fi
done
