# timing of the planner's tuning knobs at 2^20 x 64 (environment variables read by csrc/capi.hip)
set -e
run() { echo "== $*"; env "$@" timeout -k 5 120 python tools/bench_one.py 20 64 10 2>/dev/null; }
run DEFAULTS=1
run STARKHIP_LOGC=0
run STARKHIP_SPLIT=7,7,6
run STARKHIP_SPLIT=6,7,7
run STARKHIP_SPLIT=9,11
run STARKHIP_SPLIT=11,9
run STARKHIP_SPLIT=8,12
run STARKHIP_GROUP=-1
run STARKHIP_GROUP=2
run STARKHIP_GROUP=6
run STARKHIP_CHUNK_MB=32
run STARKHIP_CHUNK_MB=64
run STARKHIP_CHUNK_MB=128
run STARKHIP_CHUNK_MB=512
run DEFAULTS=1
