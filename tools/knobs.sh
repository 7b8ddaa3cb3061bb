# timing of the planner's tuning knobs at 2^20 x 64 (environment variables read by csrc/capi.hip)
set -e
run() { echo "== $*"; env "$@" timeout -k 5 120 python tools/bench_one.py 20 64 10 2>/dev/null; }
run X=1
run STARKHIP_CHUNK_MB=32
run STARKHIP_CHUNK_MB=64
run STARKHIP_CHUNK_MB=128
run STARKHIP_CHUNK_MB=256
run STARKHIP_CHUNK_MB=512
run STARKHIP_CHUNK_MB=1024
run X=1
