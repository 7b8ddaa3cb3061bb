# How the 2^20 x 64 transform responds to the number of resident wavefronts per SIMD: unused LDS on top of a workgroup's
# own keeps further workgroups off the compute unit. 512-lane workgroups (74 KB): 2 or 1 per CU = 4 or 2 wavefronts per
# SIMD; 256-lane workgroups (STARKHIP_LOGC=0, 37 KB): 4, 3, 2 or 1 per CU.
set -e
run() { echo "== $1"; shift; env "$@" timeout -k 5 120 python tools/bench_one.py 20 64 10 2>/dev/null; }
for rep in 1 2; do
run "512 lanes, 4 waves/SIMD" A=1
run "512 lanes, 2 waves/SIMD" STARKHIP_LDS_PAD=12288
run "256 lanes, 4 waves/SIMD" STARKHIP_LOGC=0
run "256 lanes, 3 waves/SIMD" STARKHIP_LOGC=0 STARKHIP_LDS_PAD=6144
run "256 lanes, 2 waves/SIMD" STARKHIP_LOGC=0 STARKHIP_LDS_PAD=20480
run "256 lanes, 1 wave/SIMD" STARKHIP_LOGC=0 STARKHIP_LDS_PAD=49152
done
