#!/bin/bash
# Rehearsal of bench.py's paths on one card: the default single-rank run, two ranks over gloo sharing the card, and
# two ranks of which one fails alone in the sharded region (rank 0 must still print the line, and the job must end with
# a non-zero status and the region named in "regions_failed").
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 > gpurun_out/rehearse_n1.json 2> gpurun_out/rehearse_n1.err || { echo "n1 failed"; exit 1; }
echo "n1 ok"
export STARKHIP_BENCH_BACKEND=gloo
SMALL="--gpus 2 --steps 2 --warmup 1 --ntt-batch 32 --merkle-n 65536 --merkle-total 65536 --proofs 50 --proofs-total 100 --sharded-log-n 22"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py $SMALL > gpurun_out/rehearse_n2.json 2> gpurun_out/rehearse_n2.err || { echo "n2 failed"; exit 1; }
echo "n2 ok"
STARKHIP_BENCH_FAIL=sharded_ntt:1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py $SMALL --region-timeout 0.2 > gpurun_out/rehearse_n2_fail1.json 2> gpurun_out/rehearse_n2_fail1.err && { echo "n2 fail1: status 0 although a region failed"; exit 1; }
grep -q '"regions_failed": \["sharded_ntt"\]' gpurun_out/rehearse_n2_fail1.json || { echo "n2 fail1: no line or no regions_failed"; exit 1; }
echo "n2 fail1 ok"
STARKHIP_BENCH_FAIL=sharded_ntt:0 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py $SMALL --region-timeout 0.2 > gpurun_out/rehearse_n2_fail0.json 2> gpurun_out/rehearse_n2_fail0.err && { echo "n2 fail0: status 0 although a region failed"; exit 1; }
grep -q '"regions_failed": \["sharded_ntt"\]' gpurun_out/rehearse_n2_fail0.json || { echo "n2 fail0: no line or no regions_failed"; exit 1; }
echo "n2 fail0 ok"
