# rocprofv3 runs behind profiles/r03 (run on the GPU box from the repository root; each pass is its own run)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_r03.json 2> $O/bench_r03.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py > $O/bench_under_rocprof_r03.json 2> $O/kt.err
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --proofs 0 --cfg2-batch 0 --sharded-log-n 0"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/pmc_f.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/pmc_w.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_s -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/pmc_s.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/pmc_l.err
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/pmc_a -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/pmc_a.err
find $O/kt -name "*kernel_stats.csv" | head -2
echo profiled
