# HBM traffic of the pass kernel with the strided and the tile-by-tile intermediate (one --pmc pass per run)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --proofs 0 --cfg2-batch 0 --sharded-log-n 0"
for m in 0 1; do
  export STARKHIP_TILED_MID=$m
  rm -rf $O/tab_f$m $O/tab_w$m
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/tab_f$m -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/tab_f$m.err
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/tab_w$m -o p --output-format csv -- python3 bench.py $BARGS > /dev/null 2> $O/tab_w$m.err
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for m in (0, 1):
    for c, d in (("FETCH_SIZE", "tab_f%d" % m), ("WRITE_SIZE", "tab_w%d" % m)):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "pass_kernel<10" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[c].append(float(r["Counter_Value"]))
        v = acc[c]
        # launches alternate pass 1, pass 2
        print("tiled_mid=%d %s KiB per launch: pass1 %.0f pass2 %.0f (n=%d)" % (m, c, sum(v[0::2]) / max(1, len(v[0::2])), sum(v[1::2]) / max(1, len(v[1::2])), len(v)))
PY
