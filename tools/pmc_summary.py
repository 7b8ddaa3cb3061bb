"""Fold rocprofv3 counter CSVs (one directory per --pmc pass) into profiles/<round>/pmc_summary.json
and profiles/traffic.json (HBM bytes per unit of work of the two dominant kernels).

    python tools/pmc_summary.py r01 gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_s --ntt-polys 16 --merkle-n 1048576
"""
import argparse, collections, csv, glob, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument("round")
ap.add_argument("dirs", nargs="+")
ap.add_argument("--ntt-polys", type=int, required=True, help="polynomials per pass-kernel launch in the profiled run")
ap.add_argument("--merkle-n", type=int, required=True, help="branches per Merkle launch in the profiled run")
args = ap.parse_args()
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args.dirs:
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            # the instantiations of the pass kernel for one tile size (<10, 1> first pass, <10, 2> last pass, ...) are
            # reported together under the tile size alone, <10>: a launch of either moves one pass's bytes
            k = re.sub(r"(ntt29_pass_kernel<\d+), \d+>", r"\1>", r["Kernel_Name"].split("(")[0].replace("void starkhip::", ""))
            # the Merkle kernel also runs on slices of the batch (the strong-scaling records of bench.py): only the
            # launches over the whole batch are averaged (one lane per branch)
            if k.startswith("merkle_verify_kernel") and int(r["Grid_Size"]) < args.merkle_n:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
os.makedirs(os.path.join(root, "profiles", args.round), exist_ok=True)
json.dump(summary, open(os.path.join(root, "profiles", args.round, "pmc_summary.json"), "w"), indent=1, sort_keys=True)

traffic = {}
for k, units in (("ntt29_pass_kernel<10>", args.ntt_polys), ("merkle_verify_kernel<true>", args.merkle_n)):
    s = summary.get(k, {})
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at
        # 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores
        fetch = 2 * s["FETCH_SIZE"]["avg_per_launch"] * 1024
        write = s["WRITE_SIZE"]["avg_per_launch"] * 1024
        traffic[k] = {"hbm_bytes_per_unit": (fetch + write) / units, "fetch_bytes_per_launch_corrected_x2": fetch,
                      "write_bytes_per_launch": write, "units_per_launch": units, "round": args.round}
json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
