"""Summarise the two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate passes, --kernel-trace only) into
profiles/r6_pmc_<config>.json, stamped with the hash of the kernel sources and the batch size, so bench.py's
`roofline.traffic` can never quote a pass measured on another build.
    python tools/pmc_summary.py <config> <pairs_per_gpu> <fetch_dir> <write_dir> [out_dir]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def counter_total(d, name, kernel):
    """counter `name` per launch of the BATCH kernel (the dispatches of `kernel` with the largest grid: bench.py also
    times a few single-pair calls), summed over the rows of a dispatch (one row per XCD / instance)"""
    per = {}; grid = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == name and kernel in row.get("Kernel_Name", ""):
                k = row.get("Dispatch_Id")
                per[k] = per.get(k, 0.0) + float(row["Counter_Value"]); grid[k] = int(row.get("Grid_Size", 0))
    if not per:
        raise SystemExit(f"no {name} rows for {kernel} under {d}")
    g = max(grid.values())
    vals = [v for k, v in per.items() if grid[k] == g]
    return sum(vals) / len(vals), len(vals)


if __name__ == "__main__":
    cfg, pairs, fdir, wdir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    out_dir = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles")
    kern = "dg_find_homography_kernel" if bench.CONFIGS[cfg]["which"] == "H" else "dg_find_fundamental_kernel"
    fetch, nf = counter_total(fdir, "FETCH_SIZE", kern)
    write, nw = counter_total(wdir, "WRITE_SIZE", kern)
    out = {"config": cfg, "pairs_per_gpu": pairs, "source_id": bench.source_id(), "kernel": kern,
           "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "dispatches": [nf, nw],
           "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
           "note": "per launch; 2 x FETCH_SIZE + WRITE_SIZE KiB (MI355X_MICROARCH.md gfx950 correction)"}
    path = os.path.join(out_dir, f"r6_pmc_{cfg}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, json.dumps(out))
