"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) into profiles/r01_traffic.json.
FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); both are in KiB."""
import csv, glob, json, sys, collections

def per_kernel(path_glob, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(path_glob, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                out[r["Kernel_Name"]].append((int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), float(r["Counter_Value"])))
    return out

def main(fetch_dir, write_dir, out_json, mfma_dir=None):
    fetch = per_kernel(fetch_dir + "/**/*counter_collection.csv", "FETCH_SIZE")
    write = per_kernel(write_dir + "/**/*counter_collection.csv", "WRITE_SIZE")
    mfma = per_kernel(mfma_dir + "/**/*counter_collection.csv", "MfmaUtil") if mfma_dir else {}
    res = {}
    for label, match in (("gram", "gram_kernel"), ("trailing_update", "gemm_nt_kernel<64, 64, true, 0>"), ("potrf_diag", "potrf_diag_kernel"), ("panel_gemm", "gemm_nt_onepass_kernel")):
        f = [v for k, vs in fetch.items() if match in k for _, v in vs]
        w = [v for k, vs in write.items() if match in k for _, v in vs]
        if not f or not w:
            continue
        fb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)
        res[label] = {"launches": len(f), "fetch_bytes_per_launch_x2_corrected": fb, "write_bytes_per_launch": wb,
                      "hbm_bytes_per_launch": fb + wb}
        u = [v for k, vs in mfma.items() if match in k for _, v in vs]
        if u:
            res[label]["mfma_util_percent_mean"] = sum(u) / len(u)
            res[label]["mfma_util_percent_max"] = max(u)
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:5])
