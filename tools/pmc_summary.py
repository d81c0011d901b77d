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
    # gram: the tile kernels of the K build (round 3: gram_mfma_lds_kernel / gram_mfma_kernel; gram_kernel for periodic kernels) -- the
    # step's cross gram k(x*, x) runs the same kernels on a smaller grid: only the launches of the LARGEST grid are averaged
    for label, match in (("gram", ("gram_mfma", "gram_kernel")), ("trailing_update", ("gemm_nt_kernel<64, 64, true, 0>",)), ("potrf_diag", ("potrf_diag_kernel",)),
                         ("panel_gemm", ("gemm_nt_onepass_kernel",))):
        def pick(table):
            rows = [(g, v) for k, vs in table.items() if any(m in k for m in match) for g, v in vs]
            if label == "gram" and rows:
                gmax = max(g for g, _ in rows)
                rows = [(g, v) for g, v in rows if g == gmax]
            return [v for _, v in rows]
        f, w = pick(fetch), pick(write)
        if not f or not w:
            continue
        fb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)
        res[label] = {"launches": len(f), "fetch_bytes_per_launch_x2_corrected": fb, "write_bytes_per_launch": wb,
                      "hbm_bytes_per_launch": fb + wb}
        u = pick(mfma)
        if u:
            res[label]["mfma_util_percent_mean"] = sum(u) / len(u)
            res[label]["mfma_util_percent_max"] = max(u)
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:5])
