"""Parse rocprofv3 counter CSVs (separate FETCH_SIZE / WRITE_SIZE passes) into profiles/<name>.json.

usage: python tools/pmc_traffic.py <dir_fetch> <dir_write> <out.json> [kernel substring=k_apply] [contour steps in the run=4]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches: on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact."""
import csv, glob, json, sys

def collect(d, counter, sub):
    tot, n = 0.0, 0
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and sub in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    return tot, n

def main():
    df, dw, out = sys.argv[1:4]
    sub = sys.argv[4] if len(sys.argv) > 4 else "k_apply"
    nsolve = int(sys.argv[5]) if len(sys.argv) > 5 else 4      # bench.py --steps 2 --warmup 1 + the profiling sweep
    f, nf = collect(df, "FETCH_SIZE", sub)
    w, nw = collect(dw, "WRITE_SIZE", sub)
    res = {"kernel": sub, "launches": nf, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "hbm_read_bytes_per_launch": 2.0 * f * 1024 / max(nf, 1), "hbm_write_bytes_per_launch": w * 1024 / max(nw, 1),
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024 / max(nf, 1),
           "contour_steps": nsolve, "traffic_bytes_per_contour_step": (2.0 * f + w) * 1024 / nsolve,
           "note": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE exact"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)

if __name__ == "__main__":
    main()
