"""profiles/r01_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by
2x (MI355X_MICROARCH.md, HBM section) -> doubled here."""
import collections
import csv
import json
import sys


def family(n):
    if "conv3x3_kernel" in n:
        return "conv_dgrad" if n.rstrip("E").endswith("Lb1") or "Lb1EE" in n else "conv_fwd"
    if "igemm_kernel" in n or "igemm_dma_kernel" in n:
        import re
        flags = re.findall(r"Lb(\d)E", n)
        return "conv_dgrad" if flags and flags[0] == "1" else "conv_fwd"
    if "wgrad_kernel" in n:
        return "conv_wgrad"
    return None


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    ker = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        f = family(n) or "other"
        agg[f][0] += 1
        agg[f][1] += float(r["Counter_Value"])
        if f != "other":
            ker[n][0] += 1
            ker[n][1] += float(r["Counter_Value"])
    return agg, ker


(fetch, kf), (write, kw) = load(sys.argv[1]), load(sys.argv[2])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, python3 bench.py --steps 1 --warmup 1",
       "note": "bytes = 1024*(2*FETCH_SIZE + WRITE_SIZE); FETCH doubled per the gfx950 correction; these are L2-miss "
               "(fabric) bytes: Infinity-Cache hits are included", "families": {}, "kernels": {}}
for k in fetch:
    n = fetch[k][0]
    fb, wb = 2 * 1024 * fetch[k][1], 1024 * write.get(k, [0, 0.0])[1]
    out["families"][k] = {"launches": n, "fetch_bytes": fb, "write_bytes": wb,
                          "hbm_bytes_per_launch": (fb + wb) / max(1, n)}
for k in kf:
    n = kf[k][0]
    fb, wb = 2 * 1024 * kf[k][1], 1024 * kw.get(k, [0, 0.0])[1]
    out["kernels"][k] = {"launches": n, "fetch_bytes": fb, "write_bytes": wb,
                         "hbm_bytes_per_launch": (fb + wb) / max(1, n)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["families"], indent=1))
