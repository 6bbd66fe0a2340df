"""per-shape L2-miss traffic of the dense kernels: parses the two rocprofv3 --pmc passes of scratch/conv_bench.py
(REP=1 -> every kernel of a shape is launched twice: fwd, fwd, dgrad, dgrad, wgrad, wgrad)"""
import collections
import csv
import glob
import sys


def load(pattern):
    f = glob.glob(pattern, recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("igemm_kernel", "conv3x3_kernel", "wgrad_kernel")):
            continue
        key = int(r["Dispatch_Id"])
        per.setdefault(key, [n, 0.0])
        per[key][1] += float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


fetch = load(sys.argv[1] + "/**/*counter_collection.csv")
write = load(sys.argv[2] + "/**/*counter_collection.csv")
assert len(fetch) == len(write), (len(fetch), len(write))
for i in range(0, len(fetch), 2):
    n = fetch[i + 1][0]
    short = n.split("_GLOBAL__N_1")[-1][:60]
    fb, wb = 2 * 1024 * fetch[i + 1][1], 1024 * write[i + 1][1]
    print(f"{i // 2:3d} {short:60s} fetch {fb / 1e6:9.1f} MB  write {wb / 1e6:9.1f} MB")
