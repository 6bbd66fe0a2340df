"""aggregate several rocprofv3 --pmc passes of scratch/conv_bench.py (REP=1: each kernel launched twice, the second
launch is reported) -> one row per dense kernel, one column per counter"""
import collections
import csv
import glob
import sys

tab = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not any(k in n for k in ("igemm_kernel", "conv3x3_kernel", "wgrad_kernel")):
                continue
            per.setdefault(int(r["Dispatch_Id"]), [n, {}])[1][r["Counter_Name"]] = \
                per.get(int(r["Dispatch_Id"]), [n, {}])[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seq = [per[k] for k in sorted(per)]
        for i in range(1, len(seq), 2):
            key = (i // 2, seq[i][0].split("_GLOBAL__N_1")[-1][:48])
            tab.setdefault(key, {}).update(seq[i][1])
for key, cs in tab.items():
    print(key[0], key[1])
    for k, v in cs.items():
        print(f"      {k:40s} {v:16.0f}")
