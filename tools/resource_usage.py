"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks: one line per kernel (VGPRs, AGPRs, SGPRs, scratch,
occupancy, LDS).  usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> log; python tools/resource_usage.py log
Exit code 1 when a kernel named by --no-scratch PATTERN carries scratch and no --allow PATTERN (matched against the mangled
name) accepts it (make check-scratch)."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True,
                             text=True, check=True).stdout.splitlines()
        return out
    except Exception:
        return names


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and key not in cur:
                cur[key] = int(m.group(1))
    return rows


def main():
    args = sys.argv[1:]
    pats = []
    while "--no-scratch" in args:
        i = args.index("--no-scratch")
        pats.append(args[i + 1])
        del args[i:i + 2]
    allow = []
    while "--allow" in args:
        i = args.index("--allow")
        allow.append(args[i + 1])
        del args[i:i + 2]
    bad = 0
    for path in args:
        rows = parse(path)
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            n = re.sub(r"\(.*", "", n)
            flag = ""
            if r.get("scratch", 0) > 0 and any(re.search(p, n) for p in pats):
                if any(re.search(p, r["name"]) for p in allow):
                    flag = "  (scratch: accepted, see the Makefile)"
                else:
                    flag, bad = "  <-- SCRATCH on a hot-path kernel", bad + 1
            print(f"{r.get('vgpr', -1):4d}v {r.get('agpr', -1):4d}a {r.get('sgpr', -1):4d}s  scratch {r.get('scratch', -1):4d}  "
                  f"occ {r.get('occ', -1)}  lds {r.get('lds', -1):6d}  {n}{flag}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
