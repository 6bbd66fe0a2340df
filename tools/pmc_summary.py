"""Summarise the rocprofv3 --pmc passes of tools/profile_step.sh into one JSON (committed under profiles/):

    python3 tools/pmc_summary.py <FETCH dir> <WRITE dir> <MFMA dir> <LDS dir> <kernel_stats.csv> <out.json>

Per kernel symbol (and per family conv_fwd / conv_dgrad / conv_wgrad / other):
  hbm_bytes_per_launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) / launches.  FETCH_SIZE / WRITE_SIZE count KB at the L2's
      memory side (fabric requests, Infinity-Cache hits included); on gfx950 FETCH_SIZE reports exactly half of the
      bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled.
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
      counter is summed over the 8 XCDs); SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16 MFMA).
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles)
  wait_frac = SQ_WAIT_ANY / (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY)
Counters of a pass are summed over the launches of a symbol in that pass (bench.py --steps 1 --warmup 1)."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def provenance():
    """which build these counters belong to: the build id of the library that was profiled (sha256 of the kernel sources,
    baked into the binary: msf_wsi_amd/_lib.built_id) and the commit (git where a work tree exists -- the GPU box has none,
    the caller passes MSFWSI_GIT_HEAD there).  bench.py replays a summary only into a run of the SAME build id."""
    from msf_wsi_amd import _lib

    head = os.environ.get("MSFWSI_GIT_HEAD")
    if not head:
        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True,
                                  timeout=10).stdout.strip() or None
        except (OSError, subprocess.SubprocessError):
            head = None
    return {"build_id": _lib.built_id(os.environ.get("MSFWSI_LIB")), "source_id": _lib.source_id(), "git_head": head}


def family(n):
    if "conv3x3_kernel" in n:
        return "conv_dgrad" if "Lb1EE" in n else "conv_fwd"
    if "igemm_kernel" in n or "igemm_dma_kernel" in n:
        flags = re.findall(r"Lb(\d)E", n)
        return "conv_dgrad" if flags and flags[0] == "1" else "conv_fwd"
    if "wgrad_kernel" in n:
        return "conv_wgrad"
    return "other"


def load(d):
    """{kernel: {counter: sum, '_n': dispatches}}"""
    tab = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            tab[n][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[n].add(r["Dispatch_Id"])
    for n in tab:
        tab[n]["_n"] = len(seen[n])
    return tab


def short(n):
    m = re.search(r"_GLOBAL__N_1\d+(.*?)EvNS", n)
    if m:
        return m.group(1)
    n = re.sub(r"^void\s+", "", n).replace("(anonymous namespace)::", "")  # demangled names of non-template kernels
    return n.split("(")[0].split("<")[0][:80] or n[:80]


def main():
    fd, wd, md, ld, stats_csv, out_path = sys.argv[1:7]
    F, W, M, L = load(fd), load(wd), load(md), load(ld)
    times = {}
    for r in csv.DictReader(open(stats_csv)):
        times[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]))
    out = {"source": "tools/profile_step.sh: rocprofv3 --pmc passes of `python3 bench.py --steps 1 --warmup 1` (2 steps per "
                     "pass) + --kernel-trace --stats of `--steps 4 --warmup 2`",
           "note": "hbm bytes = 1024*(2*FETCH_SIZE + WRITE_SIZE): FETCH doubled per the gfx950 correction; L2-miss "
                   "(fabric) bytes, Infinity-Cache hits included",
           "families": {}, "kernels": {}}
    out.update(provenance())
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for n in F:
        launches = F[n]["_n"]
        fb, wb = 2 * 1024 * F[n]["FETCH_SIZE"], 1024 * W.get(n, {}).get("WRITE_SIZE", 0.0)
        k = {"launches_in_pass": launches, "fetch_bytes": fb, "write_bytes": wb,
             "hbm_bytes_per_launch": (fb + wb) / max(1, launches)}
        m = M.get(n)
        if m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            k["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc)
            k["sq_busy_frac"] = m["SQ_BUSY_CYCLES"] / (8 * cyc) if m.get("SQ_BUSY_CYCLES") else None
            k["mfma_insts_per_launch"] = m["SQ_INSTS_MFMA"] / max(1, m["_n"])
        l_ = L.get(n)
        if l_:
            if l_.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
                k["lds_conflict_frac"] = l_["SQ_LDS_BANK_CONFLICT"] / l_["SQ_LDS_IDX_ACTIVE"]
            tot = l_.get("SQ_WAIT_ANY", 0) + l_.get("SQ_WAIT_INST_ANY", 0) + l_.get("SQ_ACTIVE_INST_ANY", 0)
            if tot > 0:
                k["wait_frac"] = l_["SQ_WAIT_ANY"] / tot
                k["issue_stall_frac"] = l_["SQ_WAIT_INST_ANY"] / tot
        if n in times:
            k["kernel_trace"] = {"calls": times[n][0], "avg_ms": times[n][2] / 1e6, "pct_of_gpu_time": times[n][3]}
        f = family(n)
        fam[f]["launches"] += launches
        fam[f]["fetch_bytes"] += fb
        fam[f]["write_bytes"] += wb
        if f != "other" or (n in times and times[n][3] > 0.5):
            out["kernels"][short(n)] = k
    for f, v in fam.items():
        out["families"][f] = dict(v, hbm_bytes_per_launch=(v["fetch_bytes"] + v["write_bytes"]) / max(1, v["launches"]))
    tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in fam.values())
    out["hbm_bytes_per_pass"] = tot
    out["hbm_bytes_per_step"] = tot / 2.0  # --steps 1 --warmup 1: two identical steps in the pass
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({"hbm_TB_per_step": tot / 2e12, "families": out["families"]}, indent=1)[:1500])
    top = sorted(((v.get("kernel_trace", {}).get("pct_of_gpu_time", 0), k) for k, v in out["kernels"].items()), reverse=True)
    for pct, k in top[:12]:
        v = out["kernels"][k]
        print(f"{pct:5.1f}%  {k[:70]:70s} mfma_util {v.get('mfma_util', float('nan')):.3f}  "
              f"lds_conflict {v.get('lds_conflict_frac', float('nan')):.3f}  wait {v.get('wait_frac', float('nan')):.2f}  "
              f"GB/launch {v['hbm_bytes_per_launch'] / 1e9:.3f}")


if __name__ == "__main__":
    main()
