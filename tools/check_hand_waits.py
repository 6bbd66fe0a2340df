"""Static audit of the hand-counted vector-memory waits of the activation-stationary kernels (csrc/panel.hip HAND = true
instances, csrc/img3x3.hip).

The block loop issues its global loads from inline asm (invisible to hipcc's wait-count pass) and waits for them with
hand-written `s_waitcnt vmcnt(N)` statements.  `vmcnt(N)` returns once all but the N youngest vector-memory operations of the
wave have completed, so a load L is known complete at a wait exactly when at least N operations were issued after L.
This script disassembles nothing: it reads hipcc's `-S` output, walks the innermost loop of every HAND instance TWICE
(the second pass models the back edge) and checks for every asm load:
  * the first instruction that reads one of its destination registers comes after an asm `s_waitcnt vmcnt(N)` at which at
    least N younger vector-memory operations had been issued (else: a consumer, or a compiler copy / spill, reads the
    register before the data has landed);
  * hipcc has put no vector-memory wait of its own, no scratch access and no load of its own into the loop.
usage: python tools/check_hand_waits.py [file.s ...]   (no argument: compiles csrc/panel.hip and csrc/img3x3.hip to /tmp first)
exit status 1 on any finding."""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                yield name, body
                name = None


def loop_body(body):
    """(instructions of the MFMA loop in execution order of one iteration, the code behind the loop in text order).
    hipcc marks a loop header `.LBBx_y: ... Loop Header` (for nested loops on comment lines under the label) and every other
    block of the loop `in Loop: Header=BBx_y`; blocks that sit before the header in the text (the latch) run at the end of an
    iteration.  The loop taken is the first (innermost) one whose own blocks hold MFMAs."""
    # split into blocks: (label or None, marker text, lines)
    blocks, cur = [], [None, "", []]
    for ln in body:
        m = re.match(r"^\.L(BB\d+_\d+):(.*)", ln)
        m2 = re.match(r"^; %bb\.\d+:(.*)", ln)
        if m or m2:
            blocks.append(cur)
            cur = [m.group(1) if m else None, (m.group(2) if m else m2.group(1)), []]
        elif re.match(r"^\s+;", ln) and not cur[2]:
            cur[1] += " " + ln.strip()  # continuation of the block's marker comment (nested loops)
        else:
            cur[2].append(ln)
    blocks.append(cur)
    header = None
    for label, mark, lines in blocks:
        if label and "Loop Header" in mark:
            own = [b for b in blocks if b[0] == label or f"in Loop: Header={label} " in b[1] + " "]
            if any("v_mfma" in ln for b in own for ln in b[2]):
                header = label
                break
    if header is None:
        return None
    before, after, tail, seen, done = [], [], [], False, False
    for label, mark, lines in blocks:
        mine = label == header or f"in Loop: Header={header} " in mark + " "
        if label == header:
            seen = True
        if mine and not done:
            (after if seen else before).extend(lines)
        elif seen:
            done = True  # the first foreign block behind the header ends the loop's text
            tail.extend(lines)
    return after + before, tail


def audit(name, body):
    got = loop_body(body)
    if got is None:
        return [f"{name}: no loop found"]

    def instructions(lines):
        out, in_asm = [], False
        for ln in lines:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            out.append((t.split(";")[0].strip(), in_asm))
        return out

    ins, tail = instructions(got[0]), instructions(got[1])
    problems = []
    # after the loop: until an asm wait for everything, hipcc must not touch a register an asm load may still write
    inflight = set()
    for t, a in ins:
        if a and t.startswith("global_load"):
            inflight |= regs_of(t.split(",")[0])
    for t, a in tail:
        if a and re.match(r"s_waitcnt vmcnt\(0\)", t):
            break
        if not a and regs_of(t) & inflight and not t.startswith("s_"):
            problems.append(f"{name}: after the loop `{t}` touches a register an asm load may still be writing "
                            f"(no draining wait names it)")
            break
    for t, a in ins:
        if not a and re.match(r"s_waitcnt.*vmcnt", t):
            problems.append(f"{name}: compiler wait inside the loop: {t}")
        if not a and re.match(r"(global|buffer|flat)_load|scratch_", t):
            problems.append(f"{name}: compiler load / scratch access inside the loop: {t}")
    seq = ins + ins  # second copy = the next iteration
    nasm = 0
    for i, (t, a) in enumerate(ins):
        if not (a and t.startswith("global_load")):
            continue
        nasm += 1
        dst = regs_of(t.split(",")[0])
        younger, covered = 0, False
        for t2, a2 in seq[i + 1:i + 1 + len(ins)]:
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t2) if a2 else None
            if m and younger >= int(m.group(1)):
                covered = True
            if a2 and t2.startswith("global_load") and regs_of(t2.split(",")[0]) & dst:
                if not covered:
                    problems.append(f"{name}: {t.split(',')[0]} is re-loaded before any covering wait")
                break
            ops = t2.split(None, 1)
            srcs = regs_of(ops[1].split(",", 1)[1]) if len(ops) > 1 and "," in ops[1] and not t2.startswith(("global_store", "ds_write")) \
                else regs_of(ops[1]) if len(ops) > 1 and t2.startswith(("global_store", "ds_write", "s_waitcnt")) else set()
            dsts = regs_of(ops[1].split(",", 1)[0]) if len(ops) > 1 and not t2.startswith(("global_store", "ds_write", "s_")) else set()
            if not a2 and dsts & dst and not covered:
                problems.append(f"{name}: `{t2}` overwrites {t.split(',')[0].split()[-1]} while its asm load may still be in flight")
                break
            if not m and srcs & dst and not covered:
                problems.append(f"{name}: `{t2}` reads {t.split(',')[0].split()[-1]} before a covering wait "
                                f"({younger} younger operations issued)")
                break
            if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t2):
                younger += 1
            if covered and srcs & dst:
                break
    if nasm == 0:
        problems.append(f"{name}: no asm load in the loop (is this a HAND instance?)")
    return problems


def main():
    # (source, kernel-name test): every instance whose block / k loop issues its loads from inline asm
    # panel_kernel<T, K, BM, PRO, EPI, HAND, WIDE>: the HAND = true instances (both block-loop forms)
    targets = [("panel.hip", lambda n: "panel_kernel" in n and re.search(r"ELb1ELb[01]EEEvNS_11PanelParamsE$", n) is not None),
               ("img3x3.hip", lambda n: "img3x3_kernel" in n or "img3x3_s2d_kernel" in n)]
    if len(sys.argv) > 1:
        files = [(a, None) for a in sys.argv[1:]]
    else:
        files, jobs = [], []
        for src, _ in targets:  # both compiles at once
            path = f"/tmp/msfwsi_audit_{os.getpid()}_{src}.s"
            jobs.append(subprocess.Popen([os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
                                          "-munsafe-fp-atomics", "-Wno-inline-asm", "-S", "--cuda-device-only",
                                          os.path.join(ROOT, "msf_wsi_amd", "csrc", src), "-o", path],
                                         stderr=subprocess.DEVNULL))
            files.append((path, src))
        for j in jobs:
            if j.wait() != 0:
                raise SystemExit("hipcc failed on a hand-counted source")
    found, n = [], 0
    for path, _ in files:
        lines = open(path).read().splitlines()
        for name, body in kernels(lines):
            if any(test(name) for _, test in targets):
                n += 1
                found += audit(name, body)
    if len(sys.argv) <= 1:
        for path, _ in files:
            os.remove(path)
    print(f"{n} hand-counted instances audited, {len(found)} findings")
    for f in found:
        print("  " + f)
    return 1 if found or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
