"""Static audit of the hand-counted vector-memory waits of the panel kernels (csrc/panel.hip, HAND = true instances).

The block loop issues its global loads from inline asm (invisible to hipcc's wait-count pass) and waits for them with
hand-written `s_waitcnt vmcnt(N)` statements.  `vmcnt(N)` returns once all but the N youngest vector-memory operations of the
wave have completed, so a load L is known complete at a wait exactly when at least N operations were issued after L.
This script disassembles nothing: it reads hipcc's `-S` output, walks the innermost loop of every HAND instance TWICE
(the second pass models the back edge) and checks for every asm load:
  * the first instruction that reads one of its destination registers comes after an asm `s_waitcnt vmcnt(N)` at which at
    least N younger vector-memory operations had been issued (else: a consumer, or a compiler copy / spill, reads the
    register before the data has landed);
  * hipcc has put no vector-memory wait of its own, no scratch access and no load of its own into the loop.
usage: python tools/check_hand_waits.py [panel.s]      (no argument: compiles msf_wsi_amd/csrc/panel.hip to /tmp first)
exit status 1 on any finding."""
import os
import re
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
    """instructions of the MFMA loop in execution order of one iteration: hipcc labels every block of a loop with
    `in Loop: Header=BBx_y`; blocks that sit before the header in the text (the latch) run at the end of an iteration"""
    header = None
    for i, ln in enumerate(body):
        m = re.match(r"^\.L(BB\d+_\d+):.*Loop Header", ln)
        if m:
            for nxt in body[i + 1:]:
                if re.match(r"^\.LBB", nxt):
                    break
                if "v_mfma" in nxt:
                    header = m.group(1)
                    break
        if header:
            break
    if header is None:
        return None
    before, after, cur, seen_header = [], [], None, False
    for ln in body:
        m = re.match(r"^\.L(BB\d+_\d+):(.*)", ln)
        if m:
            if m.group(1) == header:
                cur, seen_header = after, True
            elif f"Header={header} " in m.group(2) + " ":
                cur = after if seen_header else before
            else:
                cur = None
            continue
        if cur is not None:
            cur.append(ln)
    return after + before


def audit(name, body):
    loop = loop_body(body)
    if loop is None:
        return [f"{name}: no loop found"]
    ins, in_asm = [], False
    for ln in loop:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((t.split(";")[0].strip(), in_asm))
    problems = []
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
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/msfwsi_panel_audit.s"
        src = os.path.join(ROOT, "msf_wsi_amd", "csrc", "panel.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                        "-Wno-inline-asm", "-S", "--cuda-device-only", src, "-o", path], check=True,
                       stderr=subprocess.DEVNULL)
    lines = open(path).read().splitlines()
    found, n = [], 0
    for name, body in kernels(lines):
        if "panel_kernel" in name and name.endswith("Lb1EEEvNS_11PanelParamsE"):
            n += 1
            found += audit(name, body)
    print(f"{n} hand-counted instances audited, {len(found)} findings")
    for f in found:
        print("  " + f)
    return 1 if found or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
