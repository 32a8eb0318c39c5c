#!/usr/bin/env python3
"""dev probe: read the schedule the compiler produced for a kernel, one character per instruction.

    python tools/isa_view.py pointrcnn_amd/csrc/mlp.hip                      # scan: kernels whose stores / loads sit behind s_waitcnt vmcnt(0)
    python tools/isa_view.py pointrcnn_amd/csrc/mlp.hip <mangled-name-part>  # schedule of the matching kernels

Legend: M v_mfma  G global load  S global store / atomic  r ds_read  w ds_write  |B| s_barrier  [v(n) l(n)] s_waitcnt
<label> branch  . anything else (runs of >= 6 shown as .{n}).
What it found in round 3: accumulator stores each inside its own bounds-checked region open with
s_waitcnt vmcnt(0) -- on gfx9 that counter includes stores, so they were issued one memory round trip after the other; chunk loads
behind run-time mode branches were serialised the same way; arithmetic on loaded values hoisted above the MFMA loop it was meant to
hide behind."""
import re
import subprocess
import sys
import tempfile


def compile_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def functions(asm):
    for m in re.finditer(r"^(_Z\w+):", asm, re.M):
        end = asm.find(".Lfunc_end", m.start())
        yield m.group(1), asm[m.start():end]


def instrs(body):
    return [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]


def schedule(body):
    out = []
    for ln in body.split("\n"):
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                out.append(" " + t.split(":")[0] + ": ")
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            out.append("M")
        elif op.startswith(("global_load", "buffer_load")):
            out.append("G")
        elif op.startswith(("global_store", "global_atomic")):
            out.append("S")
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            out.append("r")
        elif op.startswith("ds_write") or op.startswith("ds_store"):
            out.append("w")
        elif op.startswith("s_waitcnt"):
            out.append("[" + t.replace("s_waitcnt ", "").replace("vmcnt", "v").replace("lgkmcnt", "l") + "]")
        elif op.startswith("s_barrier"):
            out.append("|B|")
        elif op.startswith(("s_cbranch", "s_branch")):
            out.append("<" + t.split()[-1] + ">")
        else:
            out.append(".")
    txt = re.sub(r"\.{6,}", lambda m: ".{%d}" % len(m.group(0)), "".join(out))
    return re.sub(r"M{6,}", lambda m: "M{%d}" % len(m.group(0)), txt)


def main():
    asm = compile_asm(sys.argv[1])
    if len(sys.argv) > 2:
        for name, body in functions(asm):
            if sys.argv[2] in name:
                print(name)
                print(schedule(body))
                print()
        return
    rows = []
    for name, body in functions(asm):
        ins = instrs(body)
        st = sw = ld = lw = 0
        for k, l in enumerate(ins):
            if l.startswith(("global_store", "global_atomic")):
                st += 1
                sw += any(w.startswith("s_waitcnt") and "vmcnt(0)" in w for w in ins[max(0, k - 14):k])
            if l.startswith("global_load"):
                ld += 1
                lw += any(w.startswith("s_waitcnt") and "vmcnt(0)" in w for w in ins[max(0, k - 6):k])
        if sw >= 6 or lw >= 8:
            rows.append((sw + lw, "%-80s stores behind vmcnt(0): %d/%d   loads right after vmcnt(0): %d/%d" % (name[:80], sw, st, lw, ld)))
    for _, r in sorted(rows, reverse=True):
        print(r)


if __name__ == "__main__":
    main()
