#!/usr/bin/env python3
"""Development aid: per-basic-block instruction mix of one function of a hipcc -S listing.
usage: isa_blocks.py file.s <substring of the mangled name> [min_instr]
Prints, per label-delimited block: instructions, VALU (of which fp64 / MFMA), DS reads / writes, VMEM, scalar memory,
s_waitcnt lines, and the block's branch targets -- enough to find the stage loops of the recursion and to count
what one iteration issues."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    minins = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\w+:", ln) and key in ln:
            start = i
            break
    if start is None:
        sys.exit("function not found")
    end = start + 1
    while end < len(lines) and not lines[end].startswith(".Lfunc_end"):
        end += 1
    print("# %s: lines %d-%d" % (lines[start].split(":")[0], start + 1, end))
    blocks = []
    cur = {"label": "entry", "ins": []}
    for ln in lines[start + 1:end]:
        s = ln.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur["ins"].append(s.split(";")[0].strip())
    blocks.append(cur)
    tot = dict(n=0, valu=0, f64=0, mfma=0, dsr=0, dsw=0, vm=0, sm=0, wait=0, scr=0)
    print("%-14s %6s %6s %5s %5s %5s %5s %5s %5s %5s %5s  branches" % ("block", "ins", "valu", "f64", "mfma", "ds_r", "ds_w", "vmem", "smem", "wait", "scr"))
    for b in blocks:
        c = dict(n=len(b["ins"]), valu=0, f64=0, mfma=0, dsr=0, dsw=0, vm=0, sm=0, wait=0, scr=0)
        br = []
        for s in b["ins"]:
            op = s.split()[0]
            if op.startswith("v_"):
                c["valu"] += 1
                if "_f64" in op:
                    c["f64"] += 1
                if "mfma" in op:
                    c["mfma"] += 1
            elif op.startswith("ds_"):
                if "read" in op or "bpermute" in op or "swizzle" in op or "permute" in op:
                    c["dsr"] += 1
                else:
                    c["dsw"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                c["vm"] += 1
            elif op.startswith("scratch_"):
                c["scr"] += 1
            elif op.startswith(("s_load", "s_buffer_load")):
                c["sm"] += 1
            elif op == "s_waitcnt":
                c["wait"] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                br.append(s.split()[-1])
        for k in tot:
            tot[k] += c[k]
        if c["n"] >= minins:
            print("%-14s %6d %6d %5d %5d %5d %5d %5d %5d %5d %5d  %s" % (b["label"], c["n"], c["valu"], c["f64"], c["mfma"], c["dsr"], c["dsw"], c["vm"], c["sm"], c["wait"], c["scr"], " ".join(br)))
    print("%-14s %6d %6d %5d %5d %5d %5d %5d %5d %5d %5d" % ("TOTAL", tot["n"], tot["valu"], tot["f64"], tot["mfma"], tot["dsr"], tot["dsw"], tot["vm"], tot["sm"], tot["wait"], tot["scr"]))


if __name__ == "__main__":
    main()
