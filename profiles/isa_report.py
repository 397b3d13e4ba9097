"""Static instruction budget of one kernel of rt_device.hip, attributed to source functions.

Compiles ray_tracer_amd/csrc/rt_device.hip for gfx950 to assembly with line tables (device side only), takes
the kernel whose mangled name contains the given substrings, and attributes every instruction to the source
function whose body holds the line its `.loc` directive names (the innermost inlined location).  Prints the
compiler's resource usage for the kernel and, per source function, the instruction counts by class
(VALU / SALU / branch / scalar memory / LDS / vector memory / waits and nops).  The counts are static (one per
instruction in the binary, however often it runs); together with how often a phase runs per sample they give
a per-phase instruction budget.

usage: python profiles/isa_report.py [--lean] [--keep /tmp/x.s] [--flags "..."] k_renderILi1ELb0
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracer_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -structurizecfg-skip-uniform-regions "
         "-Wno-unused-function --cuda-device-only -S -gline-tables-only -Rpass-analysis=kernel-resource-usage")


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "VMEM"
    if op.startswith("scratch_"):
        return "SCRATCH"
    return "OTHER"


def function_ranges(path):
    """(first line, last line, name) of every function body in a source file, by brace matching from lines that look
    like a definition."""
    src = open(path).read().split("\n")
    out = []
    head = re.compile(r"^(?:template.*>\s*)?(?:RT_HD|__global__|__device__|static|inline|constexpr|extern)\b.*?\b(\w+)\s*\(")
    i = 0
    while i < len(src):
        m = head.match(src[i].strip()) if not src[i].startswith((" ", "\t", "//")) or src[i].lstrip().startswith(("RT_HD", "__device__")) else None
        if m:
            name = m.group(1)
            depth, j, opened = 0, i, False
            while j < len(src):
                line = re.sub(r"//.*", "", src[j])
                for ch in line:
                    if ch == "{":
                        depth += 1
                        opened = True
                    elif ch == "}":
                        depth -= 1
                if opened and depth <= 0:
                    break
                if not opened and line.rstrip().endswith(";"):
                    break
                j += 1
            if opened:
                out.append((i + 1, j + 1, name))
        i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", nargs="+", help="substrings of the mangled kernel name")
    ap.add_argument("--lean", action="store_true", help="compile with -DRT_LEAN_BUILD (fewer instantiations)")
    ap.add_argument("--keep", default="/tmp/rt_device_isa.s")
    ap.add_argument("--reuse", action="store_true", help="do not compile; read --keep")
    ap.add_argument("--flags", default="")
    ap.add_argument("--dump", default="", help="write the kernel's assembly here")
    a = ap.parse_args()
    if not a.reuse:
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS.split() + a.flags.split() + (["-DRT_LEAN_BUILD"] if a.lean else []) + [
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", a.keep, os.path.join(CSRC, "rt_device.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-4000:])
        open(a.keep + ".remarks", "w").write(r.stderr)
    asm = open(a.keep).read()
    remarks = open(a.keep + ".remarks").read() if os.path.exists(a.keep + ".remarks") else ""
    files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', asm)}
    files.update({int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]+)"\s*(?:md5|$)', asm, re.M) if int(m.group(1)) not in files})
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", asm, re.M)]
    match = [n for n in names if all(k in n for k in a.kernel)]
    if len(match) != 1:
        sys.exit(f"{len(match)} kernels match {a.kernel}: {match[:8]}")
    name = match[0]
    body = asm[asm.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    if a.dump:
        open(a.dump, "w").write(body)
    blk = re.search(r"Function Name: " + re.escape(name) + r".*?LDS Size[^\n]*", remarks, re.S)
    if blk:
        for line in blk.group(0).split("\n"):
            m = re.search(r"remark: [^:]*:\d+:\d+: (.*)|remark: (.*)", line)
            t = (m.group(1) or m.group(2)) if m else line
            print("   ", t.strip().split("remark:")[-1].strip())
    ranges = {}
    for fid, path in files.items():
        p = path if os.path.isabs(path) else os.path.join(CSRC, path)
        if os.path.basename(p) in ("rt_core.h", "rt_device.hip", "rt_prepare.h", "rt_lbvh.h") and os.path.exists(p):
            ranges[fid] = (os.path.basename(p), function_ranges(p))
    per = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    cur = ("?", "?")
    for line in body.split("\n"):
        t = line.strip()
        if t.startswith(".loc"):
            parts = t.split()
            fid, ln = int(parts[1]), int(parts[2])
            if fid in ranges:
                base, rs = ranges[fid]
                fn = next((n for (lo, hi, n) in rs if lo <= ln <= hi), f"line{ln}")
                cur = (base, fn)
            else:
                cur = (os.path.basename(files.get(fid, "?")), "(library)")
            continue
        if not t or t.startswith((".", ";", "_Z")) or t.endswith(":"):
            continue
        k = classify(t.split()[0])
        per[cur][k] += 1
        total[k] += 1
    cols = ["VALU", "SALU", "BRANCH", "SMEM", "LDS", "VMEM", "SCRATCH", "WAIT"]
    print(f"\n{name}\n{'function':44s}" + "".join(f"{c:>8s}" for c in cols))
    for key, cnt in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{(key[0] + ':' + key[1])[:44]:44s}" + "".join(f"{cnt[c]:8d}" for c in cols))
    print(f"{'total':44s}" + "".join(f"{total[c]:8d}" for c in cols))


if __name__ == "__main__":
    main()
