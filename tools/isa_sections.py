#!/usr/bin/env python3
"""Static instruction census of a kernel by source section.

    python tools/isa_sections.py <file.hip> <kernel-name-substring> [-D...]

Compiles the translation unit to gfx950 assembly with -DLNH_ISA_MARKS (csrc/common.h: LNH_MARK("name") becomes an
assembly comment) and counts VALU / SALU / LDS / VMEM instructions between consecutive marks of the kernel whose mangled
name contains the substring.  Static counts: a section holding a cold path (spill handling, a second level class) shows
it too — read the numbers next to the source, and use -DLNH_ONLY_MODE-style switches to isolate one body.  The scatter /
reduce rewrites of round 4 were steered with this (profiles/r04_isa_sections.txt)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-mllvm",
         "-amdgpu-mfma-vgpr-form=1", "-DLNH_ISA_MARKS", "-S", "--cuda-device-only"]


def main():
    src, name = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, src], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        s = open(out).read()
    keep = os.environ.get("ISA_KEEP")
    for m in re.finditer(r"^(_Z[^\n:]*):", s, re.M):
        if name not in m.group(1):
            continue
        k, e = m.start(), s.index(".amdhsa_kernel", m.start())
        body = s[k:e]
        if keep:
            open(keep, "w").write(body)
        vg = re.search(re.escape(m.group(1)) + r"\.num_vgpr, (\d+)", s)
        print(f"{m.group(1)[:100]}  vgpr {vg.group(1) if vg else '?'}")
        sec, order, cnt = "(head)", [], {}
        for ln in body.split("\n"):
            t = ln.strip()
            mm = re.match(r"; MARK (.*)", t)
            if mm:
                sec = mm.group(1)
                continue
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            op = t.split()[0]
            cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                   else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
            if sec not in cnt:
                cnt[sec] = {}
                order.append(sec)
            cnt[sec][cls] = cnt[sec].get(cls, 0) + 1
        tot = {}
        for sec in order:
            c = cnt[sec]
            print(f"   {sec:22s} valu {c.get('valu', 0):4d}  salu {c.get('salu', 0):4d}  lds {c.get('lds', 0):3d}  vmem {c.get('vmem', 0):3d}")
            for a, b in c.items():
                tot[a] = tot.get(a, 0) + b
        print(f"   {'total':22s} valu {tot.get('valu', 0):4d}  salu {tot.get('salu', 0):4d}  lds {tot.get('lds', 0):3d}  vmem {tot.get('vmem', 0):3d}")


if __name__ == "__main__":
    main()
