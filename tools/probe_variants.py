#!/usr/bin/env python3
"""Build timing-probe variants of liblidarnerf_hip.so WITHOUT touching the product sources.

    python tools/probe_variants.py <set> [name ...]        (runs here, in the build container; hipcc cross-compiles)

A variant = a copy of lidar-nerf_amd/csrc/ in a temporary directory with a few textual substitutions applied to ONE file,
compiled with -DLNH_VARIANT_TAG="<name>" and linked against the product's other objects into
lidar-nerf_amd/lib/liblidarnerf_hip_<name>.so.  The product translation units hold no probe code; a variant library reports
its name through lnh_build_variant() and lidarnerf/_hip.py refuses it unless LNH_ALLOW_VARIANT=1 (tools/ab.sh sets it).
Results of most variants are WRONG by construction: they exist to be timed (tools/ab.sh), never to be shipped."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lidar-nerf_amd")
sys.path.insert(0, PKG)
import build as product_build  # noqa: E402

KEEP = "if (B == 0xffffffffu) { uint32_t acc = 0; ACC; cursor[tid] = acc; }\n"


def _keep(exprs):
    return KEEP.replace("ACC;", " ".join(f"acc ^= (uint32_t)({e});" for e in exprs))


VAL_WORDS = [f"__builtin_bit_cast(uint32_t, (float)val[{c}][0]) ^ __builtin_bit_cast(uint32_t, (float)val[{c}][1])" for c in range(8)]
# ---- scatter pass of the table-gradient backward cut after each of its phases (profiles/r04_scatter_phases.txt)
SCATTER_CUTS = {
    "cutA": ("grid.hip", [("    // ---- rank inside the workgroup's (bucket) counters",
                           "    " + _keep(VAL_WORDS + ["r0[0] ^ r0[1] ^ r0[2] ^ r0[3]", "emit", "codesh"]) + "    if (B != 0xffffffffu) return;\n"
                           "    // ---- rank inside the workgroup's (bucket) counters")]),
    "cutB": ("grid.hip", [("    // ---- global reservation + exclusive scan of the workgroup's bucket counts (first wave: one counter per lane)\n    if (tid < 64) {\n        static_assert",
                           "    " + _keep(VAL_WORDS + ["rank[0] ^ rank[1] ^ rank[2] ^ rank[3] ^ rank_x[0] ^ rank_x[1] ^ rank_x[2] ^ rank_x[3]", "lcnt[lane]"]) + "    if (B != 0xffffffffu) return;\n"
                           "    // ---- global reservation + exclusive scan of the workgroup's bucket counts (first wave: one counter per lane)\n    if (tid < 64) {\n        static_assert")]),
    "cutC": ("grid.hip", [("    const bool fast = __builtin_amdgcn_readfirstlane((int)lflag) != 0;",
                           "    " + _keep(VAL_WORDS + ["rank[0] ^ rank[1] ^ rank[2] ^ rank[3]", "lox[lane] ^ lstart[lane]"]) + "    if (B != 0xffffffffu) return;\n"
                           "    const bool fast = __builtin_amdgcn_readfirstlane((int)lflag) != 0;")]),
    "cutD": ("grid.hip", [("        // ---- write out: consecutive lanes write consecutive pool slots; all LDS reads",
                           "        " + _keep(["skey[tid] ^ skey[tid + 1024]"]) + "        if (B != 0xffffffffu) return;\n"
                           "        // ---- write out: consecutive lanes write consecutive pool slots; all LDS reads")]),
    "noatomic": ("grid.hip", [("        if (lane < nb && n0) base = atomicAdd(&cursor[fb + lane], n0);\n        const uint32_t incl = wave_scan_add_u32(n0);  // DPP network: no LDS round trips while the atomics are in flight\n        const uint32_t st = incl - n0;\n        lstart[lane] = st;\n        // staging slot pos of bucket bk goes",
                               "        if (lane < nb && n0) base = (blockIdx.y * 97u) % (cap - n0);\n        const uint32_t incl = wave_scan_add_u32(n0);  // DPP network: no LDS round trips while the atomics are in flight\n        const uint32_t st = incl - n0;\n        lstart[lane] = st;\n        // staging slot pos of bucket bk goes")]),
    "full": ("grid.hip", []),
}
SETS = {"scatter": SCATTER_CUTS}


def build_variant(name, fname, subs):
    tmp = tempfile.mkdtemp(prefix=f"lnh_{name}_")
    try:
        dst = os.path.join(tmp, "lidar-nerf_amd", "csrc")
        shutil.copytree(os.path.join(PKG, "csrc"), dst)
        os.makedirs(os.path.join(tmp, "include"))
        for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
            shutil.copy(h, os.path.join(tmp, "include"))
        path = os.path.join(dst, fname)
        text = open(path).read()
        for old, new in subs:
            if text.count(old) != 1:
                raise SystemExit(f"variant {name}: anchor found {text.count(old)} times in {fname}:\n{old[:120]}")
            text = text.replace(old, new)
        open(path, "w").write(text)
        obj = os.path.join(tmp, fname.replace(".hip", ".o"))
        cmd = [product_build.HIPCC] + product_build.FLAGS + product_build.MFMA_VGPR_FORM + \
            [f'-DLNH_VARIANT_TAG="{name}"', "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(f"variant {name}: hipcc failed\n{r.stderr[-3000:]}")
        product_build.build(verbose=False)
        objs = [o for o in sorted(glob.glob(os.path.join(PKG, "lib", "obj", "*.o")))
                if os.path.basename(o) != os.path.basename(obj)] + [obj]
        lib = os.path.join(PKG, "lib", f"liblidarnerf_hip_{name}.so")
        r = subprocess.run([product_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs,
                           capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(f"variant {name}: link failed\n{r.stderr[-3000:]}")
        print(f"[variant] {lib}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    table = SETS[sys.argv[1]]
    names = sys.argv[2:] or list(table)
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(lambda n: build_variant(n, *table[n]), names))


if __name__ == "__main__":
    main()
