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
    # A: load, locate, rows, merge masks.  B: + rank atomics, values, scan, barrier.  C: + cursor atomics issued, local scan,
    # barrier.  D: + staging, reservation completed, barrier.  full: + write-out.
    "cutA": ("grid.hip", [('    LNH_MARK("F rank");', "    " + _keep(["r0[0] ^ r0[1] ^ r0[2] ^ r0[3]", "emit", "codesh", "__builtin_bit_cast(uint32_t, fr0 + fr1 + fr2 + g0 + g1)"]) + "    if (B != 0xffffffffu) return;\n")]),
    "cutB": ("grid.hip", [('    LNH_MARK("G reserve");', "    " + _keep(VAL_WORDS + ["rank[0] ^ rank[1] ^ rank[2] ^ rank[3] ^ rank_x[0] ^ rank_x[1] ^ rank_x[2] ^ rank_x[3]", "lcnt[lane]"]) + "    if (B != 0xffffffffu) return;\n")]),
    "cutC": ("grid.hip", [('    LNH_MARK("H stage");', "    " + _keep(VAL_WORDS + ["rank[0] ^ rank[1] ^ rank[2] ^ rank[3]", "rs_base ^ lstart[lane]"]) + "    if (B != 0xffffffffu) return;\n")]),
    "cutD": ("grid.hip", [('    LNH_MARK("I writeout");', "    " + _keep(["skey[tid] ^ skey[tid + 1024] ^ lox[lane]"]) + "    if (B != 0xffffffffu) return;\n")]),
    "noatomic": ("grid.hip", [("        if (lane < nb && rs_n0) rs_base = atomicAdd(&cursor[fb + lane], rs_n0);",
                               "        if (lane < nb && rs_n0) rs_base = (blockIdx.y * 97u) % (cap - rs_n0);")]),
    "full": ("grid.hip", []),
}

# ---- phase-lock breakers: the second workgroup of every CU starts late once (the offset then persists)
def _stagger(n):
    return ("grid.hip", [("    if (!cls_hash && !cls_dense) return;  // generic class: k_grid_bwd_scatter's level\n",
                          "    if (!cls_hash && !cls_dense) return;  // generic class: k_grid_bwd_scatter's level\n"
                          "    { const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x;\n"
                          "      if (lin >= 256u && lin < 512u) __builtin_amdgcn_s_sleep(%d); }\n" % n)])


STAGGER = {"stag40": _stagger(40), "stag80": _stagger(80), "stag120": _stagger(120), "full": ("grid.hip", [])}

# ---- a workgroup walks N consecutive chunks of the same level (the stores of one chunk drain under the next chunk's work)
def _iters(n):
    return ("grid.hip", [
        ("    const uint32_t level = level0 + blockIdx.x, chunk = blockIdx.y;",
         "    const uint32_t level = level0 + blockIdx.x; uint32_t chunk = 0;"),
        ("    if (cls_hash) body(std::integral_constant<int, 1>{});\n    else body(std::integral_constant<int, 2>{});\n#endif\n}\n\n// A bucket with many entries",
         "    for (uint32_t it = 0; it < %du; it++) {\n        chunk = blockIdx.y * %du + it;\n        if (chunk * (uint32_t)NTHREADS >= B) break;\n"
         "        if (it) __syncthreads();\n"
         "        if (cls_hash) body(std::integral_constant<int, 1>{});\n        else body(std::integral_constant<int, 2>{});\n    }\n#endif\n}\n\n// A bucket with many entries" % (n, n)),
        ("dim3(n_win, div_up(B, kScatterThreads)),", "dim3(n_win, div_up(div_up(B, kScatterThreads), %d))," % n)])


ITERS = {"it2": _iters(2), "it4": _iters(4), "it13": _iters(13), "full": ("grid.hip", [])}

# ---- scatter workgroups of 512 / 256 threads (4 / 8 per CU instead of 2; the cursor atomics double / quadruple)
def _threads(n):
    return ("grid.hip", [("#define LNH_SCATTER_THREADS 1024", "#define LNH_SCATTER_THREADS %d" % n)])


WGSIZE = {"t512": _threads(512), "t256": _threads(256), "full": ("grid.hip", [])}

# ---- reduce pass with one ingredient removed at a time (profiles/r04_reduce_variants.txt)
REDUCE = {
    # the pool stream and the conversions, no LDS add (the values are folded into one word per lane instead)
    "r_noadd": ("grid.hip", [("        if constexpr (sizeof(T) == 2) atomicAdd(&img[idx], (unsigned long long)half_to_fixed24(v));",
                              "        if constexpr (sizeof(T) == 2) { probe_acc ^= (unsigned long long)half_to_fixed24(v) + idx; }"),
                             ("    auto acc_add = [&](uint32_t idx, T v) {", "    unsigned long long probe_acc = 0;\n    auto acc_add = [&](uint32_t idx, T v) {"),
                             ("    if (hashed) stream(std::true_type{});\n    else stream(std::false_type{});\n",
                              "    if (hashed) stream(std::true_type{});\n    else stream(std::false_type{});\n    if (probe_acc == 0x123456789abcull) img[threadIdx.x] = probe_acc;\n")]),
    # the pool stream and the LDS adds, no conversion (the raw half bits are added)
    "r_noconv": ("grid.hip", [("        if constexpr (sizeof(T) == 2) atomicAdd(&img[idx], (unsigned long long)half_to_fixed24(v));",
                               "        if constexpr (sizeof(T) == 2) atomicAdd(&img[idx], (unsigned long long)__builtin_bit_cast(unsigned short, v));")]),
    # conversions and LDS adds on the first quad of the slice over and over, no pool stream
    "r_noload": ("grid.hip", [("                const uint32_t j = j0 + u * blockDim.x, q = qf_begin + (j < nfull ? j : nfull - 1);  // clamped, unconditional",
                               "                const uint32_t j = j0 + u * blockDim.x, q = qf_begin + ((j < nfull ? j : nfull - 1) & 1023u);")]),
    "full": ("grid.hip", []),
}

# ---- reduce slices: entries per slice (the makespan of the pass is its longest slice)
def _slice(n):
    return ("grid.hip", [("constexpr uint32_t kSliceEntries = 512 * 1024;", "constexpr uint32_t kSliceEntries = %d * 1024;" % n)])


SLICES = {"s384": _slice(384), "s256": _slice(256), "s192": _slice(192), "s128": _slice(128), "s96": _slice(96), "full": ("grid.hip", [])}

# ---- encode forward: the two probes of rounds 2 / 3, formerly #if branches inside k_grid_forward
# fusion (profiles/r03_fusion_probe.txt): the level loop INSIDE the workgroup (grid.y = 1) — what an encode -> MLP fusion
# gives up: the level-major launch order that keeps ONE level's ~2 MB table in each XCD's L2.
FUSION = {
    "fwdloop": ("grid.hip", [
        ('    {  // (tools/probe_variants.py "fusion" turns this block into a loop over the levels: the encode -> MLP fusion probe)\n    const uint32_t level = blockIdx.y;\n',
         "    for (uint32_t level = 0; level < L; level++) {\n"),
        ("    dim3 grid(div_up(B, 256), /* levels */ L), block(256);", "    dim3 grid(div_up(B, 256), 1), block(256);")]),
    "product": ("grid.hip", []),
}


# gather cost model (profiles/r02_gather_probe.txt): the second load of every lane of a hashed level goes to
#   1 row 0 (one line for the whole wave)   2 nowhere (no instruction)   3 the other 16-byte half of the 32-byte block it
#   already fetched   4 r1 itself (same 128-byte line for 31 lanes in 32)   5 another random line.  Results are WRONG.
def _gather(n):
    far_row = {1: "0u", 3: "((r0 & ~3u) ^ 4u)", 4: "r1", 5: "((r1 * 2654435761u) & (lv.hashmap_size - 1))"}
    load = "solo[yz] = Vec<T, C>{};" if n == 2 else f"solo[yz] = load_vec<T, C>(tab + (size_t){far_row[n]} * C);"
    return ("grid.hip", [
        ('                    solo[yz] = load_vec<T, C>(tab + (size_t)(far ? r1 : 0u) * C);  // (probe anchor: tools/probe_variants.py "gather")\n',
         "                    " + load + "\n"),
        ("                    const uint32_t b1 = far ? so : a1;\n",
         "                    const uint32_t b1 = (r0s[yz] & 64u) ? so : a1;  // (keeps the probe's second load alive)\n")])


GATHER = dict({"probe0": ("grid.hip", [])}, **{f"probe{n}": _gather(n) for n in (1, 2, 3, 4, 5)})

# ---- colour-head backward: one wave per SIMD with the weights in registers (round 3) against two waves per SIMD with
# the weights in LDS (round 4)
COLOR = {"c1wave": ("lidar_color.hip", []),
         "c2wave": ("lidar_color.hip", [("#define LNH_COLOR_BWD_LDSW 0", "#define LNH_COLOR_BWD_LDSW 1")])}

# ---- chunk length of the bucketed backward (workspace size against per-chunk launch costs)
CHUNK = {"ch2m": ("grid.hip", [("constexpr uint32_t kChunkPoints = 4u << 20;", "constexpr uint32_t kChunkPoints = 2u << 20;")]),
         "ch1m": ("grid.hip", [("constexpr uint32_t kChunkPoints = 4u << 20;", "constexpr uint32_t kChunkPoints = 1u << 20;")]),
         "full": ("grid.hip", [])}
# ---- non-temporal hints on the buffers the encoder streams (written once / read once): do they keep the level's table in L2?
_NT_OUT = ("    store_vec<T, C>(out, res);\n    if constexpr (DYDX) {",
           "    if constexpr (sizeof(T) * C == 4) { uint32_t raw_o; __builtin_memcpy(&raw_o, &res, 4); "
           "__builtin_nontemporal_store(raw_o, reinterpret_cast<uint32_t *>(out)); } else store_vec<T, C>(out, res);\n    if constexpr (DYDX) {")
_NT_IN = ("    float x[D];\n#pragma unroll\n    for (int d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];\n    const T *tab = table",
          "    float x[D];\n#pragma unroll\n    for (int d = 0; d < D; d++) x[d] = __builtin_nontemporal_load(inputs + (size_t)b * D + d);\n    const T *tab = table")
_NT_SIN = ("    float x0 = px[0], x1 = px[1], x2 = px[2];",
           "    float x0 = __builtin_nontemporal_load(px), x1 = __builtin_nontemporal_load(px + 1), x2 = __builtin_nontemporal_load(px + 2);")
_NT_POOL = [("                    *reinterpret_cast<Pair2 *>(pvals + slot * (uint32_t)sizeof(Pair)) = pr;",
             "                    if constexpr (sizeof(Pair2) == 16) { typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4))); "
             "__builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, pr), reinterpret_cast<u32x4_nt *>(pvals + slot * (uint32_t)sizeof(Pair))); } "
             "else *reinterpret_cast<Pair2 *>(pvals + slot * (uint32_t)sizeof(Pair)) = pr;"),
            ("                    *reinterpret_cast<uint32_t *>(prows + slot * 2u) = (ks2[i][0] & 0xffffu) | (ks2[i][1] << 16);",
             "                    __builtin_nontemporal_store((ks2[i][0] & 0xffffu) | (ks2[i][1] << 16), reinterpret_cast<uint32_t *>(prows + slot * 2u));")]
_RD_NORMAL = [("                for (uint32_t w = 0; w < VQ; w++) v[u][w] = __builtin_nontemporal_load(vals4 + (size_t)q * VQ + w);",
               "                for (uint32_t w = 0; w < VQ; w++) v[u][w] = vals4[(size_t)q * VQ + w];"),
              ("                r[u] = __builtin_nontemporal_load(rows4 + q);", "                r[u] = rows4[q];")]
NTMEM = {"full": ("grid.hip", []),
         "ntvals": ("grid.hip", _NT_POOL[:1]),
         "ntrows": ("grid.hip", _NT_POOL[1:]),
         "rdnorm": ("grid.hip", _RD_NORMAL),
         "ntout": ("grid.hip", [_NT_OUT]),
         "ntin": ("grid.hip", [_NT_IN]),
         "ntfwd": ("grid.hip", [_NT_OUT, _NT_IN]),
         "ntsin": ("grid.hip", [_NT_SIN]),
         "ntpool": ("grid.hip", _NT_POOL),
         "ntall": ("grid.hip", [_NT_OUT, _NT_IN, _NT_SIN] + _NT_POOL)}
# ---- workgroup size of the encode forward (256 threads in the product)
def _fwd_threads(n):
    return ("grid.hip", [("// ------------------------------------------------------------------------------------------------ forward\ntemplate <typename T, int D, int C, bool DYDX>\n__global__ void __launch_bounds__(256)",
                          "// ------------------------------------------------------------------------------------------------ forward\ntemplate <typename T, int D, int C, bool DYDX>\n__global__ void __launch_bounds__(%d)" % n),
                         ("    dim3 grid(div_up(B, 256), /* levels */ L), block(256);", "    dim3 grid(div_up(B, %d), /* levels */ L), block(%d);" % (n, n))])


FWDWG = {"full": ("grid.hip", []), "fw64": _fwd_threads(64), "fw128": _fwd_threads(128), "fw512": _fwd_threads(512), "fw1024": _fwd_threads(1024)}
SETS = {"fwdwg": FWDWG, "ntmem": NTMEM, "chunk": CHUNK, "color": COLOR, "fusion": FUSION, "gather": GATHER, "slices": SLICES, "reduce": REDUCE, "scatter": SCATTER_CUTS, "stagger": STAGGER, "iters": ITERS, "wgsize": WGSIZE}


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
        core = os.path.join(tmp, "core.o")  # (lnh_build_variant lives there)
        r = subprocess.run([product_build.HIPCC] + product_build.FLAGS + [f'-DLNH_VARIANT_TAG="{name}"', "-c",
                                                                          os.path.join(dst, "core.hip"), "-o", core],
                           capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(f"variant {name}: hipcc failed (core.hip)\n{r.stderr[-3000:]}")
        product_build.build(verbose=False)
        objs = [o for o in sorted(glob.glob(os.path.join(PKG, "lib", "obj", "*.o")))
                if os.path.basename(o) not in (os.path.basename(obj), "core.o")] + [obj, core]
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
