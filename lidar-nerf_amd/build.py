#!/usr/bin/env python3
"""Build liblidarnerf_hip.so (gfx950 only) with hipcc — in-tree, no JIT cache.

    python lidar-nerf_amd/build.py [--force] [-j N]

Every csrc/*.hip is compiled to an object (in parallel) and linked into lib/liblidarnerf_hip.so.  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import argparse
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# Developer hook for A/B experiments (tools/): LNH_VARIANT=name builds lib/liblidarnerf_hip_<name>.so from objects in
# lib/obj_<name>/ with LNH_EXTRA_FLAGS appended; the product library and its objects are not touched.
_VARIANT = os.environ.get("LNH_VARIANT", "")
OBJ = os.path.join(HERE, "lib", "obj" + ("_" + _VARIANT if _VARIANT else ""))
LIB = os.path.join(HERE, "lib", "liblidarnerf_hip" + ("_" + _VARIANT if _VARIANT else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: fused multiply-adds are written explicitly (fmaf) where the reference's compiler fuses them, so
# integer results derived from float math (cell indices, step counts) are reproducible against the CPU oracle.
# -amdgpu-mfma-vgpr-form: let v_mfma write architectural VGPRs.  By default hipcc puts EVERY MFMA destination into an
# AGPR and copies it back with v_accvgpr_read before the VALU can touch it (4 moves per MFMA); the tiny-MLP kernels apply
# an activation to every MFMA result, so those copies were ~35 % of their instruction stream.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


MFMA_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# hipcc (ROCm 7.2) crashes on the 2-hidden-matrix backward instantiations with that option: they keep the default
NO_VGPR_FORM = {"mlp_bwd_nhm2.hip", "mlp_bwd_nhm2_bf16.hip"}


def _newer(src, deps, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def _local_includes(src, seen=None):
    """Files of csrc/ a translation unit pulls in with #include "..." (transitively): the *_bf16.hip wrappers include the
    .hip file they re-compile, so a change there must rebuild them too."""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(src).read()
    except OSError:
        return seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        path = os.path.join(CSRC, name)
        if os.path.exists(path) and path not in seen:
            seen.add(path)
            _local_includes(path, seen)
    return seen


def _compile(src, force):
    out = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
    if _VARIANT and os.path.basename(src) not in os.environ.get("LNH_VARIANT_FILES", "grid.hip").split():
        return os.path.join(HERE, "lib", "obj", os.path.basename(out)), False  # unchanged files: the product's objects
    deps = (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")) +
            sorted(_local_includes(src)))
    if not force and not _newer(src, deps, out):
        return out, False
    extra = [] if os.path.basename(src) in NO_VGPR_FORM else MFMA_VGPR_FORM
    cmd = [HIPCC] + FLAGS + extra + os.environ.get("LNH_EXTRA_FLAGS", "").split() + ["-c", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return out, True


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for out, did in ex.map(lambda s: _compile(s, force), srcs):
            objs.append(out)
            rebuilt |= did
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] linked {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    elif verbose:
        print(f"[build] {LIB} is up to date")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    a = ap.parse_args()
    try:
        build(a.force, a.j)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
