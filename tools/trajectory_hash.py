#!/usr/bin/env python3
"""sha256 of the training state after N steps of the benchmark's step from seed 0 (tests/test_determinism_gpu.py's recipe):
run it on two boxes and compare — the training path has no order-dependent sum, so the digests agree.
    python tools/trajectory_hash.py [--steps 100] [--patch 2x8]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-nerf_amd"), os.path.join(ROOT, "tests")]
import torch
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--patch", default="1x1")
a = ap.parse_args()
import test_determinism_gpu as T
patch = tuple(int(v) for v in a.patch.split("x"))
for graph in (False, True):
    st = T._train(patch, a.steps, graph)
    h = hashlib.sha256()
    for t in st:
        h.update(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    print(f"patch {a.patch} steps {a.steps} graph {graph}: {h.hexdigest()[:32]}  {torch.cuda.get_device_name(0)}  {os.uname().nodename}", flush=True)
