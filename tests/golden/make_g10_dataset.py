#!/usr/bin/env python3
"""G10 — the reference's OWN dataset classes on a small synthetic sequence (CPU, build container only: needs /root/reference):
KITTI360Dataset (lidarnerf/dataset/kitti360_dataset.py:13-123, `.npy` frames) and NeRFMVLDataset
(lidarnerf/dataset/nerfmvl_dataset.py:13-114, `.npz` frames).  Written: the raw frames and poses that were put on disk, and
what the classes made of them (images_lidar, poses_lidar) — the pin of lidarnerf/dataset/range_image.py.

    python tests/golden/make_g10_dataset.py     # writes tests/golden/g10_dataset.npz
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))
sys.path.insert(0, REF)

from lidarnerf.dataset.kitti360_dataset import KITTI360Dataset  # noqa: E402
from lidarnerf.dataset.nerfmvl_dataset import NeRFMVLDataset  # noqa: E402

H, W, N = 6, 20, 3
SCALE_K, OFFSET_K = 0.010784853507573345, [1.5, -2.25, 0.5]
SCALE_M = 0.25


def frames(rng, negative_depths):
    pcs = []
    for _ in range(N):
        pc = np.zeros((H, W, 3), dtype=np.float32)
        pc[:, :, 0] = rng.random((H, W))  # unused channel
        pc[:, :, 1] = rng.random((H, W))
        pc[:, :, 2] = rng.random((H, W)) * 70 + 1
        pc[:, :, 2][rng.random((H, W)) < 0.2] = 0.0  # no return
        if negative_depths:
            pc[0, :3, 2] = [-1.0, -0.5, -3.0]  # NeRF-MVL's mask keeps non-positive depths as they are
        pcs.append(pc)
    poses = []
    for i in range(N):
        a = 0.3 * i
        p = np.eye(4, dtype=np.float32)
        p[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        p[:3, 3] = [10.0 * i + 1.5, -2.0 - i, 0.5 + 0.1 * i]
        poses.append(p)
    return np.stack(pcs), np.stack(poses)


def main():
    rng = np.random.default_rng(10)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        pcs, poses = frames(rng, False)
        os.makedirs(os.path.join(root, "train"))
        fr = []
        for i in range(N):
            np.save(os.path.join(root, "train", f"{i:05d}.npy"), pcs[i])
            fr.append({"lidar2world": poses[i].tolist(), "lidar_file_path": f"train/{i:05d}.npy"})
        with open(os.path.join(root, "transforms_1908_train.json"), "w") as f:
            json.dump({"h_lidar": H, "w_lidar": W, "frames": fr}, f)
        ds = KITTI360Dataset(device="cpu", split="train", root_path=root, sequence_id="1908", preload=False, scale=SCALE_K,
                             offset=OFFSET_K, fp16=False)
        out.update(k_raw=pcs, k_poses_raw=poses, k_scale=np.float64(SCALE_K), k_offset=np.array(OFFSET_K),
                   k_images=ds.images_lidar.numpy(), k_poses=ds.poses_lidar.numpy())
        ds16 = KITTI360Dataset(device="cpu", split="train", root_path=root, sequence_id="1908", preload=True, scale=SCALE_K,
                               offset=OFFSET_K, fp16=True)
        out.update(k_images_fp16=ds16.images_lidar.numpy())

    with tempfile.TemporaryDirectory() as root:
        pcs, poses = frames(rng, True)
        os.makedirs(os.path.join(root, "car"))
        fr = []
        for i in range(N):
            np.savez(os.path.join(root, "car", f"{i:05d}.npz"), data=pcs[i])
            fr.append({"lidar2world": poses[i].tolist(), "lidar_file_path": f"car/{i:05d}.npz"})
        with open(os.path.join(root, "transforms_car_train.json"), "w") as f:
            json.dump({"h_lidar": H, "w_lidar": W, "frames": fr}, f)
        obb = (rng.random((8, 3)) * 4 - 2).astype(np.float64)
        np.save(os.path.join(root, "dataset_bbox_7k.npy"), {"car": obb}, allow_pickle=True)
        ds = NeRFMVLDataset(device="cpu", split="train", root_path=root, sequence_id="car", preload=False, scale=SCALE_M,
                            fp16=False)
        out.update(m_raw=pcs, m_poses_raw=poses, m_scale=np.float64(SCALE_M), m_offset=np.asarray(ds.offset),
                   m_images=ds.images_lidar.numpy(), m_poses=ds.poses_lidar.numpy())
    path = os.path.join(OUT, "g10_dataset.npz")
    np.savez_compressed(path, **out)
    print(path, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
