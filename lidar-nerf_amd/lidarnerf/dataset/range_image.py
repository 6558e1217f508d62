"""Range-image files -> the tensors the training / evaluation path consumes.

The reference reads them inside its dataset classes (lidarnerf/dataset/kitti360_dataset.py:72-112 — `.npy`, one [H, W, 3]
float array per frame: channel 1 = intensity, channel 2 = depth in metres, 0 = no return — and
lidarnerf/dataset/nerfmvl_dataset.py:55-113 — `.npz` with the same array under "data").  Those classes keep working unchanged
(they resolve from the reference checkout behind this package); this module is the same I/O as plain
functions for callers that feed the HIP path directly (bench / tools / a data loader of their own):

    images_lidar[..., 0] = ray-drop mask    KITTI-360: 0 where depth == 0, else 1   (kitti360_dataset.py:82-84)
                                            NeRF-MVL : depth where depth <= 0, else 1 (nerfmvl_dataset.py:67-69)
    images_lidar[..., 1] = intensity
    images_lidar[..., 2] = depth * scale
    poses_lidar[:, :3, 3] = (t - offset) * scale                                      (kitti360_dataset.py:95-98)

Pinned to the reference's own classes by tests/golden/g10_dataset.npz (tests/golden/make_g10_dataset.py runs them on a
small synthetic sequence)."""
import json
import os

import numpy as np
import torch

KINDS = ("kitti360", "nerfmvl")


def read_range_image_file(path):
    """The raw [H, W, 3] array of one frame: `.npy`, or `.npz` with the array under "data"."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        with np.load(path) as z:
            if "data" not in z.files:
                raise KeyError(f"{path}: a NeRF-MVL range image is expected under the key 'data' (found {z.files})")
            pc = z["data"]
    elif ext == ".npy":
        pc = np.load(path)
    else:
        raise ValueError(f"{path}: range images are .npy (KITTI-360) or .npz (NeRF-MVL) files")
    if pc.ndim != 3 or pc.shape[2] != 3:
        raise ValueError(f"{path}: expected a [H, W, 3] array (unused, intensity, depth), got {pc.shape}")
    return pc


def range_image_channels(pc, scale=1.0, kind="kitti360"):
    """[H, W, 3] raw array -> [H, W, 3] (ray-drop mask, intensity, depth * scale), in the array's own float type like the
    reference's np.concatenate (the stack of frames is cast to float32 afterwards)."""
    if kind not in KINDS:
        raise ValueError(f"kind must be one of {KINDS}")
    depth = pc[:, :, 2]
    if kind == "kitti360":
        drop = np.where(depth == 0.0, 0.0, 1.0)
    else:
        drop = depth.copy()
        drop[drop > 0] = 1.0
    return np.concatenate([drop[:, :, None], pc[:, :, 1, None], pc[:, :, 2, None] * scale], axis=-1)


def load_range_image(path, scale=1.0, kind=None, device=None, dtype=torch.float32):
    """One frame as a tensor [H, W, 3] on `device`.  kind: None = by extension (.npy -> kitti360, .npz -> nerfmvl)."""
    if kind is None:
        kind = "nerfmvl" if path.lower().endswith(".npz") else "kitti360"
    img = torch.from_numpy(np.ascontiguousarray(range_image_channels(read_range_image_file(path), scale, kind))).float()
    return img.to(device=device, dtype=dtype) if device is not None or dtype != torch.float32 else img


def scale_poses(poses, offset, scale):
    """lidar2world matrices [N, 4, 4] float32 -> the unit cube's frame: translation (t - offset) * scale."""
    poses = np.array(poses, dtype=np.float32, copy=True)
    poses[:, :3, -1] = (poses[:, :3, -1] - np.asarray(offset)) * scale
    return poses


def load_sequence(root_path, transforms_file, kind="kitti360", scale=1.0, offset=(0.0, 0.0, 0.0), device=None, fp16=True,
                  preload=True):
    """The frames of one transforms_*.json (the reference's nerf-style index: h_lidar, w_lidar, frames[*].lidar2world and
    .lidar_file_path) as {poses_lidar [N,4,4] f32, images_lidar [N,H,W,3], H_lidar, W_lidar}.  preload (with a device):
    both on the device, the images as fp16 if `fp16` — the reference's preload rule (kitti360_dataset.py:112-121)."""
    with open(os.path.join(root_path, transforms_file)) as f:
        tr = json.load(f)
    H, W = int(tr["h_lidar"]), int(tr["w_lidar"])
    poses, images = [], []
    for fr in tr["frames"]:
        poses.append(np.array(fr["lidar2world"], dtype=np.float32))
        pc = read_range_image_file(os.path.join(root_path, fr["lidar_file_path"]))
        if pc.shape[:2] != (H, W):
            raise ValueError(f"{fr['lidar_file_path']}: {pc.shape[:2]} but the index says {H} x {W}")
        images.append(range_image_channels(pc, scale, kind))
    out = dict(H_lidar=H, W_lidar=W,
               poses_lidar=torch.from_numpy(scale_poses(np.stack(poses, 0), offset, scale)),
               images_lidar=torch.from_numpy(np.stack(images, 0)).float())
    if preload and device is not None:
        out["poses_lidar"] = out["poses_lidar"].to(device)
        out["images_lidar"] = out["images_lidar"].to(torch.half if fp16 else torch.float).to(device)
    return out
