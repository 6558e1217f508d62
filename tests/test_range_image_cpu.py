"""lidarnerf/dataset/range_image.py against G10: what the reference's OWN KITTI360Dataset / NeRFMVLDataset made of the same
files (tests/golden/make_g10_dataset.py; kitti360_dataset.py:72-121, nerfmvl_dataset.py:55-113).  Bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from lidarnerf.dataset import range_image as ri

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_dataset.npz"))


def _write(root, raw, poses, sub, ext):
    os.makedirs(os.path.join(root, sub))
    frames = []
    for i in range(raw.shape[0]):
        p = os.path.join(sub, f"{i:05d}{ext}")
        if ext == ".npy":
            np.save(os.path.join(root, p), raw[i])
        else:
            np.savez(os.path.join(root, p), data=raw[i])
        frames.append({"lidar2world": poses[i].tolist(), "lidar_file_path": p})
    with open(os.path.join(root, "transforms.json"), "w") as f:
        json.dump({"h_lidar": raw.shape[1], "w_lidar": raw.shape[2], "frames": frames}, f)


def test_kitti360_sequence_matches_the_reference_dataset_class(tmp_path):
    _write(str(tmp_path), G["k_raw"], G["k_poses_raw"], "train", ".npy")
    seq = ri.load_sequence(str(tmp_path), "transforms.json", "kitti360", float(G["k_scale"]), G["k_offset"].tolist())
    assert seq["images_lidar"].dtype == torch.float32 and (seq["H_lidar"], seq["W_lidar"]) == G["k_raw"].shape[1:3]
    np.testing.assert_array_equal(seq["images_lidar"].numpy(), G["k_images"])
    np.testing.assert_array_equal(seq["poses_lidar"].numpy(), G["k_poses"])
    # the preload rule: images as fp16 on the device
    seq16 = ri.load_sequence(str(tmp_path), "transforms.json", "kitti360", float(G["k_scale"]), G["k_offset"].tolist(),
                             device="cpu", fp16=True)
    assert seq16["images_lidar"].dtype == torch.float16
    np.testing.assert_array_equal(seq16["images_lidar"].numpy(), G["k_images_fp16"])
    # one frame, kind from the extension
    one = ri.load_range_image(str(tmp_path / "train" / "00001.npy"), float(G["k_scale"]))
    np.testing.assert_array_equal(one.numpy(), G["k_images"][1])
    assert set(np.unique(G["k_images"][..., 0])) == {0.0, 1.0}


def test_nerfmvl_sequence_matches_the_reference_dataset_class(tmp_path):
    _write(str(tmp_path), G["m_raw"], G["m_poses_raw"], "car", ".npz")
    seq = ri.load_sequence(str(tmp_path), "transforms.json", "nerfmvl", float(G["m_scale"]), G["m_offset"])
    np.testing.assert_array_equal(seq["images_lidar"].numpy(), G["m_images"])
    np.testing.assert_array_equal(seq["poses_lidar"].numpy(), G["m_poses"])
    one = ri.load_range_image(str(tmp_path / "car" / "00000.npz"), float(G["m_scale"]))
    np.testing.assert_array_equal(one.numpy(), G["m_images"][0])
    assert (G["m_images"][0, 0, :3, 0] < 0).all()  # the NeRF-MVL mask keeps non-positive depths (nerfmvl_dataset.py:67-69)


def test_malformed_files_are_refused_by_name(tmp_path):
    np.save(tmp_path / "flat.npy", np.zeros((4, 8), np.float32))
    with pytest.raises(ValueError, match=r"\[H, W, 3\]"):
        ri.load_range_image(str(tmp_path / "flat.npy"))
    np.savez(tmp_path / "other.npz", points=np.zeros((4, 8, 3), np.float32))
    with pytest.raises(KeyError, match="data"):
        ri.load_range_image(str(tmp_path / "other.npz"))
    with pytest.raises(ValueError, match="npy"):
        ri.load_range_image(str(tmp_path / "x.bin"))
