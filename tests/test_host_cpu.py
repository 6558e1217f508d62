"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the Python mirror of
the reference API keeps its shapes / names, the loss matches the restated Trainer.train_step arithmetic, and the
product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lidarnerf_hip.h")).read()
    return sorted(set(re.findall(r"LNH_API\s+[\w\s\*]+?\b(lnh_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lidarnerf import _hip
    names = _declared_symbols()
    assert len(names) >= 30 and "lnh_grid_encode_forward" in names and "lnh_lidar_color_backward" in names
    path = _hip.lib_path()
    assert os.path.exists(path), "build with `python lidar-nerf_amd/build.py` (done by __graft_entry__.build())"
    lib = ctypes.CDLL(path)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lidarnerf_hip.h but not exported"
    assert set(_hip.EXPORTS) <= set(names), set(_hip.EXPORTS) - set(names)
    lib.lnh_arch.restype = ctypes.c_char_p
    assert lib.lnh_arch() == b"gfx950"


def test_argument_errors_are_reported_without_a_gpu():
    """Validation happens before any launch, so the error contract is testable on the CPU."""
    from lidarnerf import _hip
    L = _hip.lib()
    off = np.array([0, 8, 16], dtype=np.int32)
    rc = L.lnh_grid_encode_forward(1, 1, off.ctypes.data, 1, 4, 3, 3, 2, 1.0, 16, None, 0, 0, 0, 0, None)
    assert rc == -2 and b"C must be 1, 2, 4, or 8" in L.lnh_last_error()
    rc = L.lnh_mlp_forward(1, 1, 16, 20, 16, 64, 0, 0, 6, None, 1, None)
    assert rc == -2 and b"input_dim should be 16" in L.lnh_last_error()
    rc = L.lnh_sh_encode_forward(1, 1, 4, 3, 9, None, None)
    assert rc == -2 and b"degree" in L.lnh_last_error()
    assert L.lnh_grid_backward_workspace_size(off.ctypes.data, 1000, 3, 2, 2, 1.0, 16, 0, 0, 1) > 0
    assert L.lnh_grid_backward_workspace_size(off.ctypes.data, 1000, 2, 2, 2, 1.0, 16, 0, 0, 1) == 0


def test_level_offsets_match_oracle_and_survey():
    from lidarnerf.gridencoder.grid import GridEncoder, level_offsets
    from oracle import grid_ref
    for res, log2h, D in ((32768, 19, 3), (2048, 19, 3), (2048, 19, 2), (512, 14, 3)):
        pls = grid_ref.per_level_scale(res, 16, 16)
        np.testing.assert_array_equal(level_offsets(D, 16, pls, 16, log2h, False),
                                      grid_ref.make_offsets(D, 16, pls, 16, log2h))
    enc = GridEncoder(desired_resolution=32768)
    assert enc.embeddings.shape == (6837544, 2) and enc.output_dim == 32 and enc.offsets.dtype == torch.int32
    assert abs(enc.per_level_scale - 1.662476) < 1e-6
    assert float(enc.embeddings.abs().max()) <= 1e-4


def test_module_api_and_state_dict_layout():
    from lidarnerf.encoding import get_encoder
    from lidarnerf.ffmlp import FFMLP
    from lidarnerf.nerf.network import NeRFNetwork
    enc, dim = get_encoder("frequency", multires=12)
    assert dim == 75 and enc.degree == 12
    enc, dim = get_encoder("sphere_harmonics")
    assert dim == 16
    with pytest.raises(NotImplementedError):
        get_encoder("ash")
    m = FFMLP(32, 3, 64, 2)
    assert m.weights.numel() == 64 * (32 + 64 + 16) and m.padded_output_dim == 16
    torch.manual_seed(42)
    w = torch.empty(m.num_parameters).uniform_(-np.sqrt(3 / 64), np.sqrt(3 / 64))
    assert torch.equal(m.weights.data, w)  # seed-42 initialisation of the reference (ffmlp.py:242-245)
    # every width of the reference and deeper nets construct (fused kernels exist for all of them) ...
    for hidden, layers in ((16, 2), (128, 2), (256, 3), (64, 5)):
        assert FFMLP(32, 3, hidden, layers).weights.numel() == hidden * (32 + hidden * (layers - 1) + 16)
    # ... what has no kernel constructs like the reference's does (any input_dim % 16 == 0, ffmlp.py:202-216) and takes the
    # library-GEMM chain with ONE warning; strict_fused=True refuses it like the C ABI refuses it
    import warnings
    FFMLP._warned_gemm_chain = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for in_dim, layers in ((256, 2), (32, 17)):
            assert FFMLP(in_dim, 3, 64, layers).gemm_chain
    assert len([w for w in rec if "no fused MFMA kernel" in str(w.message)]) == 1
    for in_dim, layers in ((256, 2), (32, 17)):
        with pytest.raises(RuntimeError, match="no fused MFMA kernel"):
            FFMLP(in_dim, 3, 64, layers, strict_fused=True)
        assert FFMLP(in_dim, 3, 64, layers, gemm_chain=True).gemm_chain
    net = NeRFNetwork(encoding="hashgrid", desired_resolution=2048, bound=1, min_near_lidar=0.01)
    keys = set(net.state_dict().keys())
    assert {"encoder.embeddings", "encoder.offsets", "sigma_net.0.weight", "sigma_net.1.weight",
            "lidar_color_net.0.weight", "lidar_color_net.2.weight", "color_net.0.weight", "aabb_train",
            "aabb_infer"} <= keys
    assert net.lidar_color_net[0].weight.shape == (64, 90) and net.sigma_net[1].weight.shape == (16, 64)
    assert len(net.get_params(1e-2)) == 6 and net.cascade == 1


def test_no_cpu_fallback():
    from lidarnerf.gridencoder import GridEncoder
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    enc = GridEncoder(desired_resolution=512, log2_hashmap_size=12)
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        enc(torch.rand(8, 3))


def test_lidar_loss_matches_restated_train_step():
    from lidarnerf.nerf.train_step import lidar_loss, patch_gradient_loss
    from oracle import render_ref
    g = torch.Generator().manual_seed(0)
    depth, image = torch.rand(1, 64, generator=g), torch.rand(1, 64, 2, generator=g)
    gt = torch.rand(1, 64, 3, generator=g)
    gt[..., 0] = (gt[..., 0] > 0.3).float()
    loss, pd, gd = lidar_loss({"depth_lidar": depth, "image_lidar": image}, gt)
    want = render_ref.lidar_loss(depth[0], image[0], gt[0])
    torch.testing.assert_close(loss, want)
    pl = patch_gradient_loss(pd, gd, gt[..., 0], 2, 8, 0.01)
    want_p = render_ref.patch_grad_loss(depth[0], gt[0], 2, 8, 0.01)
    torch.testing.assert_close(pl, want_p)


def test_tcnn_facade_api_and_state_dict_layout():
    """network_tcnn.NeRFNetwork (the class `-L` selects, network_tcnn.py:10-219) without tinycudann: constructor
    arguments, module names, tcnn's one-flat-`params`-per-module state dict, widths, GPU-only modules."""
    from lidarnerf import tcnn_compat
    from lidarnerf.nerf.network_tcnn import NeRFNetwork
    net = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, log2_hashmap_size=19, n_features_per_level=2,
                      bound=1, min_near_lidar=0.01, density_scale=1)
    sd = net.state_dict()
    assert {"encoder.params", "sigma_net.params", "encoder_dir.params", "encoder_lidar_dir.params", "color_net.params",
            "lidar_color_net.params", "aabb_train", "aabb_infer"} <= set(sd.keys())
    assert not any(".impl." in k for k in sd)
    assert sd["encoder.params"].numel() == 6837544 * 2 and sd["encoder_lidar_dir.params"].numel() == 0
    assert net.encoder.n_output_dims == 32 and net.encoder_dir.n_output_dims == 16
    assert net.encoder_lidar_dir.n_output_dims == 72 and net.in_dim_lidar_color == 87
    assert sd["sigma_net.params"].numel() == 64 * 32 + 16 * 64
    assert sd["lidar_color_net.params"].numel() == 64 * 96 + 64 * 64 + 16 * 64   # input padded 87 -> 96, output 2 -> 16
    assert sd["color_net.params"].numel() == 64 * 32 + 64 * 64 + 16 * 64
    assert abs(tcnn_compat.per_level_scale(32768, 1) - 1.6624757922855755) < 1e-12
    # round trip through a tcnn-shaped checkpoint
    net2 = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, bound=1, min_near_lidar=0.01)
    missing, unexpected = net2.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert torch.equal(net2.encoder.impl.params.data, net.encoder.impl.params.data)
    groups = [g for g in net.get_params(1e-2)]
    assert len(groups) == 6 and sum(len(list(g["params"])) for g in groups) == 4
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="CUDA"):
            net.sigma_net(torch.rand(4, 32))
        with pytest.raises(RuntimeError, match="CUDA"):
            net.encoder_lidar_dir(torch.rand(4, 3))
    # tcnn Frequency layout: out[d*2K + 2k + p] = sin(2^k pi x_d + p pi/2)
    x = torch.tensor([[0.25, 0.5, 0.8]])
    f = net.encoder_lidar_dir.frequency(x)[0]
    assert f.shape == (72,)
    for (dd, k, ph) in [(0, 0, 0), (0, 0, 1), (1, 3, 0), (2, 11, 1)]:
        want = np.sin(np.float32(2.0 ** k * np.pi) * np.float32(x[0, dd]) + ph * np.pi / 2)
        assert abs(float(f[dd * 24 + 2 * k + ph]) - want) < 2e-3 * max(1.0, 2.0 ** k * 1e-3)


def test_build_tracks_the_sources_a_wrapper_translation_unit_includes():
    """The *_bf16.hip translation units re-compile another .hip file through #include: the build must treat that file (and
    what it includes) as a dependency, or the bf16 kernels silently lag behind the fp16 ones."""
    import importlib.util
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lidar-nerf_amd")
    spec = importlib.util.spec_from_file_location("lnh_build", os.path.join(root, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    csrc = os.path.join(root, "csrc")
    for wrapper, included in (("lidar_color_bf16.hip", "lidar_color.hip"), ("mlp_bf16.hip", "mlp.hip"),
                              ("mlp_bwd_nhm0_bf16.hip", "mlp_bwd_nhm0.hip")):
        deps = {os.path.basename(p) for p in mod._local_includes(os.path.join(csrc, wrapper))}
        assert included in deps and "mlp_common.h" in deps, (wrapper, deps)
    assert "mlp_bwd.h" in {os.path.basename(p) for p in mod._local_includes(os.path.join(csrc, "mlp_bwd_nhm0_bf16.hip"))}


def test_header_is_plain_c():
    """include/lidarnerf_hip.h is the drop-in boundary: it must compile as C99 (no C++-isms, no torch / HIP types), so that a
    cgo / JNI / ctypes-generator binding can consume it as it is."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "h.c")
        with open(src, "w") as f:
            f.write('#include "lidarnerf_hip.h"\nint main(void) { return 0; }\n')
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                            "-fsyntax-only", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_signatures_match_the_header():
    """Every prototype of include/lidarnerf_hip.h against the argument list lidarnerf/_hip.py binds it with: same number of
    arguments, same class per position (pointer / uint32 / int32 / float / uint64 / double), stream last.  A drifted
    signature would not fail at load time — ctypes would pass garbage."""
    import ctypes as C
    from lidarnerf import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "lidarnerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = re.findall(r"LNH_API\s+int\s+(lnh_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    assert len(protos) >= 60

    def cls(arg):
        a = " ".join(arg.split())
        if "*" in a:
            return C.c_void_p
        ty = a.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"uint32_t": C.c_uint32, "int32_t": C.c_int, "int": C.c_int, "float": C.c_float, "uint64_t": C.c_uint64,
                "size_t": C.c_uint64, "double": C.c_double, "lnh_stream_t": C.c_void_p}[ty]

    checked = 0
    for name, args in protos:
        if name not in _hip._SIGS:
            continue
        want = [cls(a) for a in args.split(",")]
        assert want[-1] is C.c_void_p and "lnh_stream_t" in args.split(",")[-1], name
        got = list(_hip._SIGS[name]) + [C.c_void_p]
        assert [w for w in want] == got, (name, [w.__name__ for w in want], [g.__name__ for g in got])
        checked += 1
    assert checked == len(_hip._SIGS), (checked, len(_hip._SIGS), sorted(set(_hip._SIGS) - {n for n, _ in protos}))


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box without a GPU")
def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path or nothing: on a box without a GPU it exits non-zero with a message instead of timing
    a fallback."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())  # no result line
