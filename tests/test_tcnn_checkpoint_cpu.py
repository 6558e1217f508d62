"""tcnn_compat.Encoding and checkpoints: its own flat `params` round-trips; a checkpoint written by REAL
tiny-cuda-nn — recognised by the parameter count tiny-cuda-nn's published HashGrid layout gives for the config — is refused
with an explanation instead of a bare size mismatch (the dense levels' geometry differs: INTEGRATION.md §A)."""
import pytest
import torch

from lidarnerf import tcnn_compat as T

CFG = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
       "per_level_scale": 1.4472692012786865}  # network_tcnn.py:40-57 with desired_resolution 2048 * bound


def test_own_params_round_trip_and_real_tcnn_checkpoint_is_refused_by_name():
    e = T.Encoding(3, CFG)
    mine, theirs = e.impl.params.numel(), T._tcnn_hashgrid_param_count(CFG, 3)
    assert mine != theirs and list(e.state_dict()) == ["params"]
    # levels 0-4 are dense here (resolutions 16, 24, 34, 49, 71): (res + 1)^3 rows against tiny-cuda-nn's res^3, both rounded up to 8
    assert mine - theirs == 2 * sum(-(-(r + 1) ** 3 // 8) * 8 - -(-r ** 3 // 8) * 8 for r in (16, 24, 34, 49, 71))
    e.load_state_dict({"params": torch.full((mine,), 0.5)})
    assert float(e.impl.params[0]) == 0.5
    with pytest.raises(RuntimeError, match="REAL tiny-cuda-nn"):
        e.load_state_dict({"params": torch.zeros(theirs)})
    with pytest.raises(RuntimeError, match="another encoding config"):
        e.load_state_dict({"params": torch.zeros(1000)})


def test_converted_tcnn_table_reads_the_same_value_at_every_vertex(monkeypatch):
    """convert_tcnn_hashgrid_params: every grid vertex of a dense level reads, through THIS package's index function (the
    oracle's restatement of gridencoder.cu:69-93), the row tiny-cuda-nn's index function (x + y res + z res^2, aliasing
    included) would have read; hashed levels are copied.  The tiny-cuda-nn side is a restatement (parity unpinned)."""
    import numpy as np
    from oracle import grid_ref
    levels = T._tcnn_levels(CFG, 3)
    rows_t = sum(r for _, r, _ in levels)
    src = torch.arange(rows_t * 2, dtype=torch.float32)  # feature f of tcnn row i holds 2 i + f (exact in fp32 below 2^24)
    out = T.convert_tcnn_hashgrid_params(src, CFG, 3).view(-1, 2)
    offs = grid_ref.make_offsets(3, 16, CFG["per_level_scale"], 16, 19)
    assert out.shape[0] == offs[-1]
    rng = np.random.default_rng(14)
    o_t = 0
    for l, (res, rows, hashed) in enumerate(levels):
        mine_rows = int(offs[l + 1] - offs[l])
        if hashed:
            assert mine_rows == rows and torch.equal(out[offs[l]:offs[l + 1]], src.view(-1, 2)[o_t:o_t + rows])
        else:
            v = rng.integers(0, res + 1, size=(4000, 3)).astype(np.uint32)
            v[:8] = [[0, 0, 0], [res, res, res], [res, 0, 0], [0, res, 0], [0, 0, res], [res, res, 0], [1, 2, 3], [res - 1, res, 1]]
            mine = grid_ref.grid_index(v, mine_rows, res).astype(np.int64) + int(offs[l])
            theirs = (v[:, 0].astype(np.int64) + v[:, 1] * res + v[:, 2].astype(np.int64) * res * res) % rows + o_t
            np.testing.assert_array_equal(out[mine, 0].numpy(), 2.0 * theirs)
            np.testing.assert_array_equal(out[mine, 1].numpy(), 2.0 * theirs + 1)
        o_t += rows
    assert sum(not h for _, _, h in levels) == 5
    # through load_state_dict: refused by default, converted with LNH_TCNN_CONVERT=1
    e = T.Encoding(3, CFG)
    with pytest.raises(RuntimeError, match="LNH_TCNN_CONVERT"):
        e.load_state_dict({"params": src})
    monkeypatch.setenv("LNH_TCNN_CONVERT", "1")
    with pytest.warns(UserWarning, match="converting"):
        e.load_state_dict({"params": src})
    assert torch.equal(e.impl.params.detach().view(-1, 2), out)
