"""get_encoder — encoder factory with the reference's names and keyword arguments (lidarnerf/encoding.py:50-120)."""
import torch
import torch.nn as nn


class FreqEncoder(nn.Module):
    """Pure-torch positional encoding (the reference keeps this twin next to the kernel version,
    lidarnerf/encoding.py:6-47); same output layout as freqencoder.FreqEncoder."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True,
                 periodic_fns=(torch.sin, torch.cos)):
        super().__init__()
        self.input_dim, self.include_input, self.periodic_fns = input_dim, include_input, periodic_fns
        self.output_dim = (input_dim if include_input else 0) + input_dim * N_freqs * len(periodic_fns)
        bands = 2.0 ** torch.linspace(0.0, max_freq_log2, N_freqs) if log_sampling \
            else torch.linspace(2.0 ** 0.0, 2.0 ** max_freq_log2, N_freqs)
        self.freq_bands = bands.numpy().tolist()

    def forward(self, input, **kwargs):
        parts = [input] if self.include_input else []
        for freq in self.freq_bands:
            parts += [fn(input * freq) for fn in self.periodic_fns]
        return torch.cat(parts, dim=-1)


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency":
        from .freqencoder import FreqEncoder as HipFreqEncoder
        enc = HipFreqEncoder(input_dim=input_dim, degree=multires)
    elif encoding == "sphere_harmonics":
        from .shencoder import SHEncoder
        enc = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        from .gridencoder import GridEncoder
        enc = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim,
                          base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                          desired_resolution=desired_resolution,
                          gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners)
    else:
        raise NotImplementedError(
            "Unknown encoding mode, choose from [None, frequency, sphere_harmonics, hashgrid, tiledgrid]")
    return enc, enc.output_dim
