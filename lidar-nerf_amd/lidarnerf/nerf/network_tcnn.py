"""NeRFNetwork, tcnn flavour — API / module names / state-dict keys of lidarnerf/nerf/network_tcnn.py:10-219, the class
`main_lidarnerf.py` instantiates for `-L` / `--tcnn` (main_lidarnerf.py:280-324), built on `lidarnerf.tcnn_compat`
instead of tinycudann.  Differences from `network.NeRFNetwork` that change numbers: positions and directions are mapped
to [0,1] before the encoders, the LiDAR direction encoding is tcnn's 72-wide Frequency (no raw input, pi-scaled), and
every module owns a flat `params` vector.  PARITY UNPINNED against real tiny-cuda-nn (see tcnn_compat.py).
"""
import torch

from .. import tcnn_compat as tcnn
from ..activation import trunc_exp
from . import fused
from .renderer import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="HashGrid", desired_resolution=2048, log2_hashmap_size=19,
                 encoding_dir="SphericalHarmonics", n_features_per_level=2, num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, out_color_dim=3, out_lidar_color_dim=2,
                 bound=1, fused_lidar=True, **kwargs):
        super().__init__(bound, **kwargs)
        self.fused_lidar = fused_lidar
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.desired_resolution, self.log2_hashmap_size = desired_resolution, log2_hashmap_size
        self.out_color_dim, self.out_lidar_color_dim = out_color_dim, out_lidar_color_dim
        self.n_features_per_level = n_features_per_level
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color

        pls = tcnn.per_level_scale(desired_resolution, bound)
        self.encoder = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16,
                                         "n_features_per_level": n_features_per_level,
                                         "log2_hashmap_size": log2_hashmap_size, "base_resolution": 16,
                                         "per_level_scale": pls})
        mlp = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None"}
        self.sigma_net = tcnn.Network(self.encoder.n_output_dims, 1 + geo_feat_dim,
                                      dict(mlp, n_neurons=hidden_dim, n_hidden_layers=num_layers - 1))
        self.encoder_dir = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})
        self.encoder_lidar_dir = tcnn.Encoding(3, {"otype": "Frequency", "degree": 12})
        self.in_dim_color = self.encoder_dir.n_output_dims + geo_feat_dim
        self.color_net = tcnn.Network(self.in_dim_color, out_color_dim,
                                      dict(mlp, n_neurons=hidden_dim_color, n_hidden_layers=num_layers_color - 1))
        self.in_dim_lidar_color = self.encoder_lidar_dir.n_output_dims + geo_feat_dim
        self.lidar_color_net = tcnn.Network(self.in_dim_lidar_color, out_lidar_color_dim,
                                            dict(mlp, n_neurons=hidden_dim_color,
                                                 n_hidden_layers=num_layers_color - 1))

    # ---- what the fused LiDAR chain (nerf/fused.py) needs to know about this field
    def fused_spec(self):
        g = self.encoder.impl
        s0, s1 = self.sigma_net.matrices()
        c = self.lidar_color_net.matrices()
        kd = self.encoder_lidar_dir.n_output_dims
        return fused.FieldSpec(grid=g, table=g.embeddings, ws0=s0, ws1=s1,
                               wc0=c[0][:, :kd + self.geo_feat_dim], wc1=c[1] if len(c) == 3 else None,
                               wc2=c[-1][:self.out_lidar_color_dim], n_dir=kd,
                               dir_features=lambda d: self.encoder_lidar_dir.frequency((d + 1) / 2),
                               n_color_mats=len(c), table_param=g.params)

    def run(self, rays_o, rays_d, cal_lidar_color=False, num_steps=128, upsample_steps=128, bg_color=None,
            perturb=False, **kwargs):
        if (self.fused_lidar and rays_o.is_cuda and torch.is_autocast_enabled()
                and fused.supported(self, cal_lidar_color, num_steps, upsample_steps)):
            self.out_dim = self.out_lidar_color_dim
            return fused.render_lidar(self, rays_o, rays_d, num_steps, upsample_steps, perturb,
                                      noise=kwargs.get("noise"), u=kwargs.get("u"))
        return super().run(rays_o, rays_d, cal_lidar_color=cal_lidar_color, num_steps=num_steps,
                           upsample_steps=upsample_steps, bg_color=bg_color, perturb=perturb, **kwargs)

    def forward(self, x, d):  # the reference leaves it empty (network_tcnn.py:134-135)
        pass

    def density(self, x):
        x = (x + self.bound) / (2 * self.bound)
        h = self.sigma_net(self.encoder(x))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, cal_lidar_color=False, mask=None, geo_feat=None, **kwargs):
        rgbs = None
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], self.out_dim, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        d = (d + 1) / 2
        enc, net = (self.encoder_lidar_dir, self.lidar_color_net) if cal_lidar_color \
            else (self.encoder_dir, self.color_net)
        h = torch.sigmoid(net(torch.cat([enc(d).to(geo_feat.dtype), geo_feat], dim=-1)))
        if mask is None:
            return h
        rgbs[mask] = h.to(rgbs.dtype)
        return rgbs

    def get_params(self, lr):
        groups = [self.encoder, self.sigma_net, self.encoder_dir, self.encoder_lidar_dir, self.color_net,
                  self.lidar_color_net]
        return [{"params": g.parameters(), "lr": lr} for g in groups]
