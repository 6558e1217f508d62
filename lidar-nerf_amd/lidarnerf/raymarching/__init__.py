from .raymarching import (composite_rays, composite_rays_train, composite_rays_train_lidar, march_capacity, march_rays, march_rays_train, morton3D, morton3D_invert,  # noqa: F401
                          near_far_from_aabb, packbits, sph_from_ray)
