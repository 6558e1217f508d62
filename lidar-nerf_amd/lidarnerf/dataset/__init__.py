"""Ray generation of the LiDAR path (rays.get_lidar_rays)."""

# The reference checkout may sit BEHIND this package on sys.path: modules this package does not
# provide (nerf/utils.py = Trainer, the dataset classes, loss.py, ...) then resolve from there, everything it does
# provide shadows the reference's.  Nothing of the reference is copied or imported by this package itself.
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
