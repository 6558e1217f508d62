"""lidarnerf — MI355X-native drop-in for the LiDAR-NeRF train/render hot path."""
__version__ = "0.1.0"
