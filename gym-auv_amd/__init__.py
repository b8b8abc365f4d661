"""MI355X-native vectorised AUV simulation: fp64 vessel dynamics, LiDAR
ray-cast and collision/risk bookkeeping behind a gym-style reset()/step()."""
__version__ = "0.1.0"
