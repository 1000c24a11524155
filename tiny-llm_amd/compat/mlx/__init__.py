"""Import-name facade: `mlx` on top of torch (see ../README.md).  Not MLX."""
__version__ = "0.32.0+torch-facade"
