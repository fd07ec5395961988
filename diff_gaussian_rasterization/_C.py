"""Drop-in for the native module `diff_gaussian_rasterization._C` (RAST/ext.cpp:15-19)."""
from s3gaussian_amd.raster_C import mark_visible, rasterize_gaussians, rasterize_gaussians_backward  # noqa: F401
