"""Drop-in for the reference's `diff_gaussian_rasterization` package (RAST/setup.py:18-22): same import surface,
backed by the MI355X-native HIP library (s3gaussian_amd/lib/libs3g.so)."""
from s3gaussian_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians,  # noqa: F401
                                       rasterize_gaussians)
from . import _C  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
