"""s3gaussian_amd -- MI355X-native hot path of S3Gaussian (rasterizer, simple-knn, hexplane+deformation MLP).

Only what the hot path needs lives here: `csrc/` (gfx950 HIP kernels + the C ABI of include/*.h), the ctypes
binding, and the host-side mirrors of the reference's operator interfaces.  See DESIGN.md.
"""
__version__ = "0.1.0"
