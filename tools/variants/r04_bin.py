"""Binning walks (bin_kernel<false/true>): threads per workgroup and the next step's loads requested a step ahead."""
T = "#define S3G_BIN_THREADS 512"
PF = "#define S3G_BIN_PREFETCH 1"
VARIANTS = {
    "bin_256_nopf": ("raster_forward.hip", [(T, "#define S3G_BIN_THREADS 256"), (PF, "#define S3G_BIN_PREFETCH 0")]),   # rounds 1-3
    "bin_256": ("raster_forward.hip", [(T, "#define S3G_BIN_THREADS 256")]),
    "bin_512_nopf": ("raster_forward.hip", [(PF, "#define S3G_BIN_PREFETCH 0")]),
    "bin_1024": ("raster_forward.hip", [(T, "#define S3G_BIN_THREADS 1024")]),
    "bin_1024_nopf": ("raster_forward.hip", [(T, "#define S3G_BIN_THREADS 1024"), (PF, "#define S3G_BIN_PREFETCH 0")]),
}
