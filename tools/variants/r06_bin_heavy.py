"""Binning kernels on the rasterizer-heavy workload: the rect size from which a whole wave walks a rect (tree: more than 32 tiles) and
the workgroup size (tree: 512).  bash tools/ab_kstats_heavy.sh "tree bh_big16 bh_big8 bh_t1024 bh_t256" bin_kernel"""
_B = "constexpr int BIG_RECT = 32;"
_T = "#define S3G_BIN_THREADS 512"
VARIANTS = {
    "bh_big16": ("raster_forward.hip", [(_B, _B.replace("32", "16"))]),
    "bh_big8": ("raster_forward.hip", [(_B, _B.replace("32", "8"))]),
    "bh_t1024": ("raster_forward.hip", [(_T, _T.replace("512", "1024"))]),
    "bh_t256": ("raster_forward.hip", [(_T, _T.replace("512", "256"))]),
}
