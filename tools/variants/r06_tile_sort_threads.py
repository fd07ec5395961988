"""Per-tile sort: workgroup sizes of the two long-list launches (tree: 512 for (512, 4096] keys, 1024 for the longer lists)."""
_M, _L = "#define S3G_SORT_THREADS_MID 512", "#define S3G_SORT_THREADS_LONG 1024"
VARIANTS = {
    "tt_256_256": ("raster_forward.hip", [(_M, _M.replace("512", "256")), (_L, _L.replace("1024", "256"))]),
    "tt_256_1024": ("raster_forward.hip", [(_M, _M.replace("512", "256"))]),
    "tt_1024_1024": ("raster_forward.hip", [(_M, _M.replace("512", "1024"))]),
    "tt_512_512": ("raster_forward.hip", [(_L, _L.replace("1024", "512"))]),
}
