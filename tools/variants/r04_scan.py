"""bin_scan_kernel: parts the nb table rows of a tile are split into (one wave each); rounds 1-3 = one thread per tile over all rows, 24 us."""
OLD = "#define S3G_SCAN_PARTS 8"
VARIANTS = {f"scan_p{n}": ("raster_forward.hip", [(OLD, f"#define S3G_SCAN_PARTS {n}")]) for n in (2, 4, 16)}
