"""geometry_backward_kernel: tiles of a Gaussian's rect whose records are in flight at a time (1 = the one-at-a-time walk of rounds 1-3)."""
OLD = "#define S3G_GATHER_BATCH 3 "
VARIANTS = {f"gather_b{b}": ("raster_backward.hip", [(OLD, f"#define S3G_GATHER_BATCH {b} ")]) for b in (1, 2, 4, 6, 8)}
