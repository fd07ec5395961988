"""Per-tile sort, bucket pass (round 6): where the bucket pass should start and how many bins it should have.
    python tools/mkvariants.py tools/variants/r06_tile_sort.py; bash tools/ab_kstats.sh "tree ts_min512 ts_min256 ts_bins4096 ts_bins1024" (grep sort_tiles)"""
_C = "constexpr int SORT_BIN_BITS = 11;\nconstexpr uint32_t SORT_BINS = 1u << SORT_BIN_BITS, BUCKET_MIN = 1024,"
VARIANTS = {
    "ts_min512": ("raster_forward.hip", [(_C, _C.replace("BUCKET_MIN = 1024", "BUCKET_MIN = 512"))]),
    "ts_min256": ("raster_forward.hip", [(_C, _C.replace("BUCKET_MIN = 1024", "BUCKET_MIN = 256"))]),
    "ts_bins4096": ("raster_forward.hip", [(_C, _C.replace("SORT_BIN_BITS = 11", "SORT_BIN_BITS = 12"))]),
    "ts_bins1024": ("raster_forward.hip", [(_C, _C.replace("SORT_BIN_BITS = 11", "SORT_BIN_BITS = 10"))]),
    "ts_min512_bins1024": ("raster_forward.hip", [(_C, _C.replace("BUCKET_MIN = 1024", "BUCKET_MIN = 512").replace("SORT_BIN_BITS = 11", "SORT_BIN_BITS = 10"))]),
}
