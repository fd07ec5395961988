"""Per-tile sort, bucket pass (round 6): where the bucket pass should start and how many bins it should have (tree: from 512 keys on,
2048 bins; profiles/r06_tile_sort_ab.txt was measured when the tree still started it at 1024 keys).
    python tools/mkvariants.py tools/variants/r06_tile_sort.py; bash tools/ab_kstats_heavy.sh "tree ts_min1024 ts_min256 ts_bins4096 ts_bins1024" """
_C = "constexpr int SORT_BIN_BITS = 11;\nconstexpr uint32_t SORT_BINS = 1u << SORT_BIN_BITS, BUCKET_MIN = 512,"
VARIANTS = {
    "ts_min1024": ("raster_forward.hip", [(_C, _C.replace("BUCKET_MIN = 512", "BUCKET_MIN = 1024"))]),
    "ts_min256": ("raster_forward.hip", [(_C, _C.replace("BUCKET_MIN = 512", "BUCKET_MIN = 256"))]),
    "ts_bins4096": ("raster_forward.hip", [(_C, _C.replace("SORT_BIN_BITS = 11", "SORT_BIN_BITS = 12"))]),
    "ts_bins1024": ("raster_forward.hip", [(_C, _C.replace("SORT_BIN_BITS = 11", "SORT_BIN_BITS = 10"))]),
}
