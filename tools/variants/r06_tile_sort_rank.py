"""Per-tile sort: lists of at most RANK_DIRECT keys ranked by counting, one thread per key (tree: 256); 0 = the network for every short
list; 128.  (Two keys per thread up to 512 keys was measured too -- slower than the network: profiles/r06_tile_sort_ab.txt.)"""
_R = "#define S3G_SORT_RANK_DIRECT 256"
VARIANTS = {
    "tr_0": ("raster_forward.hip", [(_R, _R.replace("256", "0"))]),
    "tr_128": ("raster_forward.hip", [(_R, _R.replace("256", "128"))]),
}
