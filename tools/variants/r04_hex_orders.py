# python tools/mkvariants.py tools/variants/r04_hex_orders.py   (round 4: walk orders of the HexPlane scatter, same-box A/B)
# finest_order = rounds 1-3: three orders (the finest level's cells), two levels per walk, two-entry footprint cache, 4 waves / SIMD.
# The tree: one order per (orientation, level), one level per walk, single-entry footprint, <= 64 VGPRs.
# (profiles/r04_hex_orders.txt, r04_hex_entries.txt: the intermediate steps -- per-level orders with the two-entry cache at 4 / 5 / 6
#  waves, segment lengths 128 ... 1024 -- were measured with earlier revisions of this file.)
_PL = ("#ifndef S3G_HEX_PER_LEVEL\n#define S3G_HEX_PER_LEVEL 1\n#endif", "#define S3G_HEX_PER_LEVEL 0")
_WV = ("#define S3G_HEX_SCATTER_WAVES 6", "#define S3G_HEX_SCATTER_WAVES 4")
VARIANTS = {
    "finest_order": ("hexplane.hip", [_PL, _WV]),
}
