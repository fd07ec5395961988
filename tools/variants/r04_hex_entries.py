# python tools/mkvariants.py tools/variants/r04_hex_entries.py   (round 4: footprint entries / occupancy / segment length of the per-level scatter)
_WV = ("#define S3G_HEX_SCATTER_WAVES 6", "#define S3G_HEX_SCATTER_WAVES {}")
_EN = ("#define S3G_HEX_FOOT_ENTRIES (S3G_HEX_PER_LEVEL ? 1 : 2)", "#define S3G_HEX_FOOT_ENTRIES {}")
_SEG = ("static inline int segment_length(int P) { return P >= 1000000 ? 256 : 128; }", "static inline int segment_length(int P) {{ return {}; }}")
VARIANTS = {
    "two_entries_w6": ("hexplane.hip", [(_EN[0], _EN[1].format(2))]),
    "one_entry_w8": ("hexplane.hip", [(_WV[0], _WV[1].format(8))]),
    "one_entry_w4": ("hexplane.hip", [(_WV[0], _WV[1].format(4))]),
    "one_entry_w8_seg128": ("hexplane.hip", [(_WV[0], _WV[1].format(8)), (_SEG[0], _SEG[1].format(128))]),
    "one_entry_w6_seg128": ("hexplane.hip", [(_SEG[0], _SEG[1].format(128))]),
}
