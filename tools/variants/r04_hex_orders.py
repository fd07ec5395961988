# python tools/mkvariants.py tools/variants/r04_hex_orders.py   (round 4: per-level walk orders of the HexPlane scatter, same-box A/B)
_PL = ("#ifndef S3G_HEX_PER_LEVEL\n#define S3G_HEX_PER_LEVEL 1\n#endif", "#define S3G_HEX_PER_LEVEL {}")
_WV = ("#define S3G_HEX_SCATTER_WAVES 4", "#define S3G_HEX_SCATTER_WAVES {}")
_SEG = ("static inline int segment_length(int P) { return P >= 1000000 ? 256 : 128; }", "static inline int segment_length(int P) {{ return {}; }}")
VARIANTS = {
    "finest_order": ("hexplane.hip", [(_PL[0], _PL[1].format(0))]),                       # rounds 1-3: three orders, two levels per walk
    "perlevel_w5": ("hexplane.hip", [(_WV[0], _WV[1].format(5))]),
    "perlevel_w6": ("hexplane.hip", [(_WV[0], _WV[1].format(6))]),
    "perlevel_seg512": ("hexplane.hip", [(_SEG[0], _SEG[1].format(512))]),
    "perlevel_seg1024_w5": ("hexplane.hip", [(_SEG[0], _SEG[1].format(1024)), (_WV[0], _WV[1].format(5))]),
}
