# python tools/mkvariants.py tools/variants/r04_pointdiv.py   (round 4: occupancy of the division-form per-point pass)
_WV = ("#define S3G_HEX_POINTDIV_WAVES 4", "#define S3G_HEX_POINTDIV_WAVES {}")
VARIANTS = {
    "pointdiv_w3": ("hexplane.hip", [(_WV[0], _WV[1].format(3))]),
    "pointdiv_w5": ("hexplane.hip", [(_WV[0], _WV[1].format(5))]),
    "pointdiv_w6": ("hexplane.hip", [(_WV[0], _WV[1].format(6))]),
    "pointdiv_w8": ("hexplane.hip", [(_WV[0], _WV[1].format(8))]),
}
