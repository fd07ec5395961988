"""Timing variant (wrong results on purpose) of the staged glue backward: the direction-derivative math compiled out, loads kept."""
VARIANTS = {
    "gb_nomath": ("glue.hip", [("""  const float ddx = dRdx[0] * dRGB[0] + dRdx[1] * dRGB[1] + dRdx[2] * dRGB[2];
  const float ddy = dRdy[0] * dRGB[0] + dRdy[1] * dRGB[1] + dRdy[2] * dRGB[2];
  const float ddz = dRdz[0] * dRGB[0] + dRdz[1] * dRGB[1] + dRdz[2] * dRGB[2];""", """  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += sh[k][0] + sh[k][1] + sh[k][2];
  const float ddx = acc, ddy = acc * 0.5f, ddz = acc * 0.25f;""")]),
}
