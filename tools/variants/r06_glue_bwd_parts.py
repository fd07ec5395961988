"""Where do the 275 us of glue_backward_kernel go?  TIMING variants (wrong results on purpose): parts of the kernel compiled out.
    python tools/mkvariants.py tools/variants/r06_glue_bwd_parts.py; bash tools/ab_kstats.sh "tree gb_nosh gb_norest gb_nodshs gb_nophase2" (grep glue_backward)"""
_LOAD = """  float sh[16][3];
  load_sh(a, p, sh);
  float dRGB[3];"""
_NOLOAD = """  float sh[16][3];
#pragma unroll
  for (int k = 0; k < 16; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) sh[k][c] = __int_as_float(p & 1);
  float dRGB[3];"""
_REST = """  for (int e = threadIdx.x; e < n * 45; e += 256) {
    const int q = e / 45, kc = e % 45 + 3;
    a.g_f_rest[(size_t)p0 * 45 + e] = stage[q * 19 + kc / 3] * stage[q * 19 + 16 + kc % 3];
  }"""
_DSHS = """  if (a.g_dshs != nullptr) {
    const float l1 = a.g_dshs_l1"""
VARIANTS = {
    "gb_nosh": ("glue.hip", [(_LOAD, _NOLOAD)]),                       # no per-lane coefficient-row loads in phase 1
    "gb_norest": ("glue.hip", [(_REST, "")]),                          # no g_f_rest sweep in phase 2
    "gb_nodshs": ("glue.hip", [(_DSHS, "  if (false) {\n    const float l1 = a.g_dshs_l1")]),   # no g_dshs sweep (and no dshs read)
    "gb_nophase2": ("glue.hip", [(_REST, ""), (_DSHS, "  if (false) {\n    const float l1 = a.g_dshs_l1")]),
}
