# python tools/mkvariants.py tools/variants/r04_hex_cache.py   (round 4: cacheable vs streaming loads of the T rows in the scatter walks:
# each row is read by three orientation walks; measured 0.81 (streaming) vs ... ms)
VARIANTS = {
    "t_loads_cacheable": ("hexplane.hip", [("template <> __device__ __forceinline__ float load_g<float>(const float* p) { return G_NONTEMPORAL_LOAD ? __builtin_nontemporal_load(p) : *p; }",
                                           "template <> __device__ __forceinline__ float load_g<float>(const float* p) { return *p; }")]),
}
