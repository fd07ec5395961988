# python tools/mkvariants.py tools/variants/r04_split_hazard.py
# The staging store of the split inference kernel (csrc/mlp.hip).  Tree (round 4) = every stored component re-written by a plain
# v_mov_b32 right before ds_write_b128, no wait states (a register dependency on a non-packed VALU write);
#   split_nopad  = nothing between the packed products and the store (round 3's first build: stale lanes 48..63) -- the stress test
#                  (tools/diag_split.py 1200000 1000) must CATCH this one: 110 872 wrong rows in 1000 launches;
#   split_nops   = round 3's cure: 16 wait states.
# Results of all three: profiles/r04_split_hazard.jsonl.
_VMOV = '    if (SPLIT) asm volatile("v_mov_b32 %0, %0\\n\\tv_mov_b32 %1, %1\\n\\tv_mov_b32 %2, %2\\n\\tv_mov_b32 %3, %3" : "+v"(prod.x), "+v"(prod.y), "+v"(prod.z), "+v"(prod.w));'
#   split_scalarized = an EMPTY asm statement with the same per-component operands: no wait state, no v_mov -- it only makes hipcc form the
#                  products with plain v_mul_f32 instead of v_pk_mul_f32 (separates "packed instruction" from "v_mov dependency").
VARIANTS = {
    "split_scalarized": ("mlp.hip", [(_VMOV, '    if (SPLIT) asm volatile("" : "+v"(prod.x), "+v"(prod.y), "+v"(prod.z), "+v"(prod.w));')]),
    "split_nopad": ("mlp.hip", [(_VMOV, "    /* no pad */")]),
    "split_nops": ("mlp.hip", [(_VMOV, '    if (SPLIT) asm volatile("s_nop 7\\n\\ts_nop 7" : "+v"(prod.x), "+v"(prod.y), "+v"(prod.z), "+v"(prod.w));')]),
}
