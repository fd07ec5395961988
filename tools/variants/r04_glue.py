"""Render-glue forward A/B (profiles/r04_glue.txt).  The comparison build is the round-3 kernel -- 24 dwordx4 ROW loads per lane,
every instruction touching 64 different lines -- compiled from history, not from string edits:

    git show 7020f56:s3gaussian_amd/csrc/glue.hip > s3gaussian_amd/csrc/_variant_glue_rows.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -c <that> -o /tmp/glue_rows.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls s3gaussian_amd/lib/*.o | grep -v /glue.o) /tmp/glue_rows.o \\
          -o s3gaussian_amd/lib/variants/libs3g_glue_rows.so
    S3G_LIB_PATH=$PWD/s3gaussian_amd/lib/variants/libs3g_glue_rows.so python tools/glue_probe.py

Tried on the way and dropped: 16-byte loads for the coalesced sweeps (four LDS read-modify-writes each: 150 vs 133 us for the dword
sweeps) and index arithmetic by division (133 vs 125 us with the incremental (q, r) walk)."""
VARIANTS = {}
