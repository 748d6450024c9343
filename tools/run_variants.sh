#!/bin/bash
# Time every rust-pathtracer_amd/variants/*.so on the GPU box: bash tools/run_variants.sh [program args...]
# (default: tools/ab_time.py c2).  One process per variant (RPT_LIB is read at import).
PROG=("$@"); [ ${#PROG[@]} -eq 0 ] && PROG=(tools/ab_time.py c2)
for so in rust-pathtracer_amd/variants/*.so; do
    echo -n "$(basename $so .so): "
    RPT_LIB=$PWD/$so python3 "${PROG[@]}" 2>/dev/null | tail -1
done
