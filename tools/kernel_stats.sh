#!/bin/bash
# Registers, scratch and LDS of the kernels of one translation unit (device code compiled to assembly with the product's flags):
#   bash tools/kernel_stats.sh k_sdf.hip [extra flags...]        -> /tmp/asmp/<name>.s and one line per kernel
# (k_compact / k_sdf / k_large are built with -DRPT_GUARD_PER_OP: pass it.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
mkdir -p /tmp/asmp
OUT=/tmp/asmp/${SRC%.hip}.s
(cd $ROOT/rust-pathtracer_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden \
    -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp "$@" --cuda-device-only -S $SRC -o $OUT 2> >(grep -v "argument unused" >&2))
awk '/^[ \t]+\.amdhsa_kernel /{name=$2} /; NumVgprs:/{v=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{o=$3} /; LDSByteSize:/{printf "%-72s vgprs %3s scratch %4s lds %6s occupancy %s\n", substr(name,1,72), v, s, $3, o}' $OUT
