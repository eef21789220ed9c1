#!/bin/bash
# Probe builds of the library (tools only; the product sources under tiny-newsrec_amd/csrc carry NO probe code since round 6):
# a copy of the sources with tools/probes/csrc_probes.patch applied, built with the probe's -D flags.
#   tools/probes/build.sh _probe -DTNR_PROBES=2      -> tools/_probe/libtnr_hip.so   (gemm_probe.py, epi_bound.py: runtime switches)
#   tools/probes/build.sh _noepi -DTNR_NOEPI         -> the NT kernels minus their epilogues (gemm_ab.py LIB=tools/_noepi)
#   tools/probes/build.sh _ntst  -DTNR_NT_STAMPS     -> barrier-arrival stamps of the NT ping-pong kernel (nt_stamps.py)
#   tools/probes/build.sh _tnst  -DTNR_TN_STAMPS [-DTNR_TN_NOLOADSEG]   -> ... of the weight-gradient kernel (tn_stamps.py)
#   tools/probes/build.sh _attn  -DTNR_ATTN_STAMPS   -> attn_stamps.py ; -DTNR_SG_SKIP=n / -DTNR_UF_SKIP=n: tools/scratch/sgemm_probe.py
# The shader-clock probe needs no such build any more: tnr_gemm_clock_stamps (include/tnr_hip.h, tools/gemm_clock.py).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
src=$ROOT/tools/$name           # two levels under the root, like csrc: the Makefile's ../../include resolves
rm -rf "$src"; mkdir -p "$src"
cp "$ROOT"/tiny-newsrec_amd/csrc/{*.hip,*.h,*.cpp,Makefile} "$src"/
patch -s -d "$src" -p1 < "$ROOT/tools/probes/csrc_probes.patch"
make -s -C "$src" -j8 EXTRA="$*"
rm -f "$src"/*.o
echo "built tools/$name/libtnr_hip.so with: $*"
