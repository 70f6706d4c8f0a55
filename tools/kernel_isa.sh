#!/bin/bash
# ISA of one kernel instance of a source file: tools/kernel_isa.sh <source.hip> <mangled-name-regex> > out.s   (CPU only; build.py's flags + DN_HIPCC_FLAGS)
SRC=$1; PAT=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I include -I dnascent_amd/csrc $DN_HIPCC_FLAGS -S --cuda-device-only -o /tmp/_isa.s $SRC 2>/dev/null
awk -v pat="^$PAT.*: *; @" '$0 ~ pat {f=1} f{print} /^\.Lfunc_end/{if(f){exit}}' /tmp/_isa.s
