#!/bin/bash
# usage: tools/isa_dump.sh <file.hip> <mangled-name-substring> [extra flags]   (build container, no GPU)
# compiles one translation unit to gfx950 assembly with the product's flags and prints the named kernel's resource
# lines (VGPRs, LDS, scratch); the kernel's text is left in /tmp/isa/<substring>.s
src=$1; name=$2; shift 2
mkdir -p /tmp/isa
cd "$(dirname "$0")/../cuda-efficient-features_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S "$@" $src -o /tmp/isa/unit.s 2>&1 | grep -v "warning: argument unused" | head -30
awk -v n="$name" '$0 ~ "^_Z.*" n ".*:" {on=1} on {print} /\.end_amdhsa_kernel/ {if (on) exit}' /tmp/isa/unit.s > /tmp/isa/$name.s
grep -E "amdhsa_next_free_vgpr|amdhsa_group_segment_fixed_size|amdhsa_private_segment_fixed_size" /tmp/isa/$name.s
wc -l /tmp/isa/$name.s
