#!/bin/bash
# Static instruction mix of conv_split_fast_kernel<4> (the whole kernel: K loop + epilogue; the conversions live in the epilogue only) for the
# epilogue header of a given commit (default: the working tree):   tools/epilogue_isa_count.sh [COMMIT]   (no GPU needed: hipcc -save-temps)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
cp -r "$ROOT/tise_toolbox_amd/csrc" "$T/csrc"; mkdir -p "$T/include"; cp "$ROOT/include/tise_hip.h" "$T/include/"
if [ -n "$1" ]; then git -C "$ROOT" show "$1:tise_toolbox_amd/csrc/conv_epilogue.h" > "$T/csrc/conv_epilogue.h"; fi
mkdir -p "$T/a/b"; mv "$T/csrc" "$T/a/b/csrc"; mv "$T/include" "$T/a/include"      # common.h includes ../../include/tise_hip.h
cd "$T/a/b/csrc" && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -c conv_split.hip -o "$T/cs.o" -save-temps=obj 2>/dev/null
S=$(ls "$T"/conv_split-hip-amdgcn-*.s)
L=$(grep -n "^_Z22conv_split_fast_kernelILi4ELb0ELb0EEv14tise_conv_args:" "$S" | cut -d: -f1)
awk -v l=$L 'NR>=l' "$S" | awk '/s_endpgm/{print; exit} {print}' > "$T/k4.s"
echo "conv_split_fast_kernel<4>, epilogue header of ${1:-the working tree}: $(wc -l < "$T/k4.s") lines of ISA, $(grep -c v_mfma "$T/k4.s") MFMAs"
grep -E "^\s+v_(cvt_f16_f32|cvt_f32_f16|cvt_pk_f16_f32|fma_mix|fma_f32|fmac_f32|pk_fma_f32|pk_mul_f32|pk_add_f32|add_f32|sub_f32|mul_f32|max_f32|max3_f32)" "$T/k4.s" | awk '{print $1}' | sort | uniq -c
rm -rf "$T"
