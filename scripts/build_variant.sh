#!/bin/bash
# build_variant.sh <tag> <extra -D flags...>: a same-box A/B build of the library, klara.jl_amd/lib/libklara_hip_<tag>.so, that differs from the
# default build in compile-time switches only.  Objects are copied from the default build; the translation units named in KLARA_VARIANT_TUS
# (default: every group-layout sampler) are recompiled with the flags.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/build/csrc; dst=$root/build/csrc_$tag
rm -rf "$dst"; cp -r "$src" "$dst"
for tu in ${KLARA_VARIANT_TUS:-klara_mh klara_mala klara_hmc klara_slice}; do rm -f "$dst/$tu.o"; done
make -C "$root/klara.jl_amd/csrc" -j8 OBJDIR="$dst" OUT="$root/klara.jl_amd/lib/libklara_hip_$tag.so" \
  CXXFLAGS="-O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Wno-unused-result $*" > /dev/null
ls -la "$root/klara.jl_amd/lib/libklara_hip_$tag.so"
