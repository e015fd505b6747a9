#!/bin/bash
# Builds libsmc_hip.so from the sources of a git revision into build/ab/<name>/ (for A/B timing against the working tree on
# ONE box):   tools/ab_rev.sh r02 37e15b7 ;  SMC_HIP_LIB=build/ab/r02/libsmc_hip.so python bench.py ...
set -e
name=$1; rev=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
pkg="python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd"
out="$root/build/ab/$name"
rm -rf "$out"; mkdir -p "$out/csrc" "$out/../include"
for f in $(git -C "$root" ls-tree --name-only "$rev" "$pkg/csrc/"); do
    git -C "$root" show "$rev:$f" > "$out/csrc/$(basename $f)"
done
git -C "$root" show "$rev:include/smc_hip.h" > "$out/../include/smc_hip.h"
make -C "$out/csrc" -j6 EXTRA="$*" OUT=../libsmc_hip.so >/dev/null
ls -la "$out/libsmc_hip.so"
