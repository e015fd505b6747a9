#!/bin/bash
# Builds a second copy of libsmc_hip.so with extra compiler flags (e.g. -DSMC_LEAN_DIV_SIX) into build/ab/<name>/ for A/B
# timing on ONE box:   tools/ab_build.sh six -DSMC_LEAN_DIV_SIX ;  SMC_HIP_LIB=build/ab/six/libsmc_hip.so python tools/chain_latency.py
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src="$root/python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd/csrc"
out="$root/build/ab/$name"
mkdir -p "$out/csrc"
cp "$src"/*.hip "$src"/*.h "$src"/Makefile "$out/csrc/"
mkdir -p "$out/../../../include" 2>/dev/null || true
# the sources include ../../include/smc_hip.h relative to csrc
mkdir -p "$out/../include"; cp "$root/include/smc_hip.h" "$out/../include/"
make -C "$out/csrc" -j6 EXTRA="$*" OUT=../libsmc_hip.so >/dev/null
ls -la "$out/libsmc_hip.so"
