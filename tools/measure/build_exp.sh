#!/bin/bash
# Experimental build of the library with one source compiled with extra defines: build/exp_<name>/libqoi_mi355x.so
# usage: tools/measure/build_exp.sh <name> <enc|dec|host> <flags...>      (the other objects come from qoi_amd/lib/obj: run make first)
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; name=$1; what=$2; shift 2
case $what in enc) src=qoi_encode;; dec) src=qoi_decode;; host) src=qoi_host;; *) echo "enc|dec|host"; exit 2;; esac
mkdir -p "$R/build/exp_$name"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-int-to-pointer-cast "$@" -c "$R/qoi_amd/csrc/$src.hip" -o "$R/build/exp_$name/$src.o"
objs=""; for o in qoi_host qoi_encode qoi_decode qoi_synth; do if [ $o = $src ]; then objs="$objs $R/build/exp_$name/$o.o"; else objs="$objs $R/qoi_amd/lib/obj/$o.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build/exp_$name/libqoi_mi355x.so" $objs
echo "built build/exp_$name/libqoi_mi355x.so"
