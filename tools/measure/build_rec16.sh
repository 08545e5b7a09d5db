#!/bin/bash
# The 16-bit-record build of the library (qoi_decode_core.h "16-bit records"): build/rec16/libqoi_mi355x.so - qoi_decode.hip and
# qoi_host.hip compiled with -DQOIMI_REC16=1, the other objects from qoi_amd/lib/obj (run make first).  QOIMI_LIB=build/rec16/libqoi_mi355x.so
# selects it for the tests and tools/measure/dec_time.py.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$R/build/rec16"
for src in qoi_decode qoi_host; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-int-to-pointer-cast -DQOIMI_REC16=1 "$@" -c "$R/qoi_amd/csrc/$src.hip" -o "$R/build/rec16/$src.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build/rec16/libqoi_mi355x.so" "$R/build/rec16/qoi_host.o" "$R/qoi_amd/lib/obj/qoi_encode.o" "$R/build/rec16/qoi_decode.o" "$R/qoi_amd/lib/obj/qoi_synth.o"
echo "built build/rec16/libqoi_mi355x.so"
