#!/bin/bash
# Experimental build that differs from the shipped library in wavefront.hip only: tools/build_wf_variant.sh <name> [-DPTX_WF_UNIT=128 ...]
# -> distributed-path-tracer_amd/exp/libptx_<name>.so (select it with PTX_LIB=...). The other objects are the shipped build's (make first).
set -e
name=$1; shift
C=$(dirname $0)/../distributed-path-tracer_amd/csrc; O=$C/build/var_$name; mkdir -p $O $C/../exp
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-unroll-loops -fno-slp-vectorize $*"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c -o $O/wavefront.o $C/wavefront.hip
B=$C/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/../exp/libptx_$name.so $B/kernels.o $O/wavefront.o $B/ptx_api.o $B/scene_build.o $B/gltf_load.o $B/png_read.o $B/jpeg_read.o $B/hdr_read.o -lz
echo built exp/libptx_$name.so
