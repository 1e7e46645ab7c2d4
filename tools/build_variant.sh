#!/bin/bash
# Experimental build of the whole library with extra flags: tools/build_variant.sh <name> [-DPTX_BLOCK=768 ...]
# -> distributed-path-tracer_amd/exp/libptx_<name>.so (select it with PTX_LIB=...). All objects are rebuilt.
set -e
name=$1; shift
C=$(dirname $0)/../distributed-path-tracer_amd/csrc; O=$C/build/var_$name; mkdir -p $O $C/../exp
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-unroll-loops -fno-slp-vectorize $*"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c -o $O/kernels.o $C/kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c -o $O/wavefront.o $C/wavefront.hip
for f in ptx_api scene_build gltf_load png_read jpeg_read hdr_read; do /opt/rocm/bin/hipcc -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $F -c -o $O/$f.o $C/$f.cpp; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/../exp/libptx_$name.so $O/kernels.o $O/wavefront.o $O/ptx_api.o $O/scene_build.o $O/gltf_load.o $O/png_read.o $O/jpeg_read.o $O/hdr_read.o -lz
echo built $C/../exp/libptx_$name.so
