#!/bin/bash
# builds gpurun_variants/librt355_<name>.so with extra compiler flags (experiments only): tools/build_variant.sh name -DFOO=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
out=../../gpurun_variants; mkdir -p $out/obj_$name
C="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function -I../../include $*"
/opt/rocm/bin/hipcc $C -c rt_api.hip -o $out/obj_$name/rt_api.o &
/opt/rocm/bin/hipcc $C -ffp-contract=off -c rt_kd.hip -o $out/obj_$name/rt_kd.o &
/opt/rocm/bin/hipcc $C -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=on -c rt_pipeline_f32.hip -o $out/obj_$name/rt_pipeline_f32.o &
/opt/rocm/bin/hipcc $C -ffp-contract=off -c rt_pipeline_f64.hip -o $out/obj_$name/rt_pipeline_f64.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/librt355_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
ls -la $out/librt355_$name.so
