#!/bin/bash
# tools/build_alt.sh NAME [-DFLAG ...]: an experimental build of gemm.hip linked with the current objects of everything else ->
# tools/_bin/libphenaki_NAME.so (load with PK_LIB_PATH for a same-box A/B; tools/_bin is git-ignored but travels to the GPU box)
set -e
name=$1; shift
src=${PK_ALT_SRC:-gemm}
mkdir -p tools/_bin
python -m phenaki_pytorch_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c phenaki_pytorch_amd/csrc/$src.hip -o tools/_bin/${src}_$name.o
objs=$(ls phenaki_pytorch_amd/csrc/_obj/*.o 2>/dev/null | grep -v "/$src.o" || true)
[ -z "$objs" ] && { echo "no objects found"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libphenaki_$name.so $objs tools/_bin/${src}_$name.o
echo built tools/_bin/libphenaki_$name.so
