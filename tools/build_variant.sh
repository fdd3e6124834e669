#!/bin/bash
# A dev-build variant of the env kernels: tools/abl/lib_dev_<name>.so from ec_env.hip with extra -D flags (the other
# objects come from the regular build).  usage: bash tools/build_variant.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../marl_amd/csrc"
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-command-line-argument -DEC_DEV_BUILD "$@" -c ec_env.hip -o /tmp/ec_env_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/lib_dev_$NAME.so /tmp/ec_env_$NAME.o $(ls *.o | grep -v "^ec_env.o")
echo built tools/abl/lib_dev_$NAME.so
