#!/usr/bin/env bash
# A/B build of one kernel source: tools/ab_build.sh <name> <source stem> <extra hipcc flags...> compiles
# csrc/<stem>.hip with the flags and links l2hmc/_lib/libl2q_<name>.so from the other objects of the
# regular build; run anything with L2Q_LIB_NAME=libl2q_<name>.so (tools/gpu_job_ab.sh does).
set -e
cd "$(dirname "$0")/../l2hmc-qcd_amd/csrc"
name="$1"; stem="$2"; shift 2
mkdir -p obj_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $stem.hip -o obj_ab/${stem}_$name.o
objs=$(ls obj/*.o | grep -v "obj/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../l2hmc/_lib/libl2q_$name.so $objs obj_ab/${stem}_$name.o -ldl
echo " -> libl2q_$name.so"
