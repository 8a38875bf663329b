#!/usr/bin/env bash
# A/B builds of one force-kernel source: tools/force_ab.sh <name> <extra hipcc flags...> compiles
# csrc/su3_force_link.hip with the flags and links l2hmc/_lib/libl2q_<name>.so from the other
# objects; run with L2Q_LIB_NAME=libl2q_<name>.so python tools/force_bench.py
set -e
cd "$(dirname "$0")/../l2hmc-qcd_amd/csrc"
name="$1"; shift
mkdir -p obj_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c su3_force_link.hip -o obj_ab/su3_force_link_$name.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|VGPRs Spill" | sort | uniq -c | tr '\n' ';'
objs=$(ls obj/*.o | grep -v su3_force_link.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../l2hmc/_lib/libl2q_$name.so $objs obj_ab/su3_force_link_$name.o
echo " -> libl2q_$name.so"
