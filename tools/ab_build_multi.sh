#!/bin/bash
# A/B build of libnisqa_hip.so with extra flags on SEVERAL translation units:
#   tools/ab_build_multi.sh NAME "-mllvm -some-flag ..." unit1 unit2 ...
# -> ab_libs/NAME.so (all other objects are the regular ones; run `make -C nisqa_amd/csrc` first; per-unit flags of the
# Makefile -- train_td.o: -fno-slp-vectorize -- are kept).  Use with NISQA_HIP_LIB=$PWD/ab_libs/NAME.so
set -e
NAME=$1; FLAGS=$2; shift; shift
UNITS=" $* "
cd "$(dirname "$0")/../nisqa_amd/csrc"
mkdir -p ../../ab_libs /tmp/nq_ab_$NAME
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -I../../include"
for f in *.hip; do s=${f%.hip}
  X=""; [ $s = train_td ] && X="-fno-slp-vectorize"
  if [[ "$UNITS" == *" $s "* ]]; then /opt/rocm/bin/hipcc $F $X $FLAGS -c $s.hip -o /tmp/nq_ab_$NAME/$s.o; else cp $s.o /tmp/nq_ab_$NAME/$s.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/$NAME.so /tmp/nq_ab_$NAME/*.o
python3 ../../tools/isa_lint.py ../../ab_libs/$NAME.so gfx950
ls -la ../../ab_libs/$NAME.so
