#!/bin/bash
# build_variant.sh NAME "UNITS" "EXTRA FLAGS": rebuild some translation units with extra flags and link a library of its own beside the
# product one (project3-pathtracer_amd/lib_NAME/libptamd.so; select it with PT_LIBPTAMD=...).  UNITS: bounce unit numbers 0..7 and / or
# kernels, context, multi.  The other objects come from the product build (make -C csrc first).
set -e
NAME=$1; UNITS=$2; EXTRA=$3
cd "$(dirname "$0")/../project3-pathtracer_amd/csrc"
mkdir -p ../lib_$NAME
FLAGS="--offload-arch=${VARIANT_ARCH:-gfx950} -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize $EXTRA"
REPL=""
for u in $UNITS; do
  case $u in
    [0-7]) f=pt_bounce_g$u ;;
    kernels) f=pt_kernels ;;
    context) f=pt_context ;;
    multi) f=pt_multi ;;
    *) echo "unknown unit $u"; exit 2 ;;
  esac
  /opt/rocm/bin/hipcc $FLAGS -c -o ../lib_$NAME/$f.o $f.hip &
  REPL="$REPL $f.o"
done
wait
OBJS=""
for f in ../lib/*.o; do b=$(basename $f); if echo " $REPL " | grep -q " $b "; then OBJS="$OBJS ../lib_$NAME/$b"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=${VARIANT_ARCH:-gfx950} -shared -o ../lib_$NAME/libptamd.so $OBJS
rm -f ../lib_$NAME/*.o
ls -la ../lib_$NAME/libptamd.so
