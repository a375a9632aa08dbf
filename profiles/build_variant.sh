# build_variant.sh NAME GEOMNUM "EXTRA FLAGS": rebuild one bounce unit with extra flags and link a library beside the product one
set -e
NAME=$1; G=$2; EXTRA=$3
cd /root/repo/project3-pathtracer_amd/csrc
mkdir -p ../lib_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize $EXTRA -c -o ../lib_$NAME/pt_bounce_g$G.o pt_bounce_g$G.hip
OBJS=""
for f in ../lib/*.o; do b=$(basename $f); if [ "$b" = "pt_bounce_g$G.o" ]; then OBJS="$OBJS ../lib_$NAME/$b"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib_$NAME/libptamd.so $OBJS
rm -f ../lib_$NAME/*.o
ls -la ../lib_$NAME/libptamd.so
