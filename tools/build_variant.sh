# Dev: build a variant of the library next to the real one: tools/build_variant.sh NAME "-DFLAG=1 ..." [source base name, default igemm]
# -> contrad_amd/csrc/variants/libcontrad_NAME.so   (use with CONTRAD_HIP_LIB=<that path>)
set -e
cd "$(dirname "$0")/.."
N=$1; F=$2; WHICH=${3:-igemm}
D=contrad_amd/csrc/variants; mkdir -p $D/$N
for s in contrad_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  if [ "$b" = "$WHICH" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -c $s -o $D/$N/$b.o & else cp contrad_amd/csrc/$b.o $D/$N/$b.o; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcontrad_$N.so $D/$N/*.o
rm -rf $D/$N
ls -la $D/libcontrad_$N.so
