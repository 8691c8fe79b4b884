#!/bin/bash
# tools/build_variant.sh TAG -DFOO=1 ...   ->  variants/libfs2_TAG.so (use with FS2_LIB=variants/libfs2_TAG.so)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p variants/obj_$tag
objs=""
for src in fastspeech2_amd/csrc/*.hip; do
  o=variants/obj_$tag/$(basename ${src%.hip}).o
  extra=$(head -60 $src | grep -m1 "FS2_HIPCC_FLAGS:" | sed 's/.*FS2_HIPCC_FLAGS://' || true)   # per-file flags (fastspeech2_amd/build.py)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I fastspeech2_amd/csrc $extra "$@" -c $src -o $o &   # base flags = fastspeech2_amd/build.py BASE_FLAGS
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libfs2_$tag.so $objs
rm -rf variants/obj_$tag
echo variants/libfs2_$tag.so
