#!/bin/bash
# ablation timings of the 256 x 256 kernel (variants built by tools/build_variant.sh); run on the GPU box
out=gpurun_out/$1; mkdir -p $out; shift
export C256_ONLY=new
for v in base "$@"; do
  if [ $v = base ]; then unset FS2_LIB; else export FS2_LIB=variants/libfs2_b2_$v.so; fi
  echo "== $v" >> $out/ablate.txt
  timeout -k 10 100 python tools/conv256_check.py 32768 >> $out/ablate.txt 2>&1 || exit 1
done
grep -v amdgpu $out/ablate.txt
