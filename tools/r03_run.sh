set -e
export FS2_LIB=variants/libfs2_exp.so
for rep in 1 2; do for m in 2048 256; do echo "== FS2_PROJ_LN_MIN_M=$m"; FS2_PROJ_LN_MIN_M=$m timeout -k 10 200 python tools/bench_latency.py 2>&1 | grep -v amdgpu | head -2; done; done
for m in 2048 256; do echo "== FS2_PROJ_LN_MIN_M=$m"; FS2_PROJ_LN_MIN_M=$m timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep "config4\|config1" | head -6; done
