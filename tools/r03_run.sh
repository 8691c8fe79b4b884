set -e
export FS2_LIB=variants/libfs2_exp.so
for rep in 1 2; do for b in 2500000 5000000; do echo "== FS2_CONV_XCD_BYTES=$b"; FS2_CONV_XCD_BYTES=$b timeout -k 10 100 python tools/conv256_sweep.py 25611 2>&1 | grep "M="; done; done
for b in 2500000 5000000; do FS2_CONV_XCD_BYTES=$b timeout -k 10 100 python bench.py --no-cpu-baseline --no-pure 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; done
for b in 2500000 5000000; do FS2_CONV_XCD_BYTES=$b timeout -k 10 100 python bench.py --no-cpu-baseline --no-pure 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; done
