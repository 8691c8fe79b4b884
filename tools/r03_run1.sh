set -e
O=gpurun_out/r03_b; mkdir -p $O
timeout -k 10 200 python tools/conv256_sweep.py 14336 16384 20480 24576 25611 28672 32768 > $O/conv256_sweep.txt 2>&1
timeout -k 10 100 python tools/k1_probe.py > $O/k1_probe.txt 2>&1
for pg in 2 3 4 2 3 4; do timeout -k 10 100 python bench.py --no-cpu-baseline --no-pure --phone-group $pg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('phone_group', d['config']['launch_env']['phone_group'], d['ms_per_step'], d['config']['host_ms_per_step'])" >> $O/phone_group.txt; done
bash tools/k1_pmc.sh r03_b qkv
