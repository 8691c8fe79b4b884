#!/bin/bash
# bench.py step time over (phone_group, frame_streams) settings, interleaved; run on the GPU box
for rep in 1 2; do
for pg in 2 3 4; do for fs in 2 3; do
  r=$(timeout 200 python bench.py --no-cpu-baseline --no-pure --phone-group $pg --frame-streams $fs --steps 48 --warmup 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "phone_group=$pg frame_streams=$fs: $r ms"
done; done; done
