#!/bin/bash
# A/B of an engine switch in the overlapped bench, interleaved: bash tools/bench_ab.sh reg_w   (run on the GPU box)
sw=${1:-reg_w}
for rep in 1 2 3; do for v in 1 0; do
  r=$(timeout 200 python bench.py --no-cpu-baseline --no-pure --engine-set $sw=$v --steps 48 --warmup 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])") || exit 1
  echo "$sw=$v: $r ms"
done; done
