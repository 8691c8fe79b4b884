set -e
for v in "" variants/libfs2_w1p32.so variants/libfs2_w2p32.so variants/libfs2_w2p16.so; do
echo "== lib: $v"; if [ -n "$v" ]; then export FS2_LIB=$v; fi
K1_SHAPE=qkv_regw timeout -k 10 100 python tools/k1_probe.py 2>&1 | grep regw
done
