set -e
export FS2_LIB=variants/libfs2_stamp.so C256_ONLY=new C256_STAMPS=1
timeout -k 10 100 python tools/conv256_check.py 16384 2>&1 | grep -v amdgpu | head -30
