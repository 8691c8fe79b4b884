"""Builds fastspeech2_amd/csrc/*.hip into libfs2_gfx950.so with hipcc (cross-compiles without a GPU).

    python -m fastspeech2_amd.build            # rebuild if sources are newer than the .so
"""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libfs2_gfx950.so")
ARCH = "gfx950"
# -fno-slp-vectorize: with the SLP vectoriser hipcc (ROCm 7.2) packs adjacent scalar fp32 arithmetic into v_pk_*_f32
# instructions (SGPR-pair / op_sel operands); such code in layernorm_bwd computed whole rows wrong, now and then, when its
# wave shared a SIMD with another kernel's bf16 MFMAs (tools/lnb_race.py).  Off for every file:
# no measurable cost (bench.py / bench_train.py A/B/A/B; the attention launches gain 1-2 %).
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def extra_flags(src: str):
    """Per-file compiler flags, declared in the source itself on a line `// FS2_HIPCC_FLAGS: ...` (first 60 lines)."""
    with open(src) as f:
        for _, line in zip(range(60), f):
            if "FS2_HIPCC_FLAGS:" in line:
                return line.split("FS2_HIPCC_FLAGS:", 1)[1].split()
    return []


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [hipcc, "--offload-arch=" + ARCH] + BASE_FLAGS + ["-I", CSRC] + extra_flags(src) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
