"""cProfile of the host side of a batch-1 forward (BASELINE config 1: 120 phones -> ~810 frames, bf16): where the
0.78 ms go when the GPU is not the limit.
    python tools/host_profile_infer.py [n_forwards]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    model, _, _ = bench.build_model("bf16", None, dev)
    eng = model.engine()
    x = bench.rank_batch(1, 0)[0]
    ids = torch.from_numpy(np.asarray(x)).to(dev)
    ilens = [len(x)]
    for _ in range(20):
        eng.forward(ids, ilens)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        eng.forward(ids, ilens)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
