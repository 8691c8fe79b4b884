"""Replays ONLY the FFN conv launches of the bench workload (config 2: the four k=9 conv shapes of the
encoder and decoder FFT blocks, in the proportions of one forward), for rocprofv3 --pmc passes:

    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d out/fetch -- python3 tools/ffn_conv_replay.py
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d out/write -- python3 tools/ffn_conv_replay.py
    python tools/pmc_traffic.py out/fetch out/write profiles/r01_ffn_conv_traffic.json

The LAST `REPLAY_LAUNCHES` conv_gemm dispatches of the run are the replayed ones (printed below)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

ROUNDS = 3


def main():
    dev = torch.device("cuda", 0)
    model, _, _ = bench.build_model("bf16", "bf16", dev)
    eng = model.engine()
    xs = bench.rank_batch(1, 0)
    ilens = [len(x) for x in xs]
    ids = torch.from_numpy(np.concatenate(xs)).to(dev)
    out = eng.forward(ids, ilens)
    torch.cuda.synchronize()
    Mp, Mf, d = sum(ilens), out["frames"], eng.d
    from fastspeech2_amd.hip import call, ptr, stream
    seg = {}
    for name, M, cu in (("enc", Mp, out["cu_phone"]), ("dec", Mf, out["cu_frames"])):
        s = torch.empty((M, 2), dtype=torch.int32, device=dev)
        call("fs2_build_seg", ptr(cu), len(ilens), ptr(s), None, M, stream())
        seg[name] = s
    n = 0
    # round 2: FS2_REPLAY=dec replays the frame-level (decoder) launches only — the launches the bf16 k=9 kernel symbol
    # serves now that the phone-level half runs on the f16x3 kernel — with the residual operand the engine passes
    stacks = (("dec", eng.dec, Mf),) if os.environ.get("FS2_REPLAY") == "dec" else (("enc", eng.enc, Mp), ("dec", eng.dec, Mf))
    for _ in range(ROUNDS):
        for name, st, M in stacks:
            for L in st.layers:
                x = torch.randn(M, d, device=dev).to(torch.bfloat16)
                r = torch.randn(M, d, device=dev)
                h = eng._gemm(x, L["w1"], L["b1"], M, L["units"], d, L["taps"], st.code, seg=seg[name], relu=True)
                eng._gemm(h, L["w2"], L["b2"], M, d, L["units"], L["taps"], st.code, seg=seg[name], out_f32=True, res=r,
                          ksplit=eng._ksplit(M, d, L["units"], L["taps"], st.code))
                n += 2
    torch.cuda.synchronize()
    print("REPLAY_LAUNCHES %d" % n)


if __name__ == "__main__":
    main()
