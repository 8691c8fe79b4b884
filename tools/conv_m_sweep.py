"""Decoder FFN conv shapes (k=9, 256 -> 1024 and 1024 -> 256, bf16) timed alone over a sweep of token counts M:
shows how much of a launch is the last, partly filled round of workgroups (768 resident 128 x 128 tiles at 3 per CU).
    python tools/conv_m_sweep.py [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    dev = torch.device("cuda", 0)
    model, _, _ = bench.build_model("bf16", None, dev)
    eng = model.engine()
    from fastspeech2_amd.hip import call, ptr, stream
    L, st, d = eng.dec.layers[0], eng.dec, eng.d
    Ms = [int(a) for a in sys.argv[1:]] or [12288, 24576, 25603, 26112, 27648, 30720, 36864, 49152]
    for M in Ms:
        cu = torch.tensor([0, M], dtype=torch.int32, device=dev)
        seg = torch.empty((M, 2), dtype=torch.int32, device=dev)
        call("fs2_build_seg", ptr(cu), 1, ptr(seg), None, M, stream())
        x = torch.randn(M, d, device=dev).to(torch.bfloat16)
        res = []
        for which in (1, 2):
            def run():
                if which == 1:
                    return eng._gemm(x, L["w1"], L["b1"], M, L["units"], d, L["taps"], st.code, seg=seg, relu=True)
                return eng._gemm(h, L["w2"], L["b2"], M, d, L["units"], L["taps"], st.code, seg=seg, out_f32=True,
                                 ksplit=eng._ksplit(M, d, L["units"], L["taps"], st.code))
            if which == 1:
                h = run()
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            res.append((us, 2.0 * M * d * L["units"] * L["taps"] / us * 1e-6))
        tiles = ((M + 127) // 128) * (L["units"] // 128)
        print("M=%6d  conv1 %3d tiles = %.2f rounds: %6.1f us %5.0f TF | conv2: %6.1f us %5.0f TF"
              % (M, tiles, tiles / 768.0, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)


if __name__ == "__main__":
    main()
