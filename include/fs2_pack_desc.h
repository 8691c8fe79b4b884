/* One weight of fs2_pack_weights (include/fs2_hip.h): the arguments of fs2_pack_weight as a 64-byte record. */
#ifndef FS2_PACK_DESC_H
#define FS2_PACK_DESC_H
typedef struct fs2_pack_desc {
    const float* src;            /* (N, C, k) fp32, the nn.Conv1d / nn.Linear parameter */
    void* fwd;                   /* [k][N][C] operand of the forward GEMM (ld_fwd per row), or NULL */
    void* bwd;                   /* [k][C][N] flipped-tap operand of the input-gradient GEMM (ld_bwd per row), or NULL */
    int N, C, k, ld_fwd, ld_bwd, reserved;
    long long tap_stride_fwd, tap_stride_bwd;   /* elements between two taps */
} fs2_pack_desc;
#endif
