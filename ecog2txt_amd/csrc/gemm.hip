// bf16 MFMA GEMM, "NT" form:  C[M][N] (+)= alpha * A[M][K] . B[N][K]^T  (+ fused epilogue)
//
// Both operands are K-contiguous, which is the natural MFMA fragment order
// (each lane reads 8 consecutive k for its row/column), so no transposing read
// is ever needed; callers that hold an operand in the other orientation run
// e2t_transpose_bf16 first (DESIGN.md "GEMM orientation").
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves as 2x2, each wave a
// 64x64 patch = 4x4 MFMA 16x16x32 tiles, 64 fp32 accumulators per lane).
// LDS image: [128 rows][8 x 16-B chunks] per operand with the chunk index
// XOR-swizzled by (row & 7): conflict-free for both the ds_write_b128 staging
// pattern and the ds_read_b128 fragment pattern (checked against the gfx950
// lane-group table, MI355X_MICROARCH.md "LDS").  Global->register->LDS staging
// with the next tile's loads issued before the current tile's MFMAs (T14).
//
// Used for every non-recurrent matmul of the path (reference rows: conv a6,
// LSTM input projections a7, aux head a8, vocab projection a9 and all their
// weight/input gradients; SURVEY.md 2.3 K2,K3,K7,K8).
#include "common.h"
#include "ecog2txt_hip.h"

#define BM 128
#define BN 128
#define BK 64

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;
    const bf16_t* mask_src; int ld_mask;
    const int* lens; int rowsB;
    float alpha;
    int flags;
    DropCfg drop; int ld_logical;
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 3) + (chunk ^ (row & 7)); }

__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs p) {
    __shared__ uint4 sA[BM * 8];      // 16 KB
    __shared__ uint4 sB[BN * 8];      // 16 KB

    // XCD-aware tile order: consecutive tile ids (sharing an A row-panel) are
    // spread by the dispatcher over the 8 XCDs (block b -> XCD b%8); remap so
    // each XCD walks a contiguous run of tiles and re-reads its panels from
    // its own L2 (cdna_hip_programming.md T1, bijective form).
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // walk N fastest inside a panel of rows
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging map: thread -> (row = tid/8 + 32*i, chunk = tid%8), i = 0..3
    const int srow = tid >> 3, schunk = tid & 7;
    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tile = [&](int k0) {
        const int kk = k0 + schunk * 8;
        const bool kin = kk < p.K;                   // K is a multiple of 8 by contract
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            const int gm = m0 + r, gn = n0 + r;
            ra[i] = (kin && gm < p.M) ? *(const uint4*)(p.A + (size_t)gm * p.lda + kk) : zero4;
            rb[i] = (kin && gn < p.N) ? *(const uint4*)(p.B + (size_t)gn * p.ldb + kk) : zero4;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            sA[swz(r, schunk)] = ra[i];
            sB[swz(r, schunk)] = rb[i];
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 fa[4], fb[4];
            const int ch = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 va = sA[swz(wm + i * 16 + frow, ch)];
                uint4 vb = sB[swz(wn + i * 16 + frow, ch)];
                fa[i] = *(bf16x8*)&va;
                fb[i] = *(bf16x8*)&vb;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue.  C/D map of mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg
    const bool out_bf16 = p.flags & E2T_GEMM_OUT_BF16;
    const bool accum = p.flags & E2T_GEMM_ACCUMULATE;
    const bool relu = p.flags & E2T_GEMM_RELU;
    const bool dodrop = (p.flags & E2T_GEMM_DROPOUT) && p.drop.rate > 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = m0 + wm + i * 16 + fq * 4 + r;
            if (gm >= p.M) continue;
            bool rowvalid = true;
            if (p.lens) rowvalid = (gm / p.rowsB) < p.lens[gm % p.rowsB];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = n0 + wn + j * 16 + frow;
                if (gn >= p.N) continue;
                float v = acc[i][j][r] * p.alpha;
                if (p.bias) v += p.bias[gn];
                if (relu) v = fmaxf(v, 0.f);
                if (p.mask_src) v = (p.mask_src[(size_t)gm * p.ld_mask + gn] & 0x7FFF) != 0 ? v : 0.f;   // kept & active
                if (dodrop) v *= drop_scale(p.drop, (unsigned long long)gm * p.ld_logical + gn);
                if (!rowvalid) v = 0.f;
                if (out_bf16) {
                    ((bf16_t*)p.C)[(size_t)gm * p.ldc + gn] = f2bf(v);
                } else {
                    float* c = (float*)p.C + (size_t)gm * p.ldc + gn;
                    *c = accum ? (*c + v) : v;
                }
            }
        }
    }
}

extern "C" int e2t_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                int M, int N, int K, const e2t_gemm_epilogue* ep, void* stream) {
    E2T_CHECK_ARG(A && B && C);
    E2T_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
    E2T_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0);
    E2T_CHECK_ARG(lda >= K && ldb >= K && ldc >= N);
    E2T_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0);
    if (M == 0 || N == 0) return E2T_OK;
    GemmArgs p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = 1.0f;
    if (ep) {
        p.bias = ep->bias;
        p.mask_src = (const bf16_t*)ep->relu_bwd_src; p.ld_mask = ep->ld_relu_bwd_src;
        p.lens = ep->row_lens; p.rowsB = ep->rows_per_step > 0 ? ep->rows_per_step : 1;
        p.alpha = ep->alpha;
        p.flags = ep->flags;
        p.drop.rate = ep->drop_rate; p.drop.seed = ep->drop_seed; p.drop.step = ep->drop_step;
        p.drop.stream = ep->drop_stream; p.ld_logical = ep->drop_ld > 0 ? ep->drop_ld : N;
        E2T_CHECK_ARG(!((p.flags & E2T_GEMM_OUT_BF16) && (p.flags & E2T_GEMM_ACCUMULATE)));
    }
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    hipLaunchKernelGGL(k_gemm_nt, dim3(ntm * ntn), dim3(256), 0, (hipStream_t)stream, p);
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}
