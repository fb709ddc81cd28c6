// bf16 MFMA GEMM with fp32 accumulation and a fused epilogue, in two operand forms:
//   NT  C[M][N] (+)= alpha * A[M][K] . B[N][K]^T   both operands K-contiguous, the natural MFMA fragment order (each lane
//       reads 8 consecutive k of its row / column with one ds_read_b128);
//   TN  C[M][N] (+)= alpha * sum_k A[k][M] B[k][N]  both operands K-major -- the shape of every weight gradient (K = S*B
//       rows of activations and of their gradients exactly as the layers wrote them): fragments are gathered from the
//       K-major LDS tile with the transposing read ds_read_b64_tr_b16, no operand is transposed in memory.
//
// Instances of ONE template (tile BM x BN x 64, WM x WN waves):
//   128 x 128, 4 waves as 2x2 (64x64 per wave, 64 fp32 accumulators per lane), 2 workgroups per CU -- small, split-K
//              and epilogue-rich products, NT and TN;
//   256 x 256, 8 waves as 2x4 (128x64 per wave, 128 accumulators), 1 workgroup per CU, 128 KiB of LDS -- large plain
//              NT products (input projections): 128 flop per staged byte.
// Staging: direct-to-LDS DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction), two LDS stages, ONE barrier
// per K tile: tile t+1 is in flight while tile t is multiplied.  NT image per operand: [rows][8 x 16-B chunks], chunk
// index XOR-swizzled by (row & 7); TN image: [64 k-rows][BM columns], chunk index XOR-swizzled so that the 8 k-rows of a
// transposing-read phase hit 8 different 32-B slots.  The DMA writes LDS linearly (wave base + lane*16), so the swizzle
// is applied to the per-lane SOURCE address and again on the fragment read (cdna_hip_programming.md rule 21).  Both
// layouts are conflict-free (SQ_LDS_BANK_CONFLICT = 0).  Rows beyond M/N are clamped to the last valid row (their
// products land in output rows/columns that are never stored); an NT K tail (K % 64 != 0) is staged through registers
// with zero fill, TN k-rows beyond K come from a zero page.
//
// Split-K (grid.y): weight-gradient GEMMs have K = S*B (8704) but only a few dozen output tiles; splitting K fills the
// 256 CUs.  Partial tiles go to dense fp32 slabs in a caller-provided workspace and are summed in fixed order by
// k_splitk_reduce, which applies the whole epilogue (deterministic; fp32 atomics were measured 5-10x slower: ~12 M
// atomics per GEMM).  Batched launch (grid.z): several products of one shape with strided operands in one launch.
//
// Used for every non-recurrent matmul of the path (reference rows: conv a6, LSTM input projections a7, aux head a8,
// vocab projection a9 and all their weight/input gradients; SURVEY.md 2.3 K2,K3,K7,K8).
#include "common.h"
#include "ecog2txt_hip.h"
#include <stdlib.h>
#include <algorithm>

#define BK 64

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;
    const bf16_t* mask_src; int ld_mask;
    const int* lens; int rowsB;
    int rowsG;                 // row order of the output: m = (tg*rowsB + b)*rowsG + g, step t = tg*rowsG + g (1: plain time-major)
    float alpha;
    int flags;
    DropCfg drop; int ld_logical;
    float* last_col_out;       // if set: column N-1 of the product goes to last_col_out[row] instead of C
    int splits;
    float* slab;               // split-K partials [splits][M][N] fp32 (dense), reduced by k_splitk_reduce
    long long a_bs, b_bs, c_bs; // batched launch: element strides of A, B, C between products
    int batch;                 // number of products in the launch (>= 1)
    int order;                 // 1: locality order of the flattened grid (default); 0: round-1 order (diagnostics)
    unsigned long long* stamp; // measurement hook (e2t_gemm_stamps): [first workgroup start, last workgroup end] of this launch, 100-MHz clock; null otherwise
    int row0;                  // row index of this launch's row 0 within the product it is a part of (the rows of a product may leave
                               // in two launches: gemm_launch); only the row-length mask and the dropout counter see it
};
// product z of a batched launch: operands, output and slabs moved to that product's
__device__ __forceinline__ GemmArgs gemm_batch_view(const GemmArgs& in, int z) {
    GemmArgs p = in;
    p.A = in.A + (size_t)z * in.a_bs;
    p.B = in.B + (size_t)z * in.b_bs;
    p.C = (char*)in.C + (size_t)z * in.c_bs * ((in.flags & E2T_GEMM_OUT_BF16) ? 2 : 4);
    if (in.slab) p.slab = in.slab + (size_t)z * in.splits * in.M * in.N;
    return p;
}

// 512 zero bytes for the k-rows of a K-major tile beyond K: a device global of this code object (zero-initialised when
// the module is loaded), so no entry point ever allocates
__device__ __attribute__((aligned(512))) bf16_t g_gemm_zero_page[256];

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 3) + (chunk ^ (row & 7)); }

// ---- epilogue shared by the GEMM kernel (direct store) and the split-K reduction ----------------------------------
struct EpiCtx {
    bool out_bf16, accum, relu, dodrop, vec_ok;
    int Nst;                          // columns that go to C (N-1 when the last column is diverted to last_col_out)
    unsigned long long dkey; unsigned dthresh; float dkeep;
};
// row m of the output is a real step of its utterance (rows beyond the decimated length are zeroed)
__device__ __forceinline__ bool row_valid(const GemmArgs& p, int gm) {
    if (!p.lens) return true;
    gm += p.row0;
    if (p.rowsG <= 1) return (gm / p.rowsB) < p.lens[gm % p.rowsB];
    const int tgb = gm / p.rowsG, g = gm - tgb * p.rowsG;
    return (tgb / p.rowsB) * p.rowsG + g < p.lens[tgb % p.rowsB];
}
__device__ __forceinline__ EpiCtx epi_ctx(const GemmArgs& p) {
    EpiCtx c;
    c.out_bf16 = p.flags & E2T_GEMM_OUT_BF16;
    c.accum = p.flags & E2T_GEMM_ACCUMULATE;
    c.relu = p.flags & E2T_GEMM_RELU;
    c.dodrop = (p.flags & E2T_GEMM_DROPOUT) && p.drop.rate > 0.f;
    c.Nst = p.last_col_out ? p.N - 1 : p.N;
    c.vec_ok = (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0;
    c.dkey = 0;
    if (c.dodrop) c.dkey = p.drop.seed + (p.drop.step ? (unsigned long long)(*p.drop.step) : 0ull);
    c.dthresh = (unsigned)(p.drop.rate * 16777216.0f);
    c.dkeep = c.dodrop ? 1.0f / (1.0f - p.drop.rate) : 1.0f;
    return c;
}
// Dropout scales of the logical elements e0 .. e0+3 (e0 % 4 != 0) of a row-major tensor whose width is not a multiple of 4 (the
// 225-unit auxiliary layer: three rows out of four): element e takes word e & 3 of the Philox block of counter e >> 2, so the
// four elements straddle TWO consecutive blocks -- two evaluations and a shift instead of one evaluation per element
// (8704 x 225 x 832 with dropout: 42 -> 28 us).  NOT inlined: 16 copies of it in the store loop of a 128 x 128 instance push
// the loop past hipcc's unrolling limit; it then stays rolled and indexes the accumulators dynamically, which puts all 64 of
// them into scratch memory (seen: 288 B of scratch per lane).  A pure function of values.
__device__ __attribute__((noinline)) f32x4 drop_scale4_straddle(unsigned long long e0, unsigned long long key, unsigned stream, unsigned thresh, float keep) {
    const unsigned sh = (unsigned)(e0 & 3ull);
    const unsigned long long c0 = e0 >> 2, c1 = c0 + 1;
    unsigned ra[4], rb[4];
    philox4x32_10((unsigned)c0, (unsigned)(c0 >> 32), stream, 0u, (unsigned)key, (unsigned)(key >> 32), ra);
    philox4x32_10((unsigned)c1, (unsigned)(c1 >> 32), stream, 0u, (unsigned)key, (unsigned)(key >> 32), rb);
    const unsigned w0 = sh == 1 ? ra[1] : (sh == 2 ? ra[2] : ra[3]);
    const unsigned w1 = sh == 1 ? ra[2] : (sh == 2 ? ra[3] : rb[0]);
    const unsigned w2 = sh == 1 ? ra[3] : (sh == 2 ? rb[0] : rb[1]);
    const unsigned w3 = sh == 1 ? rb[0] : (sh == 2 ? rb[1] : rb[2]);
    return (f32x4){((w0 >> 8) >= thresh) ? keep : 0.f, ((w1 >> 8) >= thresh) ? keep : 0.f, ((w2 >> 8) >= thresh) ? keep : 0.f, ((w3 >> 8) >= thresh) ? keep : 0.f};
}
// v: alpha * product (+ bias) of row gm, columns gn0..gn0+3.  RICH = false drops ReLU / mask / dropout.
// epi_apply4: everything between the product and the store (the values stay with the lane that computed them)
template <bool RICH>
__device__ __forceinline__ void epi_apply4(const GemmArgs& p, const EpiCtx& ec, int gm, int gn0, float (&v)[4], bool rowvalid) {
    if (p.last_col_out && gn0 + 3 >= p.N - 1 && gn0 <= p.N - 1) p.last_col_out[gm] = v[p.N - 1 - gn0];
    if (gn0 >= ec.Nst) return;
    const int nv = min(4, ec.Nst - gn0);
    if (RICH && ec.relu) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
    if (RICH && p.mask_src) {
        const bf16_t* ms = p.mask_src + (size_t)gm * p.ld_mask + gn0;
        for (int r = 0; r < nv; ++r) v[r] = (ms[r] & 0x7FFF) != 0 ? v[r] : 0.f;    // kept & active
    }
    if (RICH && ec.dodrop) {
        const unsigned long long e0 = (unsigned long long)(gm + p.row0) * p.ld_logical + gn0;
        if ((e0 & 3ull) == 0) {
            unsigned rr[4];
            const unsigned long long ctr = e0 >> 2;
            philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), p.drop.stream, 0u, (unsigned)ec.dkey, (unsigned)(ec.dkey >> 32), rr);
            for (int r = 0; r < 4; ++r) v[r] *= ((rr[r] >> 8) >= ec.dthresh) ? ec.dkeep : 0.f;
        } else {
            const f32x4 sc = drop_scale4_straddle(e0, ec.dkey, p.drop.stream, ec.dthresh, ec.dkeep);
            v[0] *= sc[0]; v[1] *= sc[1]; v[2] *= sc[2]; v[3] *= sc[3];
        }
    }
    if (!rowvalid) { for (int r = 0; r < 4; ++r) v[r] = 0.f; }
}
// epi_put4: the lane's 4 consecutive columns to C (8 B of bf16 or 16 B of fp32)
__device__ __forceinline__ void epi_put4(const GemmArgs& p, const EpiCtx& ec, int gm, int gn0, const float (&v)[4]) {
    if (gn0 >= ec.Nst) return;
    const int nv = min(4, ec.Nst - gn0);
    if (ec.out_bf16) {
        bf16_t* c = (bf16_t*)p.C + (size_t)gm * p.ldc + gn0;
        if (nv == 4 && (p.ldc & 3) == 0) *(ushort4*)c = make_ushort4(f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3]));
        else for (int r = 0; r < nv; ++r) c[r] = f2bf(v[r]);
    } else {
        float* c = (float*)p.C + (size_t)gm * p.ldc + gn0;
        if (nv == 4 && ec.vec_ok) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (ec.accum) { const float4 old = *(const float4*)c; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            *(float4*)c = o;
        } else {
            for (int r = 0; r < nv; ++r) c[r] = ec.accum ? (c[r] + v[r]) : v[r];
        }
    }
}
// epi_put8_pair: bf16 output of TWO adjacent 16-column sub-tiles.  A lane holds columns fq*4 .. fq*4+3 of both; after two
// v_permlane16_swap per pair of packed words (odd 16-lane rows of the first operand <-> even rows of the second) lane fq holds
// EIGHT consecutive columns -- of the first sub-tile for even fq, of the second for odd fq -- and stores 16 B: the wave writes
// 16 rows x 64 contiguous bytes per instruction (the shape of the fp32 store) instead of two instructions of 16 x 32 B.
// Needs all 32 columns inside the stored range, ldc % 8 == 0 and C 16-B aligned (wave-uniform conditions, checked by the caller).
__device__ __forceinline__ void epi_put8_pair(const GemmArgs& p, int gm, int gn_pair0, int fq, const float (&a)[4], const float (&b)[4]) {
    unsigned a01 = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16), a23 = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
    unsigned b01 = (unsigned)f2bf(b[0]) | ((unsigned)f2bf(b[1]) << 16), b23 = (unsigned)f2bf(b[2]) | ((unsigned)f2bf(b[3]) << 16);
    const auto lo = __builtin_amdgcn_permlane16_swap(a01, b01, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(a23, b23, false, false);
    // lo[0] / hi[0]: rows fq = 0, 2 keep the first sub-tile's own columns, rows 1, 3 got the second sub-tile's columns of fq - 1;
    // lo[1] / hi[1]: rows 0, 2 got the first sub-tile's columns of fq + 1, rows 1, 3 keep the second sub-tile's own
    bf16_t* c = (bf16_t*)p.C + (size_t)gm * p.ldc + gn_pair0 + (fq & 1) * 16 + (fq >> 1) * 8;
    *(uint4*)c = make_uint4(lo[0], hi[0], lo[1], hi[1]);
}
template <bool RICH>
__device__ __forceinline__ void epi_store4(const GemmArgs& p, const EpiCtx& ec, int gm, int gn0, float (&v)[4], bool rowvalid) {
    epi_apply4<RICH>(p, ec, gm, gn0, v, rowvalid);
    epi_put4(p, ec, gm, gn0, v);
}


extern __shared__ __attribute__((aligned(16))) uint4 gemm_smem[];

// LDS bytes of an instance and the workgroups of it that fit a CU (160 KiB of LDS; at most 2 with 64 accumulators and
// the fragments in <= 256 registers per lane, 4 for the 32-deep two-stage diagnostic form)
constexpr int gemm_lds_bytes(int bm, int bn, int kt, int ns) { return ns * (bm + bn) * (kt / 8) * 16; }
constexpr int gemm_wgs_per_cu(int bm, int bn, int kt, int ns) {
    return bm * bn > 128 * 128 ? 1 : (160 * 1024 / gemm_lds_bytes(bm, bn, kt, ns) >= 2 ? 2 : 1);
}

// RICH = false drops the ReLU / mask / dropout epilogue: with 32 accumulator tiles per wave the full epilogue body is too
// large for hipcc to unroll, and a rolled loop indexes the accumulators dynamically (= scratch memory, 4x slower kernel).
// KT = K depth of a stage, NS = stages.  Every shipped instance is 64 x 2.  Measured on the main loop in isolation
// (scripts/gemm_loop_probe.py, two 128 x 128 workgroups per CU, per pair of 64-deep K tiles): loads only 0.73 us, fragment
// reads + MFMAs only 0.77 (MFMA pipe alone: 0.43), both 1.31-1.40 K-major / 1.06 NT with random operands -- but 0.93 / 0.74
// with all-zero operands: the loop runs into the chip's power limit, not into a latency (deeper pipelines 64 x 3, 32 x 4,
// 32 x 3, 64 x 4, and a software pipeline across the barrier, were all slower: they add instructions, not overlap).  What
// pays is fewer bytes and fewer instructions per flop: the locality order of the grid and the scalar-base DMA form below.
// DBG (diagnostics, scripts/gemm_loop_probe.py): 1 = no operand loads after the prologue, 2 = loads and barriers only (no
// fragment reads, no MFMA), 3 = fragment reads without MFMAs, 4 = no stores in the epilogue -- wrong results by design, timing only
template <int BM, int BN, int WM, int WN, bool RICH, bool TN, int KT, int NS, int DBG>
__device__ __forceinline__ void gemm_body(const GemmArgs& p_in, const int L /* workgroup index within this product's launch */) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int STAGE = (BM + BN) * (KT / 8);  // 16-B units per stage: [A: BM rows | B: BN rows][KT/8 chunks]
    static_assert(KT == 64 || (TN && KT == 32), "a 32-deep stage exists for the K-major form only");
    static_assert(NS >= 2 && NS <= 6, "stages");
    constexpr int TI = BM / WM / 16, TJ = BN / WN / 16;
    constexpr int IA = BM / 8 / NW * KT / 64, IB = BN / 8 / NW * KT / 64;      // DMA instructions per wave and operand (1 KiB each)
    uint4* smem = gemm_smem;                     // [NS stages][STAGE], ONE object

    // The grid is ONE dimension over (product of a batched launch, K split, output tile); workgroup L runs on XCD L % 8
    // (cdna_hip_programming.md T1).  Work items are laid out product-major, then split, then tiles with the SHORTER tile
    // dimension innermost, and XCD x takes the x-th contiguous eighth of that list (bijective form): what runs together on
    // one XCD is then one K range of a compact block of tiles, so each operand panel is fetched into that XCD's L2 once and
    // shared by the workgroups that need it.  (Round 1 ordered tiles row-major whatever the shape and let splits and
    // products fall on the XCDs as the 3-D grid happened to: for dW_x, 7 x 25 tiles x 2 splits, every XCD streamed nearly
    // all of the wide operand -- L2 hit rate 0.48, 3.8 x the algorithmic bytes from HBM.)
    const int ntm = (p_in.M + BM - 1) / BM, ntn = (p_in.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int ksplit, tm, tn, zprod;
    {
        const int W = nwg * p_in.splits * p_in.batch;
        const int q = W >> 3, r = W & 7, xcd = L & 7, idx = L >> 3;
        int w = p_in.order ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : L;
        zprod = w / (nwg * p_in.splits);
        w -= zprod * (nwg * p_in.splits);
        ksplit = w / nwg;
        int bid = w - ksplit * nwg;
        if (!p_in.order) {
            const int q2 = nwg >> 3, r2 = nwg & 7, x2 = bid & 7, i2 = bid >> 3;
            bid = (x2 < r2 ? x2 * (q2 + 1) : r2 * (q2 + 1) + (x2 - r2) * q2) + i2;
        }
        if (p_in.order && ntm <= ntn) { tn = bid / ntm; tm = bid - tn * ntm; }
        else { tm = bid / ntn; tn = bid - tm * ntn; }
    }
    const GemmArgs p = gemm_batch_view(p_in, zprod);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WN) * (BM / WM), wn = (wave % WN) * (BN / WN);

    // K range of this split (in full 64-wide tiles; the zero-filled tail belongs to the last split)
    const int nfull = TN ? (p.K + KT - 1) / KT : p.K / BK;
    const bool has_tail = !TN && (p.K % BK) != 0;
    const int per = (nfull + p.splits - 1) / p.splits;
    const int t0 = ksplit * per;
    const int t1 = min(nfull, t0 + per);
    const bool my_tail = has_tail && (ksplit == p.splits - 1);

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // DMA map: wave w, instruction i fills rows (w*I+i)*8 .. +8 of an operand tile;
    // lane -> (row = base + lane/8, physical chunk = lane%8), source chunk = physical ^ (row&7).
    const int drow = lane >> 3, dpc = lane & 7;
    // TN: operands are K-major ([k][m] / [k][n] row-major, e.g. activations and their gradients as the layers wrote
    // them).  A tile is [64 k-rows][BM columns]; one DMA instruction moves 1 KiB = (2048 / BM) k-rows of BM*2 bytes, the
    // 16-B chunk position inside a row is XOR-swizzled with tn_swz(k-row) on the SOURCE side, rows k >= K come from a zero
    // page, columns beyond M/N are clamped (their products are never stored).  Fragments are then gathered with the
    // transposing LDS read (ds_read_b64_tr_b16, lane mapping verified by a one-off probe in round 1 and, ever since, by every K-major GEMM test).
    // The 32 lanes of one read phase take 32 B from each of 8 k-rows (4 consecutive ones of two fq groups, 8 rows apart); a
    // k-row is a multiple of 256 B = the 64-bank window, so the swizzle must send those 8 rows to 8 different 32-B slots,
    // i.e. act on chunk bits 1..3 (SQ_LDS_BANK_CONFLICT: 50 % of the LDS cycles with the swizzle on bits 0..3 -> 0).
    auto tn_swz = [](int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); };
    // one DMA instruction (1 KiB) of tile t into stage buf: pieces 0..IA-1 belong to the A tile, IA..IA+IB-1 to the B tile
    constexpr int NP = IA + IB;
    // source address of this lane's 16 B of piece pc_ of tile t, and the LDS unit its wave-instruction starts at
    auto piece_addr = [&](int t, int buf, int pc_, const bf16_t*& src, uint4*& dst) {
        const int k0 = t * KT;
        uint4* sa = smem + buf * STAGE;
        uint4* sb = sa + BM * (KT / 8);
        const bool isA = pc_ < IA;
        const int i = isA ? pc_ : pc_ - IA;
        if (TN) {
            constexpr int CA = BM / 8, CB = BN / 8;                // 16-B chunks per k-row
            constexpr int RA = 64 / CA, RB_ = 64 / CB;             // k-rows per DMA instruction
            if (isA) {
                const int rb = (wave * IA + i) * RA, r = rb + lane / CA, pc = lane % CA;
                const int c = pc ^ (tn_swz(r) & (CA - 1));
                const int gm = min(m0 + c * 8, (p.M - 1) & ~7);    // 8-column chunks; lda % 8 == 0 keeps them in the row
                src = (k0 + r < p.K) ? p.A + (size_t)(k0 + r) * p.lda + gm : g_gemm_zero_page;
                dst = sa + rb * CA;
            } else {
                const int rb = (wave * IB + i) * RB_, r = rb + lane / CB, pc = lane % CB;
                const int c = pc ^ (tn_swz(r) & (CB - 1));
                const int gn = min(n0 + c * 8, (p.N - 1) & ~7);
                src = (k0 + r < p.K) ? p.B + (size_t)(k0 + r) * p.ldb + gn : g_gemm_zero_page;
                dst = sb + rb * CB;
            }
            return;
        }
        if (isA) {
            const int rb = (wave * IA + i) * 8;
            const int r = rb + drow;
            const int gm = min(m0 + r, p.M - 1);
            src = p.A + (size_t)gm * p.lda + k0 + (dpc ^ (r & 7)) * 8;
            dst = sa + rb * 8;
        } else {
            const int rb = (wave * IB + i) * 8;
            const int r = rb + drow;
            const int gn = min(n0 + r, p.N - 1);
            src = p.B + (size_t)gn * p.ldb + k0 + (dpc ^ (r & 7)) * 8;
            dst = sb + rb * 8;
        }
    };
    // K-major fast path: for a tile that lies wholly inside K the lane's source is (uniform tile base) + (a byte offset that
    // does not depend on the tile) -- the offsets are computed once, the base lives in scalar registers, and the DMA
    // instruction takes them as they are (the general form spent ~20 instructions per piece on 64-bit multiplies, the
    // K-tail select and the zero page's address)
    unsigned tn_off[TN ? NP : 1];
    if (TN) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const bf16_t* src; uint4* dst;
            piece_addr(0, 0, q, src, dst);              // (rows of tile 0 beyond K give the zero page here; such tiles take the general form)
            tn_off[q] = (unsigned)((const char*)src - (const char*)(q < IA ? p.A : p.B));
        }
    }
    auto issue_piece = [&](int t, int buf, int pc_) {
        const bf16_t* src; uint4* dst;
        if (TN && (t + 1) * KT <= p.K) {
            uint4* sa = smem + buf * STAGE;
            uint4* sb = sa + BM * (KT / 8);
            constexpr int CA = BM / 8, CB = BN / 8, RA = 64 / CA, RB_ = 64 / CB;
            const bool isA = pc_ < IA;
            dst = isA ? sa + (wave * IA + pc_) * RA * CA : sb + (wave * IB + (pc_ - IA)) * RB_ * CB;
            const bf16_t* base = isA ? p.A + (size_t)t * KT * p.lda : p.B + (size_t)t * KT * p.ldb;
            dma16_to_lds_sbase(base, tn_off[pc_], lds_addr_of(dst));
            return;
        }
        piece_addr(t, buf, pc_, src, dst);
        dma16_to_lds(src, lds_addr_of(dst));
    };
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(t, buf, q);
    };
    const int frow = lane & 15, fq = lane >> 4;
    typedef short v4s16 __attribute__((ext_vector_type(4)));
    // nt >= 0: the DMA pieces of tile nt (into the other stage) are issued between the MFMA rows of the first K block, so
    // their issue cost (m0 set-up, address VALU, 60-180 cycles of issue each) overlaps MFMA execution instead of
    // preceding it
    auto compute = [&](int buf, int nt, int nbuf) {
        const uint4* sa = smem + buf * STAGE;
        const uint4* sb = sa + BM * (KT / 8);
        // 128 x 128 instances: the fragments of BOTH K blocks are requested before the first MFMA (registers to spare), so
        // the LDS latency is exposed once per tile instead of once per K block
        constexpr bool BOTH = (BM * BN <= 128 * 128) && !TN;      // (K-major operands: measured 4 % slower this way)
        bf16x8 fa2[2][TI], fb2[2][TJ];
#pragma unroll
        for (int kb = 0; kb < KT / 32; ++kb) {
            bf16x8 (&fa)[TI] = fa2[kb], (&fb)[TJ] = fb2[kb];
            if (TN) {
                // lane (frow = l&15, fq = l>>4) needs k = kb*32 + fq*8 .. +7 of column (tile offset + frow): two transposing
                // reads of a [4 k][16 columns] block; within the 16-lane group lane j points at k-row j>>2, columns (j&3)*4..
                constexpr int CA = BM / 8, CB = BN / 8;
                const int jr = frow >> 2, jc = frow & 3;
                v4s16 va[2][TI], vb[2][TJ];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = kb * 32 + fq * 8 + h * 4 + jr;
                    const int sw = tn_swz(row);
                    const char* ra = (const char*)sa + (size_t)row * (BM * 2);
                    const char* rb = (const char*)sb + (size_t)row * (BN * 2);
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        const int chunk = ((wm + i * 16) >> 3) + (jc >> 1);
                        va[h][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(ra + ((chunk ^ (sw & (CA - 1))) << 4) + (jc & 1) * 8));
                    }
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const int chunk = ((wn + j * 16) >> 3) + (jc >> 1);
                        vb[h][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(rb + ((chunk ^ (sw & (CB - 1))) << 4) + (jc & 1) * 8));
                    }
                }
                // the two halves are the low / high register pair of the fragment: a concatenation, no element moves
#pragma unroll
                for (int i = 0; i < TI; ++i) fa[i] = __builtin_shufflevector(va[0][i], va[1][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[j] = __builtin_shufflevector(vb[0][j], vb[1][j], 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                const int ch = kb * 4 + fq;
#pragma unroll
                for (int i = 0; i < TI; ++i) { uint4 va = sa[swz(wm + i * 16 + frow, ch)]; fa[i] = *(bf16x8*)&va; }
#pragma unroll
                for (int j = 0; j < TJ; ++j) { uint4 vb = sb[swz(wn + j * 16 + frow, ch)]; fb[j] = *(bf16x8*)&vb; }
            }
            if (BOTH) continue;
            if (TN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                if (nt >= 0) {
                    constexpr int PER = (NP + TI - 1) / TI;        // pieces per row of the first K block
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        if (kb == 0 && i * PER + q < NP) issue_piece(nt, nbuf, i * PER + q);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if (DBG == 3) { asm volatile("" :: "v"(fb[j]), "v"(fa[i])); continue; }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (BOTH) {
            __builtin_amdgcn_sched_barrier(0);       // keep the requests above the MFMAs (hipcc sinks them back otherwise)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb2[kb][j], fa2[kb][i], acc[i][j], 0, 0, 0);
        }
    };

    // A wave whose whole sub-tile lies beyond M or N (the ragged edge of a product: dW_h is 400 x 1600 -- 3.125 x 12.5 tiles of
    // 128, dW_x 801 x 3200 -- 6.26 tile rows) stages its share of the operands and meets the barriers, but reads no fragment and
    // issues no MFMA: the LDS bandwidth and the MFMA pipe go to the other waves of the CU.
    // (128 x 128 instances only: around the 128 accumulators of a 256 x 256 wave the branch costs hipcc 25 registers and spills.)
    constexpr bool SKIP_DEAD = BM * BN <= 128 * 128;
    const bool wave_dead = SKIP_DEAD && ((m0 + wm >= p.M) || (n0 + wn >= p.N));
    int cur = 0;
    {
    // NS stages: tiles t .. t+NS-2 are in flight while tile t is awaited; tile t+NS-1 goes into the stage tile t-1 was read
    // from (every wave left it before the barrier of this iteration)
    constexpr int NP_ = IA + IB;
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) if (t0 + s_ < t1) issue(t0 + s_, s_);
    for (int t = t0; t < t1; ++t) {
        const int ahead = min(NS - 2, t1 - 1 - t);             // later tiles that may stay in flight (loads retire in order)
        if (NS == 2 || ahead == 0) dma_wait_but<0>();
        else if (ahead == 1) dma_wait_but<NP_>();
        else if (ahead == 2) dma_wait_but<2 * NP_>();
        else if (ahead == 3) dma_wait_but<3 * NP_>();
        else dma_wait_but<4 * NP_>();
        __syncthreads();                                       // tile t has landed for every wave; the stage of tile t-1 is free
        // interleaving pays on the 256x256 instance (one workgroup per CU: nobody else hides the issue phase, 4 % faster);
        // with two workgroups per CU it only delays the loads (TN weight gradients 88 -> 107 us)
        constexpr bool INTERLEAVE = (BM * BN > 128 * 128);
        const int nxt = t + NS - 1;
        int nbuf = cur + NS - 1; if (nbuf >= NS) nbuf -= NS;
        if (DBG != 1 && !INTERLEAVE && nxt < t1) issue(nxt, nbuf);
        if (wave_dead) { if (DBG != 1 && INTERLEAVE && nxt < t1) issue(nxt, nbuf); }
        else if (DBG != 2) compute(cur, (DBG != 1 && INTERLEAVE && nxt < t1) ? nxt : -1, nbuf);
        if (++cur == NS) cur = 0;
    }
    }
    if (my_tail) {
        // register-staged, zero-filled K tail into the same swizzled image
        __syncthreads();
        const int srow = tid >> 3, schunk = tid & 7;
        const int kk = nfull * BK + schunk * 8;
        const bool kin = kk < p.K;
        uint4* sa = smem + cur * STAGE;
        uint4* sb = sa + BM * 8;
#pragma unroll
        for (int i = 0; i < BM / (NT / 8); ++i) {
            const int r = srow + (NT / 8) * i, gm = m0 + r;
            // (`cond ? *p : zero` makes hipcc select between p and a zero kept in SCRATCH and load through flat addressing: the
            //  out-of-range lanes load the operand's first 16 B instead and are masked to zero)
            const bool in = kin && gm < p.M;
            uint4 v = *(const uint4*)(in ? p.A + (size_t)gm * p.lda + kk : p.A);
            const unsigned mk = in ? 0xFFFFFFFFu : 0u;
            sa[swz(r, schunk)] = make_uint4(v.x & mk, v.y & mk, v.z & mk, v.w & mk);
        }
#pragma unroll
        for (int i = 0; i < BN / (NT / 8); ++i) {
            const int r = srow + (NT / 8) * i, gn = n0 + r;
            const bool in = kin && gn < p.N;
            uint4 v = *(const uint4*)(in ? p.B + (size_t)gn * p.ldb + kk : p.B);
            const unsigned mk = in ? 0xFFFFFFFFu : 0u;
            sb[swz(r, schunk)] = make_uint4(v.x & mk, v.y & mk, v.z & mk, v.w & mk);
        }
        __syncthreads();
        if (!wave_dead) compute(cur, -1, 0);
    }

    // epilogue.  MFMA roles are (B-tile fragment, A-tile fragment), so D[i][j]: column j = lane&15 is the
    // M row, rows (lane>>4)*4 + r are FOUR CONSECUTIVE N columns: every lane stores 16 B (fp32) / 8 B (bf16)
    // per sub-tile instead of four scattered words, and bias / mask / dropout are fetched 4 at a time.
    const EpiCtx ec = epi_ctx(p);
    const bool atomic = p.splits > 1;
    // the bias of this lane's 4 columns per column tile: fetched once, not per row tile (a dependent global load per
    // (i, j) in the store loop cost 30 us on the encoder input projection)
    float bias4[TJ][4];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int gn0 = n0 + wn + j * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[j][r] = (p.bias && !atomic && gn0 + r < ec.Nst) ? p.bias[gn0 + r] : 0.f;
    }
    static_assert(TJ % 2 == 0, "the epilogue handles the 16-column sub-tiles in pairs");
    // bf16 outputs go out in pairs of sub-tiles (epi_put8_pair) wherever both lie wholly inside the stored columns
    const bool pair_base = ec.out_bf16 && !atomic && (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int gm = m0 + wm + i * 16 + frow;
        if (gm >= p.M) continue;
        bool rowvalid = true;
        if (!atomic) rowvalid = row_valid(p, gm);
#pragma unroll
        for (int jp = 0; jp < TJ; jp += 2) {
            const int gnp = n0 + wn + jp * 16;                 // first column of the pair (wave-uniform)
            if (gnp >= p.N) continue;
            float v[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[h][r] = acc[i][jp + h][r] * p.alpha;
            if (atomic) {
                // split-K: dense partial slab (all N columns incl. the bias column); summed in fixed order, and run
                // through the same epilogue, by k_splitk_reduce
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int gn0 = gnp + h * 16 + fq * 4;
                    if (gn0 >= p.N) continue;
                    float* c = p.slab + ((size_t)ksplit * p.M + gm) * p.N + gn0;
                    const int nn = min(4, p.N - gn0);
                    if (nn == 4 && (p.N & 3) == 0) *(float4*)c = make_float4(v[h][0], v[h][1], v[h][2], v[h][3]);
                    else for (int r = 0; r < nn; ++r) c[r] = v[h][r];
                }
                continue;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int gn0 = gnp + h * 16 + fq * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[h][r] += bias4[jp + h][r];
                if (gn0 < p.N) epi_apply4<RICH>(p, ec, gm, gn0, v[h], rowvalid);
            }
            if (DBG == 4) { asm volatile("" :: "v"(v[0][0]), "v"(v[0][1]), "v"(v[0][2]), "v"(v[0][3]), "v"(v[1][0]), "v"(v[1][1]), "v"(v[1][2]), "v"(v[1][3])); continue; }
            if (pair_base && gnp + 32 <= ec.Nst) epi_put8_pair(p, gm, gnp, fq, v[0], v[1]);
            else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int gn0 = gnp + h * 16 + fq * 4;
                    if (gn0 < p.N) epi_put4(p, ec, gm, gn0, v[h]);
                }
            }
        }
    }
}

// Measurement hook (bench.py, e2t_gemm_stamps): the earliest workgroup start and the latest workgroup end of a launch on the
// chip-wide 100-MHz clock -- what a kernel trace reports as the launch's duration, readable INSIDE a replayed graph.
__device__ __forceinline__ unsigned long long stamp_begin(const unsigned long long* st) { return (st && threadIdx.x == 0) ? wall_clock64() : 0ull; }
__device__ __forceinline__ void stamp_end(unsigned long long* st, unsigned long long t0) {
    if (st && threadIdx.x == 0) { atomicMin(st, t0); atomicMax(st + 1, (unsigned long long)wall_clock64()); }
}
template <int BM, int BN, int WM, int WN, bool RICH, bool TN, int KT = 64, int NS = 2, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN, gemm_wgs_per_cu(BM, BN, KT, NS)) void k_gemm_nt(GemmArgs p_in) {
    const unsigned long long t0 = stamp_begin(p_in.stamp);
    gemm_body<BM, BN, WM, WN, RICH, TN, KT, NS, DBG>(p_in, blockIdx.x);
    stamp_end(p_in.stamp, t0);
}

// Several K-major products in ONE launch (the weight gradients of a backward stage: dW_x and dW_h of a layer; projection,
// embedding-side and recurrent kernel of the decoder).  Launched one by one every product pays its own ramp-up, its own last
// partly filled round of workgroups (split counts had to keep a launch inside the 512 resident slots: 175 tiles x 2 = 350 of
// 512) and its own dependent launch gap; here the workgroups of all products form one list that the dispatcher deals out as
// slots free up, so the K splits can be chosen for TWO OR MORE full rounds (e2t_gemm_tn_group_bf16) and a product's tail is
// filled by the next product's head.  Product i owns workgroups [first[i], first[i+1]) (multiples of 8, so that the
// XCD-locality order inside a product still sees workgroup L on XCD L % 8; the few padding workgroups exit at once).
#define E2T_GEMM_GROUP_MAX 8
struct GemmGroupArgs { int n; int first[E2T_GEMM_GROUP_MAX + 1]; int count[E2T_GEMM_GROUP_MAX]; GemmArgs p[E2T_GEMM_GROUP_MAX]; unsigned long long* stamp; };
__global__ __launch_bounds__(256, 2) void k_gemm_tn_group(GemmGroupArgs g) {
    const unsigned long long t0 = stamp_begin(g.stamp);
    int i = 0;
#pragma unroll
    for (int j = 1; j < E2T_GEMM_GROUP_MAX; ++j) if (j < g.n && (int)blockIdx.x >= g.first[j]) i = j;
    const int L = (int)blockIdx.x - g.first[i];
    if (L >= g.count[i]) return;
    gemm_body<128, 128, 2, 2, false, true, 64, 2, 0>(g.p[i], L);
    stamp_end(g.stamp, t0);
}

// C[m][n] (+)= epilogue(sum_s slab[s][m][n]) in fixed split order (deterministic); 4 consecutive columns per thread so
// the dropout mask is the GEMM kernel's (one Philox counter per aligned group of 4).
// A thread keeps EIGHT slab loads (16 B each) in flight whatever the split count: with 2 splits it reduces four groups of
// columns, with 3-4 splits two, beyond that one.  (One group per thread left a 2-split reduction with two loads in flight per
// lane; next to a BPTT kernel that holds 456 registers a CU has room for ONE such wave per SIMD, and cfg4's reductions --
// 100 MB each -- ran at 0.8 TB/s: 117-150 us on the weight-gradient branch that ends the step.)  Group j of thread i is
// element i + j * stride, stride = splitk_reduce_stride(): every access of a wave stays contiguous.
__host__ __device__ inline size_t splitk_reduce_stride_g(int M, int N, int groups) {
    const size_t total = (size_t)M * ((N + 3) >> 2), G = (size_t)groups;
    return ((total + G - 1) / G + 255) / 256 * 256;                // threads (= 256 x workgroups) the reduction of one product needs
}
// PLAIN: nothing between the sum and a 16-B fp32 store (the weight gradients); else the full epilogue of the GEMM kernel
template <int G, bool PLAIN> __device__ __forceinline__ void splitk_reduce_body(const GemmArgs& p_in, int z);
__host__ __device__ inline bool splitk_reduce_plain(const GemmArgs& p) {
    return !p.bias && !p.lens && !p.mask_src && !p.last_col_out && (p.flags & ~(E2T_GEMM_SPLITK | E2T_GEMM_KEEP_SLABS)) == 0 && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0;
}
__host__ __device__ inline int splitk_reduce_groups(const GemmArgs& p) { return !splitk_reduce_plain(p) ? 1 : p.splits <= 2 ? 4 : p.splits <= 4 ? 2 : 1; }
__device__ __forceinline__ void splitk_reduce_any(const GemmArgs& p, int z) {
    if (!splitk_reduce_plain(p)) splitk_reduce_body<1, false>(p, z);
    else if (p.splits <= 2) splitk_reduce_body<4, true>(p, z);
    else if (p.splits <= 4) splitk_reduce_body<2, true>(p, z);
    else splitk_reduce_body<1, true>(p, z);
}
// (56 registers: what a CU has left per SIMD lane beside a 456-register BPTT workgroup of lstm_big.hip)
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmArgs p_in) { splitk_reduce_any(p_in, blockIdx.y); }
// the reductions of a grouped launch: blockIdx.y = product, blockIdx.z = batch member
__global__ __launch_bounds__(256) void k_splitk_reduce_group(GemmGroupArgs g) {
    if ((int)blockIdx.y >= g.n) return;
    const GemmArgs& p = g.p[blockIdx.y];
    if (p.splits <= 1 || (p.flags & E2T_GEMM_KEEP_SLABS) || (int)blockIdx.z >= p.batch) return;
    splitk_reduce_any(p, blockIdx.z);
}
template <int G, bool PLAIN>
__device__ __forceinline__ void splitk_reduce_body(const GemmArgs& p_in, int z) {
    constexpr int MS = 8 / G;                                       // slabs in flight per group
    const GemmArgs p = gemm_batch_view(p_in, z);
    const int N4 = (p.N + 3) >> 2;
    const size_t total = (size_t)p.M * N4, stride = splitk_reduce_stride_g(p.M, p.N, G);
    const size_t idx0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx0 >= stride) return;
    const size_t sstride = (size_t)p.M * p.N;
    if constexpr (PLAIN) {
        // (a slab and C are < 4 GB: uniform bases + 32-bit lane offsets, and nothing else kept per group -- the kernel has to fit
        //  the 56 registers a CU has left per SIMD lane beside a 456-register BPTT workgroup)
        unsigned off[G], coff[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const size_t idx = idx0 + (size_t)j * stride;
            const bool ok = idx < total;
            const int gm = ok ? (int)(idx / N4) : 0, gn0 = ok ? (int)(idx - (size_t)gm * N4) * 4 : 0;
            off[j] = ok ? (unsigned)(((size_t)gm * p.N + gn0) * sizeof(float)) : 0xFFFFFFFFu;
            coff[j] = (unsigned)(((size_t)gm * p.ldc + gn0) * sizeof(float));
        }
        float4 v[G];
        for (int s = 0; s < p.splits; s += MS) {
            float4 t[G][MS];
#pragma unroll
            for (int u = 0; u < MS; ++u) {
                const char* sb = (const char*)(p.slab + (size_t)min(s + u, p.splits - 1) * sstride);      // (a slab beyond the last: re-read, not added)
#pragma unroll
                for (int j = 0; j < G; ++j) t[j][u] = *(const float4*)(sb + (off[j] != 0xFFFFFFFFu ? off[j] : 0u));
            }
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int u = 0; u < MS; ++u) {
                    if (s == 0 && u == 0) { v[j] = t[j][0]; continue; }
                    if (s + u < p.splits) { v[j].x += t[j][u].x; v[j].y += t[j][u].y; v[j].z += t[j][u].z; v[j].w += t[j][u].w; }
                }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) if (off[j] != 0xFFFFFFFFu) *(float4*)((char*)p.C + coff[j]) = v[j];
        return;
    } else {
        static_assert(G == 1, "the full epilogue takes one group per thread");
        if (idx0 >= total) return;
        const int gm = (int)(idx0 / N4), gn0 = (int)(idx0 - (size_t)gm * N4) * 4;
        const int nn = min(4, p.N - gn0);
        const EpiCtx ec = epi_ctx(p);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool vec = (p.N & 3) == 0;
        const float* c0 = p.slab + (size_t)gm * p.N + gn0;
        auto ld4 = [&](int s) {
            const float* c = c0 + (size_t)s * sstride;
            if (vec) return *(const float4*)c;
            return make_float4(c[0], nn > 1 ? c[1] : 0.f, nn > 2 ? c[2] : 0.f, nn > 3 ? c[3] : 0.f);
        };
        // 8 slabs in flight at a time (a dependent chain of 22 loads cost 36 us on a 3-block reduction); the sum keeps the split order
        for (int s = 0; s < p.splits; s += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (s + u < p.splits) ? ld4(s + u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (s + u < p.splits) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
        }
        if (p.bias) for (int r = 0; r < nn; ++r) if (gn0 + r < ec.Nst) v[r] += p.bias[gn0 + r];
        epi_store4<true>(p, ec, gm, gn0, v, row_valid(p, gm));
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// a5 + a6 fused: tf.reverse_sequence (trainers.py:808-810) + strided temporal convolution (_convolve_sequences,
// trainers.py:813-818) straight from the fp32 electrode grid x [B][T][C] -- ONE pass over the input.
//   E[m][n] = epilogue( sum_k bf16(x[b][len_b - 1 - (t'*N + w)][c]) * W[n][k] ),  m = t'*B + b,  k = w*C + c
// HBM-bound by design (config 5: 2.1 GB of input against 134 GFLOP), so both forms below are built around the input stream:
//  * a workgroup owns 64 output rows x all F <= 128 columns; a row's operand is ONE contiguous run of N*C floats of x
//    (N consecutive samples, walked backwards in time), so per K step of 64 floats every row contributes a 256-byte run;
//  * samples beyond an utterance's length come from the zero page; the rounding to bf16 is e2t_conv_pack's (hardware
//    converter, bit-identical); with a_out != NULL the rounded operand is also written to the packed im2row copy A[m][k] the
//    weight-gradient product of the backward pass reads (training: 2.1 GB in + 1.05 GB out instead of 2.1 + 1.05 + 1.05);
//  * K can be cut into `splits` ranges (grid = tiles x splits: fills the last round of workgroups); partial sums then go to
//    fp32 slabs and k_splitk_reduce applies the epilogue, as for the GEMM.
// k_conv_fwd (all-DMA form): raw fp32 runs AND the [128][KS] weight slice by LDS-DMA, NS stages; a lane reads its row's 8
//    floats (chunk index XOR-swizzled with the row on the source side of the DMA) and rounds them in registers.  LDS holds
//    fp32 x plus weights, which caps the bytes of x in flight at 48 KiB per CU: cfg5 507 us (4.1 TB/s) whatever the shape
//    (64x2 ... 64x5, 32x2 ... 32x5, 1-3 splits, tiles of 64 utterances or of 64 steps of one utterance: 480-580 us).
// k_conv_fwd_ws (wave-specialised form, the default): see below; cfg5 445 us (4.7 TB/s).
// For scale: plain loads of the same access pattern with nothing else in the kernel read at 6.0-6.1 TB/s = 340 us
// (a one-off streaming probe, round 2); the two-kernel path (e2t_conv_pack + GEMM) takes 894 us.
// Epilogue = the GEMM's (bias, ReLU, dropout, rows beyond an utterance's decimated length zeroed).
// ---------------------------------------------------------------------------------------------------------------------
struct ConvFwdArgs {
    const float* x; const int* lens; const bf16_t* WT; int ldw;
    int B, T, C, N, M, F;
    bf16_t* a_out; int lda_out;
    int ub, S;                  // tile = (64 / ub) consecutive decimated steps x ub consecutive utterances
    GemmArgs epi;               // C = E, ldc, N = F, flags, drop, ld_logical, lens (decimated) / rowsB, splits, slab
};

constexpr int conv_lds_bytes(int ks, int ns) { return ns * (64 * ks * 4 + 128 * ks * 2); }
constexpr int conv_wgs_per_cu(int ks, int ns) { return 160 * 1024 / conv_lds_bytes(ks, ns) >= 5 ? 5 : 160 * 1024 / conv_lds_bytes(ks, ns); }

template <int KS, int NS>
__global__ __launch_bounds__(256, conv_wgs_per_cu(KS, NS)) void k_conv_fwd(ConvFwdArgs a) {
    constexpr int BMc = 64, BNc = 128;
    constexpr int A_U = BMc * KS * 4 / 16, W_U = BNc * KS * 2 / 16, STAGE_U = A_U + W_U;      // 16-B units
    constexpr int CHA = KS * 4 / 16, RPA = 64 / CHA;          // chunks per fp32 row; rows per 1-KiB DMA piece
    constexpr int CHW = KS * 2 / 16, RPW = 64 / CHW;
    constexpr int PA = BMc / RPA / 4, PW = BNc / RPW / 4;     // pieces per wave and stage
    constexpr int NPc = PA + PW;
    static_assert(KS == 64 || KS == 32, "K step");
    uint4* smem = gemm_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GemmArgs& p = a.epi;
    const int tile = blockIdx.x / p.splits, ksplit = blockIdx.x - tile * p.splits;
    const int tb = (a.B + a.ub - 1) / a.ub;                       // tiles along the utterances
    const int tp0 = (tile / tb) * (BMc / a.ub), b0 = (tile - (tile / tb) * tb) * a.ub;
    // row r of the tile -> output row m = t' * B + b, or -1 outside the batch
    auto row_of = [&](int r) { const int tp = tp0 + r / a.ub, b = b0 + r % a.ub; return (tp < a.S && b < a.B) ? tp * a.B + b : -1; };
    const int K = a.N * a.C, nk = K / KS;
    const int per = (nk + p.splits - 1) / p.splits;
    const int t0 = ksplit * per, t1 = min(nk, t0 + per);

    // per piece of the input tile: this lane's row, the start of that row's run (tap w = 0, channel 0) and how many taps exist
    const float* rowbase[PA];
    int ntaps[PA];
    unsigned srcoff[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int r = (wave * PA + i) * RPA + lane / CHA, pc = lane % CHA;
        const int m = row_of(r);
        const int tp = max(m, 0) / a.B, b = max(m, 0) - tp * a.B;
        const int len = m >= 0 ? a.lens[b] : 0;
        ntaps[i] = len - tp * a.N;                              // tap w is a real sample iff w < ntaps
        rowbase[i] = a.x + ((size_t)b * a.T + (len - 1 - tp * a.N)) * a.C;
        srcoff[i] = (unsigned)((pc ^ (r & (CHA - 1))) * 4);      // in floats
    }
    auto issue = [&](int t, int buf) {
        const int k0 = t * KS, w = k0 / a.C, c0 = k0 - w * a.C;
        uint4* sa = smem + buf * STAGE_U;
        uint4* sw = sa + A_U;
        const long back = (long)c0 - (long)w * a.C;              // tap w lies w samples EARLIER in memory
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const float* src = (w < ntaps[i]) ? rowbase[i] + back + srcoff[i] : (const float*)g_gemm_zero_page + srcoff[i];
            dma16_to_lds(src, lds_addr_of(sa + (wave * PA + i) * 64));
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int r = (wave * PW + i) * RPW + lane / CHW, pc = lane % CHW;
            const int sc = (KS == 64) ? (pc ^ (r & 7)) : (pc ^ ((r >> 2) & 3));
            const bf16_t* src = a.WT + (size_t)min(r, a.F - 1) * a.ldw + k0 + sc * 8;
            dma16_to_lds(src, lds_addr_of(sw + (wave * PW + i) * 64));
        }
    };
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    const int arow = wave * 16 + frow;
    const int gm_mine = row_of(arow);
    bf16_t* aout = (a.a_out && gm_mine >= 0) ? a.a_out + (size_t)gm_mine * a.lda_out : nullptr;
    // every LDS read of the step is requested before its first MFMA (left to itself hipcc issues read, wait, MFMA sixteen
    // times over: ~1.5 us of exposed LDS latency per step and wave, which -- not HBM -- bounded the kernel at 4.3 TB/s)
    auto compute = [&](int t, int buf) {
        const uint4* sa = smem + buf * STAGE_U;
        const uint4* sw = sa + A_U;
        constexpr int NKB = KS / 32;
        uint4 lo[NKB], hi[NKB], fb[NKB][8];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int c = kb * 8 + fq * 2;
            lo[kb] = sa[arow * CHA + (c ^ (arow & (CHA - 1)))];
            hi[kb] = sa[arow * CHA + ((c + 1) ^ (arow & (CHA - 1)))];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = j * 16 + frow, ch = kb * 4 + fq;
                fb[kb][j] = sw[n * CHW + ((KS == 64) ? (ch ^ (n & 7)) : (ch ^ ((n >> 2) & 3)))];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            uint4 fa;
            fa.x = f2bf_pk(__uint_as_float(lo[kb].x), __uint_as_float(lo[kb].y));
            fa.y = f2bf_pk(__uint_as_float(lo[kb].z), __uint_as_float(lo[kb].w));
            fa.z = f2bf_pk(__uint_as_float(hi[kb].x), __uint_as_float(hi[kb].y));
            fa.w = f2bf_pk(__uint_as_float(hi[kb].z), __uint_as_float(hi[kb].w));
            if (aout) *(uint4*)(aout + t * KS + kb * 32 + fq * 8) = fa;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&fb[kb][j], *(const bf16x8*)&fa, acc[j], 0, 0, 0);
        }
    };
    int cur = 0;
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) if (t0 + s_ < t1) issue(t0 + s_, s_);
    for (int t = t0; t < t1; ++t) {
        const int ahead = min(NS - 2, t1 - 1 - t);
        if (ahead == 0) dma_wait_but<0>();
        else if (ahead == 1) dma_wait_but<NPc>();
        else if (ahead == 2) dma_wait_but<2 * NPc>();
        else if (ahead == 3) dma_wait_but<3 * NPc>();
        else dma_wait_but<4 * NPc>();
        __syncthreads();
        const int nxt = t + NS - 1;
        int nbuf = cur + NS - 1; if (nbuf >= NS) nbuf -= NS;
        if (nxt < t1) issue(nxt, nbuf);
        compute(t, cur);
        if (++cur == NS) cur = 0;
    }
    // epilogue: D[i][j]: column j = lane&15 is the output row, rows (lane>>4)*4 + r are four consecutive channels
    const EpiCtx ec = epi_ctx(p);
    const int gm = gm_mine;
    if (gm < 0) return;
    if (p.splits > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gn0 = j * 16 + fq * 4;
            if (gn0 >= p.N) continue;
            float* c = p.slab + ((size_t)ksplit * p.M + gm) * p.N + gn0;
            const int nn = min(4, p.N - gn0);
            if (nn == 4 && (p.N & 3) == 0) *(float4*)c = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            else for (int r = 0; r < nn; ++r) c[r] = acc[j][r];
        }
        return;
    }
    const bool rowvalid = row_valid(p, gm);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gn0 = j * 16 + fq * 4;
        if (gn0 >= a.F) continue;
        float v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v[rr] = acc[j][rr] + ((p.bias && gn0 + rr < a.F) ? p.bias[gn0 + rr] : 0.f);
        epi_store4<true>(p, ec, gm, gn0, v, rowvalid);
    }
}

// The same product with the waves of a workgroup SPECIALISED (512 threads, two workgroups per CU):
//  * waves 0-3 load: each owns 16 of the 64 rows and keeps D K steps of its rows' fp32 runs in flight IN REGISTERS (plain
//    16-B loads, 4 per lane and step), rounds a step to bf16 when it has arrived, writes it into the LDS stage (and the
//    packed copy A_out) and reloads the registers with step t+D.  The DMA form above holds the raw fp32 AND the weight
//    slice in LDS, which caps the input bytes in flight at 48 KiB per CU (4.3 TB/s); here a CU holds 2 x 4 x D x 4 KiB
//    (D = 6: 192 KiB) and LDS only carries bf16.
//  * waves 4-7 multiply: weight slices by LDS-DMA one step ahead, A fragments with ds_read_b128 (NT image, as in the GEMM).
// One s_barrier per K step couples the two halves: step t is written while step t-1 is multiplied.  Raw s_barrier (not
// __syncthreads: its fence would drain the loaders' vmcnt and with it the prefetch).
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int D>
__global__ __launch_bounds__(512, 2) void k_conv_fwd_ws(ConvFwdArgs a) {
    constexpr int BMc = 64, BNc = 128, KS = 64, NSL = 3;
    constexpr int A_U = BMc * 8, W_U = BNc * 8, STAGE_U = A_U + W_U;       // 16-B units per stage (24 KiB)
    uint4* smem = gemm_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GemmArgs& p = a.epi;
    const int tile = blockIdx.x / p.splits, ksplit = blockIdx.x - tile * p.splits;
    const int tb = (a.B + a.ub - 1) / a.ub;
    const int tp0 = (tile / tb) * (BMc / a.ub), b0 = (tile - (tile / tb) * tb) * a.ub;
    auto row_of = [&](int r) { const int tp = tp0 + r / a.ub, b = b0 + r % a.ub; return (tp < a.S && b < a.B) ? tp * a.B + b : -1; };
    const int K = a.N * a.C, nk = K / KS;
    const int per = (nk + p.splits - 1) / p.splits;
    const int t0 = ksplit * per, t1 = min(nk, t0 + per);
    if (wave < 4) {
        // ---------------- loaders ----------------
        // load instruction i of a step covers rows 4i .. 4i+3 of this wave's 16, 16 lanes x 16 B = one whole 256-B run per row
        // (a lane reading 64 contiguous bytes with four instructions would touch every 64-B sector four times)
        const int rg = lane >> 4, c = lane & 15;
        const float* rowbase[4];
        bf16_t* aout[4];
        int ntaps[4], rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 16 + 4 * i + rg;
            const int m = row_of(r);
            const int tp = max(m, 0) / a.B, b = max(m, 0) - tp * a.B;
            const int len = m >= 0 ? a.lens[b] : 0;
            rows[i] = r;
            ntaps[i] = len - tp * a.N;
            rowbase[i] = a.x + ((size_t)b * a.T + (len - 1 - tp * a.N)) * a.C + c * 4;
            aout[i] = (a.a_out && m >= 0) ? a.a_out + (size_t)m * a.lda_out + c * 4 : nullptr;
        }
        // The loads are inline asm and the waits are counted by hand: hipcc, seeing loads behind uniform and divergent branches,
        // put s_waitcnt vmcnt(0) in front of every conversion (= one step in flight).  Every sub-iteration therefore issues
        // EXACTLY four loads, whatever the row or the step: samples beyond an utterance's length, rows outside the batch and the
        // steps past the end of the K range read the zero page instead.  "At most 4 (D-1) younger operations outstanding" then
        // means "the oldest group has arrived" (the A_out stores in between only make the wait stricter).
        f32x4 R[D][4];
        auto load = [&](int t, f32x4 (&dst)[4]) {
            const int k0 = t * KS, w = k0 / a.C, c0 = k0 - w * a.C;
            const long off = (long)c0 - (long)w * a.C;
            const float* s0 = (t < t1 && w < ntaps[0]) ? rowbase[0] + off : (const float*)g_gemm_zero_page + c * 4;
            const float* s1 = (t < t1 && w < ntaps[1]) ? rowbase[1] + off : (const float*)g_gemm_zero_page + c * 4;
            const float* s2 = (t < t1 && w < ntaps[2]) ? rowbase[2] + off : (const float*)g_gemm_zero_page + c * 4;
            const float* s3 = (t < t1 && w < ntaps[3]) ? rowbase[3] + off : (const float*)g_gemm_zero_page + c * 4;
            asm volatile("global_load_dwordx4 %0, %4, off nt\n\tglobal_load_dwordx4 %1, %5, off nt\n\t"
                         "global_load_dwordx4 %2, %6, off nt\n\tglobal_load_dwordx4 %3, %7, off nt"
                         : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(s0), "v"(s1), "v"(s2), "v"(s3) : "memory");
        };
        auto arrived = [&](f32x4 (&dst)[4]) {
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]) : "n"(4 * (D - 1)) : "memory");
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load(t0 + d, R[d]);
        for (int base = t0; base <= t1; base += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int tt = base + d;
                if (tt > t1) break;
                if (tt < t1) {
                    char* sa = (char*)(smem + (tt % NSL) * STAGE_U);
                    arrived(R[d]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint2 o;
                        o.x = f2bf_pk(R[d][i][0], R[d][i][1]); o.y = f2bf_pk(R[d][i][2], R[d][i][3]);
                        *(uint2*)(sa + swz(rows[i], c >> 1) * 16 + (c & 1) * 8) = o;
                        if (aout[i]) *(uint2*)(aout[i] + tt * KS) = o;
                    }
                    load(tt + D, R[d]);
                }
                wg_barrier_lds();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the padding loads of the last steps still target this wave's registers)
        return;
    }
    // ---------------- multipliers ----------------
    const int cw = wave - 4;
    auto issue_w = [&](int t) {
        uint4* sw = smem + (t % NSL) * STAGE_U + A_U;
        const int k0 = t * KS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = (cw * 4 + i) * 8 + (lane >> 3), pc = lane & 7;
            const bf16_t* src = a.WT + (size_t)min(rr, a.F - 1) * a.ldw + k0 + (pc ^ (rr & 7)) * 8;
            dma16_to_lds(src, lds_addr_of(sw + (cw * 4 + i) * 64));
        }
    };
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    const int arow = cw * 16 + frow;
    auto compute = [&](int t) {
        const uint4* sa = smem + (t % NSL) * STAGE_U;
        const uint4* sw = sa + A_U;
        uint4 fa[2], fb[2][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            fa[kb] = sa[swz(arow, kb * 4 + fq)];
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[kb][j] = sw[swz(j * 16 + frow, kb * 4 + fq)];
        }
        __builtin_amdgcn_sched_barrier(0);                      // all 18 reads requested before the first MFMA
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&fb[kb][j], *(const bf16x8*)&fa[kb], acc[j], 0, 0, 0);
    };
    if (t0 < t1) issue_w(t0);
    for (int tt = t0; tt <= t1; ++tt) {
        if (tt + 1 < t1) issue_w(tt + 1);
        if (tt > t0) compute(tt - 1);
        if (tt + 1 < t1) dma_wait_but<4>(); else dma_wait_but<0>();      // the slice of step tt has landed (that of tt+1 may be on its way)
        wg_barrier_lds();
    }
    const EpiCtx ec = epi_ctx(p);
    const int gm = row_of(arow);
    if (gm < 0) return;
    if (p.splits > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gn0 = j * 16 + fq * 4;
            if (gn0 >= p.N) continue;
            float* c = p.slab + ((size_t)ksplit * p.M + gm) * p.N + gn0;
            const int nn = min(4, p.N - gn0);
            if (nn == 4 && (p.N & 3) == 0) *(float4*)c = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            else for (int r = 0; r < nn; ++r) c[r] = acc[j][r];
        }
        return;
    }
    const bool rowvalid = row_valid(p, gm);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gn0 = j * 16 + fq * 4;
        if (gn0 >= a.F) continue;
        float v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v[rr] = acc[j][rr] + ((p.bias && gn0 + rr < a.F) ? p.bias[gn0 + rr] : 0.f);
        epi_store4<true>(p, ec, gm, gn0, v, rowvalid);
    }
}

__global__ __launch_bounds__(256) void k_splitk_reduce(GemmArgs p_in);

extern "C" int e2t_conv_fwd_fused_ok(int C, int F) { return (C % 64 == 0 && F >= 1 && F <= 128) ? 1 : 0; }

extern "C" int e2t_conv_fwd_fused(const float* x, const int32_t* lens, int B, int T, int C, int N, const void* WT, int ldw,
                                  void* E, int lde, int F, void* A_out, int lda_out, const e2t_gemm_epilogue* ep, void* stream) {
    E2T_CHECK_ARG(x && lens && WT && E && ep);
    E2T_CHECK_ARG(B > 0 && T > 0 && N > 0 && e2t_conv_fwd_fused_ok(C, F) && ldw % 8 == 0 && ldw >= N * C && lde >= F);
    E2T_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)WT) & 15) == 0 && (ep->flags & E2T_GEMM_OUT_BF16));
    E2T_CHECK_ARG(!A_out || (lda_out % 8 == 0 && lda_out >= N * C && (((uintptr_t)A_out) & 15) == 0));
    ConvFwdArgs a{};
    a.x = x; a.lens = lens; a.WT = (const bf16_t*)WT; a.ldw = ldw;
    a.B = B; a.T = T; a.C = C; a.N = N; a.F = F;
    a.a_out = (bf16_t*)A_out; a.lda_out = lda_out;
    const int S = (T + N - 1) / N;
    a.M = S * B;
    GemmArgs& p = a.epi;
    p.C = E; p.ldc = lde; p.M = a.M; p.N = F; p.K = N * C; p.alpha = 1.0f; p.splits = 1; p.batch = 1; p.order = 1;
    p.bias = ep->bias;
    p.lens = ep->row_lens; p.rowsB = ep->rows_per_step > 0 ? ep->rows_per_step : 1; p.rowsG = ep->row_group > 0 ? ep->row_group : 1;
    p.flags = ep->flags;
    p.drop.rate = ep->drop_rate; p.drop.seed = ep->drop_seed; p.drop.step = ep->drop_step;
    p.drop.stream = ep->drop_stream; p.ld_logical = ep->drop_ld > 0 ? ep->drop_ld : F;
    // E2T_CONV_FWD (diagnostics): "ws<D>[x<splits>]" = the wave-specialised form (default ws4), "<K step>x<stages>[x<splits>[x<utterances
    // per tile>]]" = the all-DMA form (64x2, 64x4, 32x2).  splits 0 / absent: K is cut so that the last round of workgroups is full.
    struct ConvCfg { int ks, ns, splits, ub, ws; };
    static const ConvCfg cfg = [] {
        ConvCfg r{64, 2, 0, 64, 4};
        const char* e = e2t_dbg_str("E2T_CONV_FWD");
        if (!e) return r;
        if (e[0] == 'w' && e[1] == 's') { int d = 0, sp = 0; const int got = sscanf(e + 2, "%dx%d", &d, &sp); if (got >= 1 && (d == 4 || d == 6)) r.ws = d; if (got >= 2) r.splits = sp; return r; }
        int k = 0, n = 0, sp = 0, u = 0;
        const int got = sscanf(e, "%dx%dx%dx%d", &k, &n, &sp, &u);
        if (got >= 2 && ((k == 64 && (n == 2 || n == 4)) || (k == 32 && n == 2))) { r.ks = k; r.ns = n; r.ws = 0; }
        if (got >= 3) r.splits = sp;
        if (got >= 4 && u >= 1 && u <= 64 && 64 % u == 0) r.ub = u;
        return r;
    }();
    a.ub = cfg.ub; a.S = S;
    const int tiles = ((S + 64 / a.ub - 1) / (64 / a.ub)) * ((B + a.ub - 1) / a.ub);
    const int nk = N * C / cfg.ks;
    int splits = cfg.splits;
    if (splits <= 0) {
        const int wgs = 256 * (cfg.ws ? 2 : conv_wgs_per_cu(cfg.ks, cfg.ns));
        double best = 0.0;
        splits = 1;
        for (int c = 1; c <= 4 && nk / c >= 32; ++c) {
            const long w = (long)tiles * c;
            const double u = (double)w / (double)(((w + wgs - 1) / wgs) * wgs);
            if (u > best + 0.03) { best = u; splits = c; }
        }
    }
    if (splits > 1) {
        const size_t need = (size_t)splits * a.M * F * sizeof(float);
        if (!ep->splitk_ws || ep->splitk_ws_bytes < need) splits = 1;
    }
    if (splits > 1) { p.splits = splits; p.slab = (float*)ep->splitk_ws; }
    const dim3 grid((unsigned)(tiles * p.splits));
    const hipStream_t st = (hipStream_t)stream;
#define E2T_CONV_GO(KERNEL_, THREADS_, LDS_)                                                                                \
    do {                                                                                                                    \
        static const hipError_t rc_ = hipFuncSetAttribute((const void*)KERNEL_, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_); \
        if (rc_ != hipSuccess) { e2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(rc_)); return E2T_ERR_HIP; }    \
        hipLaunchKernelGGL(KERNEL_, grid, dim3(THREADS_), LDS_, st, a);                                                     \
    } while (0)
    constexpr int ws_lds = 3 * (64 + 128) * 8 * 16;
    if (cfg.ws == 6) E2T_CONV_GO((k_conv_fwd_ws<6>), 512, ws_lds);
    else if (cfg.ws) E2T_CONV_GO((k_conv_fwd_ws<4>), 512, ws_lds);
    else if (cfg.ks == 32) E2T_CONV_GO((k_conv_fwd<32, 2>), 256, conv_lds_bytes(32, 2));
    else if (cfg.ns == 4) E2T_CONV_GO((k_conv_fwd<64, 4>), 256, conv_lds_bytes(64, 4));
    else E2T_CONV_GO((k_conv_fwd<64, 2>), 256, conv_lds_bytes(64, 2));
#undef E2T_CONV_GO
    if (p.splits > 1) {
        const size_t n = (size_t)a.M * ((F + 3) / 4);
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)(splitk_reduce_stride_g(p.M, p.N, splitk_reduce_groups(p)) / 256), 1), dim3(256), 0, st, p);
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

// ---- measurement hook: per-launch [start, end] stamps of the GEMM kernels (bench.py measures the IN-STEP duration of the launches
// of a replayed graph with it; null = off, the normal state) ----
static unsigned long long* g_stamp_buf = nullptr;
static int g_stamp_slots = 0, g_stamp_seq = 0;
static int g_stamp_kind[E2T_GEMM_STAMP_MAX];
static unsigned long long* next_stamp(int kind) {
    if (!g_stamp_buf || g_stamp_slots <= 0) return nullptr;
    const int slot = g_stamp_seq++ % g_stamp_slots;
    g_stamp_kind[slot] = kind;
    return g_stamp_buf + 2 * slot;
}
extern "C" int e2t_gemm_stamps(void* buf, int slots) {
    E2T_CHECK_ARG(slots >= 0 && slots <= E2T_GEMM_STAMP_MAX && (buf || slots == 0));
    g_stamp_buf = (unsigned long long*)buf; g_stamp_slots = buf ? slots : 0; g_stamp_seq = 0;
    for (int i = 0; i < E2T_GEMM_STAMP_MAX; ++i) g_stamp_kind[i] = -1;
    return E2T_OK;
}
extern "C" int e2t_gemm_stamp_kinds(int* kinds, int n) {
    E2T_CHECK_ARG(kinds && n >= 0 && n <= E2T_GEMM_STAMP_MAX);
    for (int i = 0; i < n; ++i) kinds[i] = g_stamp_kind[i];
    return E2T_OK;
}

// Column sums of a K-major bf16 matrix: out[n] = alpha * sum_k B[k][n] (+ out[n]).  The bias row of a weight gradient [x | 1]^T . dG
// whose ones column would otherwise cost a product of its own (E2T_GEMM_LAST_ROW_ONES, gemm_launch): one pass over dG at the HBM
// rate, one launch, no slabs and no reduction kernel behind it (that reduction -- 8 workgroups -- sat 145 us behind a 256 x 256
// GEMM of the other branch that held every register of the chip, on the branch that ends cfg4's step).  A workgroup of 1024
// threads owns 32 columns (64 B of every row) and all K rows: thread = (8-column group, row lane), rows k = lane, lane + 256, ...
// summed in that order; 16 row lanes meet by wave shuffles, the 16 waves through LDS: a fixed order, the same bits every run.
__global__ __launch_bounds__(1024) void k_colsum_bf16(const bf16_t* B, int ldb, int K, int N, float* out, float alpha, int accumulate) {
    __shared__ float part[16][32];
    const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c0 = blockIdx.x * 32 + cg * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {                                          // (ldb % 8 == 0 and the row is readable up to ldb: a group is loaded whole)
        const bf16_t* q = B + c0;
#pragma unroll 8
        for (int k = rl; k < K; k += 256) {
            const uint4 v = *(const uint4*)(q + (size_t)k * ldb);
            acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xFFFF0000u);
            acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xFFFF0000u);
            acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xFFFF0000u);
            acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) acc[j] += __shfl_xor(acc[j], m, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave][lane * 8 + j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = blockIdx.x * 32 + threadIdx.x;
        if (c < N) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += part[w][threadIdx.x];
            s *= alpha;
            out[c] = accumulate ? out[c] + s : s;
        }
    }
}

struct GemmPlan { int tile, splits, batch; bool want_split; };
static int gemm_order() {
    static const int o = e2t_dbg_int("E2T_GEMM_ORDER", 1) == 0 ? 0 : 1;
    return o;
}
// Which instance a product runs on, and its split count (shared by the launcher and e2t_gemm_plan).
static GemmPlan gemm_plan(bool tn, int M, int N, int K, const e2t_gemm_epilogue* ep, int want_tile = 0, int want_splits = 0) {
    // tile choice: 256x256 for large plain products, 128x128 otherwise; E2T_GEMM_TILE=128|256 overrides (diagnostics)
    static const int forced = e2t_dbg_int("E2T_GEMM_TILE", 0);
    // Split-K: requested by the caller (weight gradients: K = S*B, a few dozen output tiles) or chosen here when the
    // product has too few 128x128 tiles to fill the chip and a long K loop (conv front-end, input gradients of narrow
    // layers).  Partial slabs go to the caller's workspace; k_splitk_reduce sums them and applies the epilogue.
    const int kt = BK;
    const int nfull = tn ? (K + kt - 1) / kt : K / BK;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const bool have_ws = ep && ep->splitk_ws && ep->splitk_ws_bytes > 0;
    const bool want_split = have_ws && ((ep->flags & E2T_GEMM_SPLITK) || (t128 <= 160 && nfull >= 16));
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const bool rich = ep && ((ep->flags & (E2T_GEMM_RELU | E2T_GEMM_DROPOUT)) || ep->relu_bwd_src);
    bool big = !tn && !want_split && !rich && t256 >= 192 && K >= 64;
    // K-major products with both output dimensions large (cfg4's weight gradients: 2049 x 8192, 1024 x 4096 x 2): 256 x 256 tiles
    // halve the bytes staged per flop; E2T_TN256=0 keeps the 128 x 128 instance
    static const bool tn256_ok = e2t_dbg_int("E2T_TN256", 1) != 0;
    const int nbatch = (ep && ep->batch > 1) ? ep->batch : 1;
    static const int tn256_min = e2t_dbg_int("E2T_TN256_MIN", 1024);
    // (at least HALF a round of 256 x 256 workgroups: the vocabulary projection's weight gradient at cfg4, 1806 x 2049 x 2560 = 72 tiles,
    //  took 75 us on this instance -- 64 tiles x 2 splits on 256 CUs -- against 38 us for the vendor BLAS; it now joins the head's grouped
    //  128 x 128 launch)
    static const int tn256_tiles = e2t_dbg_int("E2T_TN256_TILES", 128);
    const bool big_tn = tn && tn256_ok && !rich && have_ws && M >= tn256_min && N >= tn256_min && t256 * nbatch >= tn256_tiles && forced != 128;
    if (big_tn) big = true;
    if (forced == 128) big = false;
    if (forced == 256 && !tn && !want_split && !rich) big = true;
    if (want_tile == 128) big = false;                        // (the tail part of a product: gemm_launch)
    GemmPlan pl;
    pl.tile = big ? 256 : 128;
    pl.batch = (ep && ep->batch > 1) ? ep->batch : 1;
    pl.want_split = want_split || (want_splits > 1 && have_ws);
    pl.splits = 1;
    if (want_splits > 1 && have_ws) {
        int s = std::min(want_splits, std::max(1, nfull / 16));
        const size_t per = (size_t)M * N * sizeof(float) * pl.batch;
        if ((size_t)s * per > ep->splitk_ws_bytes) s = (int)(ep->splitk_ws_bytes / per);
        pl.splits = std::max(1, s);
    } else if (want_split) {
        const int ntm = (M + pl.tile - 1) / pl.tile, ntn = (N + pl.tile - 1) / pl.tile;
        const int tiles = ntm * ntn * pl.batch;
        const int slots = 512;
        int s = slots / tiles;
        if (big) {
            // one 256 x 256 workgroup per CU: the split count that fills the last round best, smallest on ties
            double bu = 0.0;
            s = 1;
            for (int c = 1; c <= 8 && (c == 1 || nfull / c >= 16); ++c) {
                const long w = (long)tiles * c;
                const double u = (double)w / (double)(((w + 255) / 256) * 256);
                if (u > bu + 0.02) { bu = u; s = c; }
            }
        }                         // fill, but never exceed, the 2 (4) x 256 resident workgroups: one block
                                                       // too many costs a whole second round (175 x 3 = 525 -> 175 x 2)
        if (s > nfull * kt / 1024) s = nfull * kt / 1024;   // keep >= 16 K tiles (of 64) per split: a workgroup's fixed cost (DMA fill, slab
                                                       // store, its share of the reduction) is worth ~8 of them (measured on the
                                                       // train step: 6 -> 16 tiles per split -1.2 %, 24 and more slower again)
        // (measured, E2T_GEMM_SPLITS sweep, dW_x 801 x 3200 x 8704 = 175 tiles: 1 split 114 us, 2: 92, 3: 105, 4: 94; batched
        //  dW_h 104 tiles: 1: 109, 2: 67, 3: 63, 4: 56 -- a workgroup ALONE on a CU is not faster than one of two: a K tile
        //  costs ~0.8 us of LDS-DMA issue + wait + barrier either way, so filling both slots of every CU is what counts)
        { static const int forced_s = e2t_dbg_int("E2T_GEMM_SPLITS", 0); if (forced_s > 0) s = forced_s; }
        const size_t per = (size_t)M * N * sizeof(float) * pl.batch;
        if ((size_t)s * per > ep->splitk_ws_bytes) s = (int)(ep->splitk_ws_bytes / per);
        if (s < 1) s = 1;
        pl.splits = s;
    }
    return pl;
}

// validated kernel arguments of one product (splits / slab are filled in by the caller from its plan)
static int gemm_make_args(bool tn, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                          int M, int N, int K, const e2t_gemm_epilogue* ep, GemmArgs& p) {
    E2T_CHECK_ARG(A && B && C);
    E2T_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
    E2T_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0);
    if (tn) E2T_CHECK_ARG(lda >= (M + 7) / 8 * 8 && ldb >= (N + 7) / 8 * 8);      // K-major operands [K][M], [K][N]
    else E2T_CHECK_ARG(K % 8 == 0 && lda >= K && ldb >= K);
    E2T_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0);
    p = GemmArgs{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = 1.0f;
    p.splits = 1;
    p.batch = 1;
    if (ep) {
        p.bias = ep->bias;
        p.mask_src = (const bf16_t*)ep->relu_bwd_src; p.ld_mask = ep->ld_relu_bwd_src;
        p.lens = ep->row_lens; p.rowsB = ep->rows_per_step > 0 ? ep->rows_per_step : 1; p.rowsG = ep->row_group > 0 ? ep->row_group : 1;
        p.alpha = ep->alpha;
        p.flags = ep->flags & ~(E2T_GEMM_LAST_ROW_ONES | E2T_GEMM_KEEP_SLABS);      // (hints to the launcher: the kernels and the choice of the reduction never see them)
        p.drop.rate = ep->drop_rate; p.drop.seed = ep->drop_seed; p.drop.step = ep->drop_step;
        p.drop.stream = ep->drop_stream; p.ld_logical = ep->drop_ld > 0 ? ep->drop_ld : N;
        p.last_col_out = ep->last_col_out;
        E2T_CHECK_ARG(!((p.flags & E2T_GEMM_OUT_BF16) && (p.flags & E2T_GEMM_ACCUMULATE)));
        if (ep->batch > 1) { p.batch = ep->batch; p.a_bs = ep->a_batch_stride; p.b_bs = ep->b_batch_stride; p.c_bs = ep->c_batch_stride; }
    }
    E2T_CHECK_ARG(ldc >= (p.last_col_out ? N - 1 : N));
    p.order = gemm_order();
    return E2T_OK;
}

static int gemm_launch(bool tn, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                       int M, int N, int K, const e2t_gemm_epilogue* ep, void* stream,
                       int row0 = 0, int want_tile = 0, int want_splits = 0) {
    GemmArgs p;
    if (int rc = gemm_make_args(tn, A, lda, B, ldb, C, ldc, M, N, K, ep, p)) return rc;
    p.row0 = row0;
    if (M == 0 || N == 0) return E2T_OK;
    // the K-major instances store through the lean epilogue (alpha, bias, accumulate, row mask, bf16: 139 registers, which lets
    // a workgroup share a CU with the persistent BPTT): ReLU / dropout / the ReLU-backward mask exist on their split-K path
    // only, where the reduction applies the full epilogue
    const bool rich_ep = ep && ((ep->flags & (E2T_GEMM_RELU | E2T_GEMM_DROPOUT)) || ep->relu_bwd_src);
    const GemmPlan pl = gemm_plan(tn, M, N, K, ep, want_tile, want_splits);
    const bool big = pl.tile == 256;
    // The last round of a large K-contiguous product on the 128 x 128 instance.  Its tiles are dealt to the chip in rounds of 512
    // resident workgroups; a few tiles beyond a whole number of rounds cost a whole round more -- cfg4's input gradient 8704 x 2048
    // x 8192: 68 x 16 = 1088 tiles = 2.125 rounds, on the critical path of the backward pass.  The tile rows of the whole rounds
    // leave as they are; the rows beyond them leave as a second launch that fills the chip once with short workgroups: K split over
    // 512 / tiles workgroups per tile, the reduction applies the full epilogue.  A sub-range of M is a row offset of A, C and the
    // epilogue's row-indexed operands; the row-length mask and the dropout counter get the offset as GemmArgs::row0.
    // Measured (same box, cfg4): the product alone 329 -> 306 us; the step 8.65 -> 8.58 ms.  (The same cut for the 256 x 256 instance
    // -- the rows beyond the whole rounds on 128 x 128 tiles: input projection 8704 x 8192 x 2112, 4.25 rounds of 256 -- is faster
    // alone, 303 -> 279 us, and made the step SLOWER, 8.65 -> 8.73 ms: not in the tree.)
    if (!tn && !big && row0 == 0 && want_tile == 0 && want_splits == 0 && pl.batch == 1 && pl.splits <= 1 && ep && !ep->last_col_out) {
        const int slots = 512;
        const int ntm_ = (M + 127) / 128, ntn_ = (N + 127) / 128;
        const long tiles = (long)ntm_ * ntn_;
        const long rem = tiles % slots;
        if (tiles > slots && rem > 0 && rem * 10 <= slots * 3 && (tiles - rem) % ntn_ == 0 && K / BK >= 32 &&
            ep->splitk_ws && ep->splitk_ws_bytes > 0 && !(ep->flags & E2T_GEMM_SPLITK)) {
            const int rows_a = (int)((tiles - rem) / ntn_) * 128;           // rows of the whole rounds
            const int rows_b = M - rows_a;
            const int elt = (ep->flags & E2T_GEMM_OUT_BF16) ? 2 : 4;
            const int splits_b = (int)std::min<long>(8, std::max<long>(1, slots / rem));
            if (splits_b > 1 && (size_t)splits_b * rows_b * N * sizeof(float) <= ep->splitk_ws_bytes) {
                e2t_gemm_epilogue eb = *ep;
                if (ep->relu_bwd_src) eb.relu_bwd_src = (const char*)ep->relu_bwd_src + (size_t)rows_a * ep->ld_relu_bwd_src * 2;
                if (int rc = gemm_launch(false, A, lda, B, ldb, C, ldc, rows_a, N, K, ep, stream, 0, 128, -1)) return rc;
                return gemm_launch(false, (const bf16_t*)A + (size_t)rows_a * lda, lda, B, ldb, (char*)C + (size_t)rows_a * ldc * elt, ldc,
                                   rows_b, N, K, &eb, stream, rows_a, 128, splits_b);
            }
        }
    }
    // Ragged edge of a large K-major product.  The weight gradient of a layer with D inputs is [x | 1]^T . dG: M = D + 1, and with
    // D a multiple of 256 the ones column costs a whole extra row of 256 x 256 tiles (cfg4's dW_x, 2049 x 8192 x 8704: 9 x 32 tiles
    // = 288 on 256 CUs -- two rounds or three K splits; measured 429 us against 257 for 2048 rows).  The few rows (columns) beyond
    // the last full tile leave as a product of their own on the 128 x 128 split-K instance, behind the main product on the same
    // stream (its slabs reuse the workspace).  A K-major sub-range of M (N) is a column offset of A (B) and a row (column) offset of C.
    if (tn && big && pl.batch == 1 && !rich_ep && !(ep && ep->row_lens) && !p.lens) {
        const int rm = M % 256, rn = N % 256;
        const int elt = (ep && (ep->flags & E2T_GEMM_OUT_BF16)) ? 2 : 4;
        if (rm == 1 && M > 256 && (ep->flags & E2T_GEMM_LAST_ROW_ONES) && !ep->bias && !ep->last_col_out && elt == 4 && ldb % 8 == 0 &&
            (((uintptr_t)B) & 15) == 0) {
            // the one row beyond the last full tile is the ones column's: column sums of B, one pass, no slabs
            e2t_gemm_epilogue e0 = *ep;
            e0.flags &= ~E2T_GEMM_KEEP_SLABS; e0.slabs_out = nullptr;            // (the row of sums is written straight into C: so is the rest)
            if (ep->slabs_out) *ep->slabs_out = e2t_slab_info{nullptr, 1, 1, 0};
            if (int rc = gemm_launch(true, A, lda, B, ldb, C, ldc, M - 1, N, K, &e0, stream)) return rc;
            hipLaunchKernelGGL(k_colsum_bf16, dim3((unsigned)((N + 31) / 32)), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)B, ldb, K, N,
                               (float*)C + (size_t)(M - 1) * ldc, ep->alpha, (ep->flags & E2T_GEMM_ACCUMULATE) ? 1 : 0);
            E2T_LAUNCH_CHECK();
            return E2T_OK;
        }
        if (rm > 0 && rm <= 32 && M > 256) {
            e2t_gemm_epilogue e0 = *ep, e1 = *ep;
            e0.flags &= ~E2T_GEMM_KEEP_SLABS; e1.flags &= ~E2T_GEMM_KEEP_SLABS; e0.slabs_out = e1.slabs_out = nullptr;      // (two launches, two split counts)
            if (ep->slabs_out) *ep->slabs_out = e2t_slab_info{nullptr, 1, 1, 0};
            if (ep->last_col_out) e1.last_col_out = ep->last_col_out + (M - rm);
            if (int rc = gemm_launch(true, A, lda, B, ldb, C, ldc, M - rm, N, K, &e0, stream)) return rc;
            return gemm_launch(true, (const bf16_t*)A + (M - rm), lda, B, ldb, (char*)C + (size_t)(M - rm) * ldc * elt, ldc, rm, N, K, &e1, stream);
        }
        if (rn > 0 && rn <= 32 && N > 256) {
            e2t_gemm_epilogue e0 = *ep, e1 = *ep;
            e0.flags &= ~E2T_GEMM_KEEP_SLABS; e1.flags &= ~E2T_GEMM_KEEP_SLABS; e0.slabs_out = e1.slabs_out = nullptr;
            if (ep->slabs_out) *ep->slabs_out = e2t_slab_info{nullptr, 1, 1, 0};
            e0.last_col_out = nullptr;
            if (ep->bias) e1.bias = ep->bias + (N - rn);
            if (int rc = gemm_launch(true, A, lda, B, ldb, C, ldc, M, N - rn, K, &e0, stream)) return rc;
            return gemm_launch(true, A, lda, (const bf16_t*)B + (N - rn), ldb, (char*)C + (size_t)(N - rn) * elt, ldc, M, rn, K, &e1, stream);
        }
    }
    const int BM = pl.tile, BN = BM;
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    const int batch = pl.batch;
    if (pl.splits > 1 || pl.want_split) { p.splits = pl.splits; p.slab = (float*)ep->splitk_ws; }
    E2T_CHECK_ARG(!(tn && rich_ep && p.splits <= 1));
    const dim3 grid((unsigned)(ntm * ntn * p.splits * batch));
    const hipStream_t st = (hipStream_t)stream;
    p.stamp = next_stamp(tn ? (big ? E2T_GEMM_KIND_TN256 : E2T_GEMM_KIND_TN128) : (big ? E2T_GEMM_KIND_NT256 : E2T_GEMM_KIND_NT128));
    // every instance asks for its LDS explicitly (above the 64-KiB default for most of them)
#define E2T_GEMM_GO(BM_, BN_, WM_, WN_, RICH_, TN_, KT_, NS_, D_)                                                           \
    do {                                                                                                                    \
        constexpr int lds_ = gemm_lds_bytes(BM_, BN_, KT_, NS_);                                                            \
        static const hipError_t rc_ = hipFuncSetAttribute((const void*)k_gemm_nt<BM_, BN_, WM_, WN_, RICH_, TN_, KT_, NS_, D_>, \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                \
        if (rc_ != hipSuccess) { e2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(rc_)); return E2T_ERR_HIP; }    \
        hipLaunchKernelGGL((k_gemm_nt<BM_, BN_, WM_, WN_, RICH_, TN_, KT_, NS_, D_>), grid, dim3(64 * WM_ * WN_), lds_, st, p); \
    } while (0)
    // E2T_GEMM_DBG=1|2|3 selects the timing-only forms of the 128 x 128 instances (scripts/gemm_loop_probe.py)
    static const int dbg = e2t_dbg_int("E2T_GEMM_DBG", 0);
    (void)dbg;
#ifdef E2T_DEBUG
    // (the timing-only instances exist in the diagnostics build only: the product library carries no kernel it never launches)
    if (tn && !big && dbg == 1) E2T_GEMM_GO(128, 128, 2, 2, false, true, 64, 2, 1);
    else if (tn && !big && dbg == 2) E2T_GEMM_GO(128, 128, 2, 2, false, true, 64, 2, 2);
    else if (tn && !big && dbg == 3) E2T_GEMM_GO(128, 128, 2, 2, false, true, 64, 2, 3);
    else if (!tn && !big && dbg == 1) E2T_GEMM_GO(128, 128, 2, 2, true, false, 64, 2, 1);
    else if (!tn && !big && dbg == 2) E2T_GEMM_GO(128, 128, 2, 2, true, false, 64, 2, 2);
    else if (big && !tn && dbg == 4) E2T_GEMM_GO(256, 256, 2, 4, false, false, 64, 2, 4);     // everything but the stores of the epilogue
    else if (big && !tn && dbg == 2) E2T_GEMM_GO(256, 256, 2, 4, false, false, 64, 2, 2);     // loads, barriers and the epilogue only
    else
#endif
    if (big && tn) E2T_GEMM_GO(256, 256, 2, 4, false, true, 64, 2, 0);
    else if (tn) E2T_GEMM_GO(128, 128, 2, 2, false, true, 64, 2, 0);
    else if (big) E2T_GEMM_GO(256, 256, 2, 4, false, false, 64, 2, 0);
    else E2T_GEMM_GO(128, 128, 2, 2, true, false, 64, 2, 0);
#undef E2T_GEMM_GO
    if (ep && ep->slabs_out) *ep->slabs_out = e2t_slab_info{nullptr, 1, batch, 0};
    if (p.splits > 1) {
        if (tn && ep && (ep->flags & E2T_GEMM_KEEP_SLABS) && ep->slabs_out && p.alpha == 1.0f && p.ldc == p.N && splitk_reduce_plain(p)) {
            // the slabs stay where they are: the optimiser kernel sums them as it reads the gradient
            *ep->slabs_out = e2t_slab_info{p.slab, p.splits, batch, (long long)p.M * p.N};
        } else {
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)(splitk_reduce_stride_g(p.M, p.N, splitk_reduce_groups(p)) / 256), batch), dim3(256), 0, (hipStream_t)stream, p);
        }
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

extern "C" int e2t_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                int M, int N, int K, const e2t_gemm_epilogue* ep, void* stream) {
    return gemm_launch(false, A, lda, B, ldb, C, ldc, M, N, K, ep, stream);
}

extern "C" int e2t_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                int M, int N, int K, const e2t_gemm_epilogue* ep, void* stream) {
    return gemm_launch(true, A, lda, B, ldb, C, ldc, M, N, K, ep, stream);
}

// Several K-major products in one launch (k_gemm_tn_group).  The K splits are chosen for the GROUP: all products are cut into
// work items of about the same K depth, `depth` K tiles, the depth that minimises (rounds of 512 resident workgroups) x (time
// of an item); slabs of all products share the workspace of the first product's epilogue.
extern "C" int e2t_gemm_tn_group_bf16(int n, const e2t_gemm_call* calls, void* stream) {
    E2T_CHECK_ARG(n >= 1 && n <= E2T_GEMM_GROUP_MAX && calls);
    GemmGroupArgs g{};
    int tiles[E2T_GEMM_GROUP_MAX], kt[E2T_GEMM_GROUP_MAX];
    const e2t_gemm_epilogue* eps[E2T_GEMM_GROUP_MAX];
    int m = 0;
    const e2t_gemm_epilogue* ep0 = nullptr;
    for (int i = 0; i < n; ++i) {
        const e2t_gemm_call& c = calls[i];
        E2T_CHECK_ARG(c.ep);
        // the grouped kernel stores through the lean epilogue and a product may end up unsplit (K no deeper than the group's item
        // depth, or the workspace too small for its slabs): ReLU / dropout / the ReLU-backward mask exist on the reduction's path only
        E2T_CHECK_ARG(!((c.ep->flags & (E2T_GEMM_RELU | E2T_GEMM_DROPOUT)) || c.ep->relu_bwd_src));
        GemmArgs p;
        if (int rc = gemm_make_args(true, c.A, c.lda, c.B, c.ldb, c.C, c.ldc, c.M, c.N, c.K, c.ep, p)) return rc;
        if (c.M == 0 || c.N == 0) continue;
        if (!ep0) ep0 = c.ep;
        tiles[m] = ((c.M + 127) / 128) * ((c.N + 127) / 128) * p.batch;
        kt[m] = (c.K + BK - 1) / BK;
        eps[m] = c.ep;
        g.p[m++] = p;
    }
    if (m == 0) return E2T_OK;
    E2T_CHECK_ARG(ep0->splitk_ws && ep0->splitk_ws_bytes > 0);
    // Item depth = the LARGEST K depth that still yields about a round of workgroups (>= 300 of the 512 resident slots), never
    // below 16 K tiles.  Measured on the train step (E2T_GEMM_GROUP_DEPTH sweep, cfg2): depth 16 / 23: 1.87 / 1.91 ms, 34: 1.765,
    // 46: 1.75, 68: 1.742, no split at all: 1.75 -- beside the persistent BPTT the products are not short of workgroups, and
    // every extra split is slab traffic plus a longer reduction; only a group with few tiles (the bottom layer's, at the tail of
    // the step on an otherwise idle chip) needs the splits to fill it.
    int kmax = 0;
    for (int i = 0; i < m; ++i) kmax = std::max(kmax, kt[i]);
    static const int forced_depth = e2t_dbg_int("E2T_GEMM_GROUP_DEPTH", 0);
    int best_d = std::min(16, kmax);
    for (int d = kmax; d >= std::min(16, kmax); --d) {
        long items = 0; size_t bytes = 0;
        for (int i = 0; i < m; ++i) {
            const int sp = (kt[i] + d - 1) / d;
            items += (long)tiles[i] * sp;
            if (sp > 1) bytes += ((size_t)sp * g.p[i].M * g.p[i].N * g.p[i].batch * sizeof(float) + 255) / 256 * 256;
        }
        if (bytes > ep0->splitk_ws_bytes) continue;
        // (round 5: a group of fewer than 200 tiles -- the bottom layer's, which ends the step beside the HBM-bound optimiser update, and the
        //  head's -- is cut for about 450 workgroups instead of 300: three K ranges per tile instead of two; cfg2 1.579 -> 1.569 ms over three
        //  same-box pairs, the other configurations unchanged; 600 was slower again.  E2T_GROUP_FEW_TILES=0 switches it off.)
        { static const int few = e2t_dbg_int("E2T_GROUP_FEW_TILES", 200), want = e2t_dbg_int("E2T_GROUP_FEW_ITEMS", 448);
          long tt = 0; for (int i = 0; i < m; ++i) tt += tiles[i];
          if (items >= ((few > 0 && tt < few) ? want : 300) || d == std::min(16, kmax)) { best_d = d; break; } }
    }
    if (forced_depth > 0) best_d = std::min(kmax, forced_depth);
    size_t off = 0;
    int first = 0;
    bool any_split = false;
    unsigned max_red = 0; int max_batch = 1;
    for (int i = 0; i < m; ++i) {
        GemmArgs& p = g.p[i];
        int sp = (kt[i] + best_d - 1) / best_d;
        const size_t need = (size_t)sp * p.M * p.N * p.batch * sizeof(float);
        if (sp > 1 && off + need > ep0->splitk_ws_bytes) sp = 1;
        p.splits = sp;
        const e2t_gemm_epilogue* epi = eps[i];
        if (epi->slabs_out) *epi->slabs_out = e2t_slab_info{nullptr, 1, p.batch, 0};
        if (sp > 1) {
            p.slab = (float*)((char*)ep0->splitk_ws + off);
            off += (need + 255) / 256 * 256;
            if ((epi->flags & E2T_GEMM_KEEP_SLABS) && epi->slabs_out && p.alpha == 1.0f && p.ldc == p.N && splitk_reduce_plain(p)) {
                // left for the optimiser kernel (E2T_GEMM_KEEP_SLABS): the grouped reduction skips this product
                *epi->slabs_out = e2t_slab_info{p.slab, sp, p.batch, (long long)p.M * p.N};
                p.flags |= E2T_GEMM_KEEP_SLABS;
            } else {
                any_split = true;
                max_red = std::max(max_red, (unsigned)(splitk_reduce_stride_g(p.M, p.N, splitk_reduce_groups(p)) / 256));
                max_batch = std::max(max_batch, p.batch);
            }
        }
        g.first[i] = first;
        g.count[i] = tiles[i] * sp;
        first += (g.count[i] + 7) / 8 * 8;
    }
    g.first[m] = first;
    g.n = m;
    g.stamp = next_stamp(E2T_GEMM_KIND_TN128G);
    static const hipError_t rc_ = hipFuncSetAttribute((const void*)k_gemm_tn_group, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(128, 128, 64, 2));
    if (rc_ != hipSuccess) { e2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(rc_)); return E2T_ERR_HIP; }
    hipLaunchKernelGGL(k_gemm_tn_group, dim3((unsigned)first), dim3(256), gemm_lds_bytes(128, 128, 64, 2), (hipStream_t)stream, g);
    if (any_split) hipLaunchKernelGGL(k_splitk_reduce_group, dim3(max_red, (unsigned)m, (unsigned)max_batch), dim3(256), 0, (hipStream_t)stream, g);
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

extern "C" int e2t_gemm_plan(int tn, int M, int N, int K, const e2t_gemm_epilogue* ep, int* tile, int* splits) {
    E2T_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
    const GemmPlan pl = gemm_plan(tn != 0, M, N, K, ep);
    if (tile) *tile = pl.tile;
    if (splits) *splits = pl.splits;
    return E2T_OK;
}
