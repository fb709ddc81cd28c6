// Fused LSTM recurrence kernels (reference rows a7, a9:
// SequenceNetwork._encode_sequences and the decoder RNN, called from
// ecog2txt/trainers.py:318 via net.fit; 4-gate packing trainers.py:527-529).
//
// One launch = one time step of ONE layer for BOTH directions; the host side
// (e2t_lstm_seq_fwd / _bwd) issues the S launches back to back on one stream so
// the whole sequence replays from a hipGraph.  Layers of a bidirectional stack
// are strictly serial (layer l+1 at t=0 needs layer l's backward direction at
// t=0, produced last), so the only concurrency is {fwd,bwd} x batch x units,
// which is exactly the grid: (unit tiles of 16, row blocks of 64, directions).
//
// A step is latency-bound (0.66 GFLOP), so the kernel is built to pay ONE memory
// round trip: a 256-thread workgroup owns 64 rows x 16 units; the W_h tile for
// those units (all four gates, pre-packed in MFMA fragment order so it is one
// linear stream) is DMA'd into LDS with global_load_lds by all four waves, each
// wave's own h_{t-1} rows and the epilogue operands (Gx, c_{t-1}) are requested
// into registers at the same time, then a single wait + barrier, then the MFMAs
// read B fragments conflict-free from LDS (ds_read_b128, lane-linear image).
// Every lane ends up holding all four gate pre-activations of its (row, unit)
// cells, so nonlinearities, cell update, dropout and stores are lane-local.
// h_{t-1} rows are gathered straight from the layer's bf16 output array with a
// per-row time index, so tf.reverse_sequence and ragged lengths cost nothing.
//
// Backward step: dh_rec = dG_{t+1} . W_h^T (K = 4H) for the same 64x16 patch,
// then the LSTM cell backward for those cells (dG_t in bf16 for the later
// weight-gradient GEMMs, dc carried in fp32).  dW_h / dW_x are NOT accumulated
// per step: they are large split-K GEMMs over all steps afterwards (SURVEY.md
// 7.3 item 3).
#include "common.h"
#include "ecog2txt_hip.h"
#include <stdlib.h>
#include <algorithm>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

extern __shared__ __attribute__((aligned(16))) uint4 lstm_smem[];

// ---------------------------------------------------------------------------
// Ownership (identical in the forward and backward step kernels):
//   workgroup = (unit tile ut of 16 units, row block of 64 utterances, direction);
//   8 waves: waves 0-3 each own the 16-utterance tile rt = row_block*4 + w (MFMA + cell update),
//   all 8 waves issue the operand DMA (a step moves ~124 KiB into the CU and is bound by how many
//   1-KiB wave transactions the CU keeps in flight: 8 issuing waves ~2x 4, LDS-DMA ~2x register loads).
//   MFMA roles: A operand = weight fragment (rows = units), B operand = state fragment
//   (columns = utterances), so D[i][j]: j = lane&15 = utterance, i = (lane>>4)*4 + r = unit.
//   => lane (frow, fq) owns utterance b = rt*16 + frow and the FOUR CONSECUTIVE units
//      u0 .. u0+3, u0 = ut*16 + fq*4.
// Saved per-cell state is "lane-native", indexed by PROCESSING STEP s (not by time):
//   Gs[((((s*ndir + dir)*RT + rt)*UT + ut)*2 + rp)*64 + lane]  8 x bf16 = (i, j, f, o) of units u0 + 2*rp, u0 + 2*rp + 1 (ABI 5: bf16;
//       the gates are what BPTT multiplies with -- rounded once, like the gate gradients it writes)
//   Cs[((((s*ndir + dir)*RT + rt)*UT + ut)*2 + half)*64 + lane]  float2 = c of units u0+2*half, u0+2*half+1
// so every save/restore is a fully coalesced 1-KiB wave transaction and needs no time index.
//
// LDS image of one K chunk of `kch` k-blocks (32 k each), in 16-B units:
//   W section : [NG gates][kch][64 lanes]                 MFMA fragment order (linear copy of the packed image)
//   S section : [4 row tiles][npair][2 row halves][64]    state rows, 8 rows x 128 B per DMA instruction,
//               16-B chunk position XOR-swizzled by (row & 7) on the SOURCE side (DMA writes LDS linearly)
//               -> conflict-free ds_read_b128 fragment reads, full 128-B lines on the global side.
// ---------------------------------------------------------------------------
__device__ __forceinline__ size_t native_tile(int s, int dir, int rt, int ut, int ndir, int RT, int UT) {
    return ((size_t)(s * ndir + dir) * RT + rt) * UT + ut;
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return fmaf(-2.0f, fast_rcp(__expf(2.0f * x) + 1.0f), 1.0f); }   // explicit fma: every kernel rounds alike

__device__ __forceinline__ float drop_scale_k(float rate, unsigned long long key, unsigned stream, unsigned long long idx) {
    if (rate <= 0.0f) return 1.0f;
    unsigned long long ctr = idx >> 2;
    unsigned r[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), stream, 0u, (unsigned)key, (unsigned)(key >> 32), r);
    unsigned v = r[idx & 3ull] >> 8;
    unsigned thresh = (unsigned)(rate * 16777216.0f);
    return v >= thresh ? 1.0f / (1.0f - rate) : 0.0f;
}
// scales for the 4 consecutive logical elements e0..e0+3: one Philox evaluation when they share a
// counter (e0 % 4 == 0, the common case), else four
__device__ __forceinline__ void drop_scale4(float rate, unsigned long long key, unsigned stream, unsigned long long e0, float (&sc)[4]) {
    const unsigned thresh = (unsigned)(rate * 16777216.0f);
    const float keep = 1.0f / (1.0f - rate);
    if ((e0 & 3ull) == 0) {
        unsigned long long ctr = e0 >> 2;
        unsigned r[4];
        philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), stream, 0u, (unsigned)key, (unsigned)(key >> 32), r);
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = (r[i] >> 8) >= thresh ? keep : 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = drop_scale_k(rate, key, stream, e0 + i);
    }
}
// streaming (non-temporal) stores for outputs nobody re-reads soon
__device__ __forceinline__ void nt_store_f4(float* p, float a, float b, float c, float d) {
    __builtin_nontemporal_store((f32x4){a, b, c, d}, (f32x4*)p);
}
__device__ __forceinline__ void nt_store_bf4(bf16_t* p, bf16_t a, bf16_t b, bf16_t c, bf16_t d) {
    unsigned long long v = (unsigned long long)a | ((unsigned long long)b << 16) | ((unsigned long long)c << 32) | ((unsigned long long)d << 48);
    __builtin_nontemporal_store(v, (unsigned long long*)p);
}
// gate saves: (i, j, f, o) of one unit as 4 x bf16 (hardware converter, round-to-nearest-even), and back
__device__ __forceinline__ uint2 gates_pack(float gi, float gj, float gf, float go) { return make_uint2(f2bf_pk(gi, gj), f2bf_pk(gf, go)); }
__device__ __forceinline__ float4 gates_unpack(unsigned ij, unsigned fo) {
    return make_float4(__uint_as_float(ij << 16), __uint_as_float(ij & 0xFFFF0000u), __uint_as_float(fo << 16), __uint_as_float(fo & 0xFFFF0000u));
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store_u4(void* p, uint4 v) { __builtin_nontemporal_store(*(u32x4_t*)&v, (u32x4_t*)p); }
// (the cell-state saves are next read by the BPTT, a whole forward and half a backward pass later: at cfg5, where a layer's saves
//  exceed the infinity cache, the non-temporal store is worth 1 % of the step -- 6.54 -> 6.48 ms same box; cfg2 / cfg4 unchanged)
__device__ __forceinline__ void nt_store_f2(float* p, float a, float b) { typedef float f2_t __attribute__((ext_vector_type(2))); __builtin_nontemporal_store((f2_t){a, b}, (f2_t*)p); }
__device__ __forceinline__ void nt_store_u2(void* p, uint2 v) { __builtin_nontemporal_store(*(unsigned long long*)&v, (unsigned long long*)p); }
__device__ __forceinline__ float4 ld4(const float* p, bool vec, int n) {
    if (vec) return *(const float4*)p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0) v.x = p[0]; if (n > 1) v.y = p[1]; if (n > 2) v.z = p[2]; if (n > 3) v.w = p[3];
    return v;
}
__device__ __forceinline__ void st4(float* p, float4 v, bool vec, int n) {
    if (vec) { *(float4*)p = v; return; }
    if (n > 0) p[0] = v.x; if (n > 1) p[1] = v.y; if (n > 2) p[2] = v.z; if (n > 3) p[3] = v.w;
}

// geometry shared by host and device
struct StepGeom {
    int kch;        // k-blocks per LDS chunk (even, or == KB when single chunk)
    int nch;        // number of chunks
    int npair;      // ceil(kch / 2)
    int bufsz;      // 16-B units per chunk buffer
    int nbuf;       // 1 or 2
};
__host__ __device__ inline StepGeom step_geom(int KB, int NG, int extra16 /* 16-B units outside the chunk buffers */) {
    // one k-block costs NG KiB (weights) + 4 KiB (4 state tiles); 160 KiB LDS per CU
    const int per_kb = NG * 64 + 4 * 64;                 // 16-B units
    const int budget = (160 * 1024) / 16 - extra16;
    StepGeom g;
    int kb_pad = (KB + 1) & ~1;
    if (kb_pad * per_kb <= budget) { g.kch = KB; g.nch = 1; g.nbuf = 1; }
    else {
        int k = (budget / 2) / per_kb; k &= ~1; if (k < 2) k = 2;
        g.kch = k; g.nch = (KB + k - 1) / k; g.nbuf = 2;
    }
    g.npair = (g.kch + 1) / 2;
    g.bufsz = NG * g.kch * 64 + 4 * g.npair * 128;
    return g;
}

// issue the DMA instructions of one chunk.  Division-free work split over the 8 waves:
//   state rows : wave w owns (row tile w>>1, row half w&1) -> one source row pointer per lane, all k pairs;
//   weights    : NG*kc fragments of 1 KiB; wave w takes gate w % NG and every (8/NG)-th k-block.
template <int NG>
__device__ __forceinline__ void issue_chunk(const uint4* wsrc /* [NG][KB][64] base for (dir, ut) */, size_t wgate_stride16, int KB,
                                            const bf16_t* srow, int c, uint4* buf, const StepGeom& G, int wave, int lane) {
    const int kb0 = c * G.kch, kc = min(G.kch, KB - kb0);
    const int npr = (kc + 1) >> 1;
    uint4* ssec = buf + NG * G.kch * 64;
    {
        const int g = wave % NG, i0 = wave / NG, istep = 8 / NG;       // NG is 4 or 1: compile-time
        for (int i = i0; i < kc; i += istep)
            dma16_to_lds(wsrc + g * wgate_stride16 + (size_t)(kb0 + i) * 64 + lane, lds_addr_of(buf + (g * G.kch + i) * 64));
    }
    const int rtl = wave >> 1, hh = wave & 1;
    const int lrow = hh * 8 + (lane >> 3);
    const bf16_t* src = srow + (size_t)((kb0 >> 1 << 3) + ((lane & 7) ^ (lrow & 7))) * 8;     // 16-B chunk, XOR-swizzled source
    uint4* dst = ssec + ((rtl * G.npair) * 2 + hh) * 64;
    for (int pp = 0; pp < npr; ++pp)
        dma16_to_lds(src + (size_t)pp * 64, lds_addr_of(dst + pp * 128));
}
// state fragment (MFMA B operand) of k-block kbl of tile rtl for lane (frow, fq)
__device__ __forceinline__ bf16x8 state_frag(const uint4* ssec, const StepGeom& G, int rtl, int kbl, int frow, int fq) {
    const int cidx = (kbl & 1) * 4 + fq;
    uint4 v = ssec[((rtl * G.npair + (kbl >> 1)) * 2 + (frow >> 3)) * 64 + (frow & 7) * 8 + (cidx ^ (frow & 7))];
    return *(bf16x8*)&v;
}

// One chunk of the recurrent product for this wave's 16-utterance tile: acc[g] += W_g(16 units x K) . state(K x 16).
// The K range of a tile is SPLIT between the two waves that share it (wave w and w+4: k-block pairs of equal
// parity), so each SIMD runs two waves.  hipcc drains LDS reads (lgkmcnt(0)) at every loop back-edge, which
// defeats a rolled software pipeline, so the wave requests ALL fragments of up to MAXP owned pairs first
// (static register arrays, uniform guards), pays the LDS latency once, then issues its MFMAs back to back.
// Chunks hold at most 16 k-blocks when K is chunked (8 pairs -> 4 per wave); single-chunk geometries with
// more pairs take additional rounds.
template <int NG, int MAXP>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[NG], const uint4* buf, const StepGeom& G, int kc, int rtl, int khalf, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    const int gs = G.kch * 64;
    const uint4* wbase = buf + lane;                                         // gate g at + g*kch*64, k-block i at + i*64
    const uint4* sbase = buf + NG * gs + ((rtl * G.npair) * 2 + (frow >> 3)) * 64 + (frow & 7) * 8;
    const int pe = fq ^ (frow & 7), po = pe ^ 4;
    const int npr = kc >> 1;                                                 // full pairs in this chunk
    for (int p0 = khalf; p0 < npr; p0 += 2 * MAXP) {
        uint4 se[MAXP], so[MAXP], we[MAXP][NG], wo[MAXP][NG];
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const int pp = p0 + 2 * q;
            if (pp < npr) {
                se[q] = sbase[pp * 128 + pe]; so[q] = sbase[pp * 128 + po];
#pragma unroll
                for (int g = 0; g < NG; ++g) { we[q][g] = wbase[g * gs + pp * 128]; wo[q][g] = wbase[g * gs + pp * 128 + 64]; }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            if (p0 + 2 * q < npr) {
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&we[q][g], *(bf16x8*)&se[q], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&wo[q][g], *(bf16x8*)&so[q], acc[g], 0, 0, 0);
            }
        }
    }
    if ((kc & 1) && (npr & 1) == khalf) {                                    // odd tail k-block (even position of pair npr)
        const uint4 sl = sbase[npr * 128 + pe];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint4 w = wbase[g * gs + (kc - 1) * 64];
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&w, *(bf16x8*)&sl, acc[g], 0, 0, 0);
        }
    }
}

// Reduce-scatter of the two K-half partial accumulators of a tile through LDS: wave half 0 ends up with the
// full sums of cells r = 0,1 of every lane, wave half 1 with r = 2,3 (so the cell update is split too).
// xbuf: [8 waves][NG*2 floats][64 lanes].  Returns sums as out[g][rr], rr = 0,1 <-> r = 2*khalf + rr.
template <int NG>
__device__ __forceinline__ void reduce_scatter(const f32x4 (&acc)[NG], float* xbuf, int wave, int khalf, int lane, float (&out)[NG][2]) {
    float* mine = xbuf + (size_t)wave * (NG * 2 * 64) + lane;
    const int send0 = khalf ? 0 : 2;                  // the cells the partner finishes
#pragma unroll
    for (int g = 0; g < NG; ++g) { mine[(g * 2 + 0) * 64] = acc[g][send0]; mine[(g * 2 + 1) * 64] = acc[g][send0 + 1]; }
    __syncthreads();
    const float* theirs = xbuf + (size_t)(wave ^ 4) * (NG * 2 * 64) + lane;
    const int keep0 = khalf ? 2 : 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        out[g][0] = acc[g][keep0] + theirs[(g * 2 + 0) * 64];
        out[g][1] = acc[g][keep0 + 1] + theirs[(g * 2 + 1) * 64];
    }
}

struct LstmFwdArgs {
    const bf16_t* Gx;       // [S*B][ndir*H*4]  bf16, (dir, unit, gate) interleaved, bias included (row-major, time-major rows)
    const bf16_t* WhF;      // [ndir][4 gates][UT][KB][64 lanes][8]  fragment-packed W_h
    bf16_t* Yext;           // [(S+3)*B][ldy]   time block tau = t+1; block 0 = initial h, S+1.. = zero slack
    bf16_t* Ydrop;          // [S*B][ldy] or null
    float* Cs;              // lane-native, see above
    bf16_t* Gs;             // lane-native, see above (bf16)
    const int* lens;        // [B]
    const float* c0;        // [B][ndir*H] or null
    int S, B, H, H8, ndir, ldy, UT, KB, step, rb_begin, rb_count;
    float forget_bias;
    DropCfg drop;
    long long* dbg;         // diagnostic timeline (E2T_LSTM_DBG), null in production
    int gx_nt;              // persistent kernels: prefetch Gx with the non-temporal policy (set by the launcher for long sequences)
    int dbgv;               // diagnostics build only (E2T_REC_VARIANT): timing experiments that SKIP parts of the side work
};
#ifdef E2T_DEBUG
#define E2T_DBGV(p) ((p).dbgv)
#else
#define E2T_DBGV(p) 0
#endif

__global__ __launch_bounds__(512) void k_lstm_step_fwd(LstmFwdArgs p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile map: the dispatcher places workgroup id on XCD id % 8 (observed, speed only);
    // XCD x owns the unit tiles ut == x (mod 8) for every row block and direction, so its slice of
    // the W_h image (1/8 of it) stays resident in that XCD's L2 for all S steps of the sequence.
    const int RB = p.rb_count, RT = (p.B + 15) >> 4;            // this launch covers row blocks [rb_begin, rb_begin + rb_count)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ut = (slot / (RB * p.ndir)) * 8 + xcd;
    if (ut >= p.UT) return;
    const int rem = slot % (RB * p.ndir);
    const int rb = p.rb_begin + rem % RB, dir = rem / RB;
    const int s = p.step, B = p.B, H = p.H, KB = p.KB;
    const int frow = lane & 15, fq = lane >> 4;
    const int NH = p.ndir * H;
    const StepGeom G = step_geom(KB, 4, 4 * 256 + 1024);
    uint4* gx_lds = lstm_smem + (size_t)G.nbuf * G.bufsz;          // [4 tiles][2 halves][2 chunk halves][64]; then the exchange area
    long long ts[8];
#define STAMP(i) do { if (p.dbg) ts[i] = clock64(); } while (0)
    STAMP(0);

    // ---- round trip 0 (tiny, L2-resident): length of the row this lane FETCHES for (wave w moves the
    //      state / Gx rows of tile w>>1, row half w&1) and of the row it UPDATES (compute waves)
    const int fb = (rb * 4 + (wave >> 1)) * 16 + (wave & 1) * 8 + (lane >> 3);
    const int flen = (fb < B) ? p.lens[fb] : 0;
    const int rt = rb * 4 + (wave & 3);             // tile this wave multiplies (half of K) and updates (2 of 4 cells)
    const int b = rt * 16 + frow;
    const int len = (b < B) ? p.lens[b] : 0;
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);
    asm volatile("" :: "v"(flen), "v"(len) : "memory");       // pin hipcc's wait for these words HERE
    STAMP(1);

    // state-row / Gx-row sources of the fetched row.  Rows that are inactive or beyond B read ext block 0
    // (finite; their results are discarded); the backward direction's FIRST step must see a zero state and
    // reads the all-zero slack block S+1 (the block at position t = len is only re-zeroed later this pass).
    const bf16_t* srow;
    const bf16_t* gxrow;
    {
        const bool act = (fb < B) && s < flen;
        const int gb = min(fb, B - 1);
        const int tt = dir ? (flen - 1 - s) : s;
        size_t tau = 0, srb = gb;
        if (act) {
            if (dir == 1 && s == 0) { tau = (size_t)p.S + 1; srb = 0; }
            else tau = dir ? (tt + 2) : tt;
        }
        srow = p.Yext + (tau * B + srb) * p.ldy + dir * p.H8;
        gxrow = p.Gx + (act ? (((size_t)tt * B + gb) * NH + dir * H) * 4 : 0);
    }

    // ---- round trip 1 (bulk, everything in flight together, issued by all 8 waves) -------------
    const uint4* wsrc = (const uint4*)p.WhF + ((size_t)(dir * 4) * p.UT + ut) * KB * 64;
    const size_t wgs = (size_t)p.UT * KB * 64;
    issue_chunk<4>(wsrc, wgs, KB, srow, 0, lstm_smem, G, wave, lane);
    if (G.nch > 1) issue_chunk<4>(wsrc, wgs, KB, srow, 1, lstm_smem + G.bufsz, G, wave, lane);
    // Gx of the fetched rows: 8 rows x 16 units x 4 gates of bf16 = 8 x 128 B: ONE instruction, lane (row lane>>3, unit pair lane&7)
    // moves the 8 gates of units 2*(lane&7), +1 (a unit pair beyond H re-reads pair 0: never used)
    {
        const int gu = ut * 16 + 2 * (lane & 7);
        dma16_to_lds(gxrow + (gu < H ? (size_t)gu * 4 : 0), lds_addr_of(gx_lds + wave * 64));
    }
    const int khalf = wave >> 2;                      // which K half this wave multiplies / which 2 cells it finishes
    const int u0 = ut * 16 + fq * 4 + 2 * khalf;      // first of this lane's 2 units
    const int nu = min(2, H - u0);
    const bool active = s < len;
    const int t = dir ? (len - 1 - s) : s;
    float2 cprev = make_float2(0.f, 0.f);
    const size_t tile = native_tile(s, dir, rt, ut, p.ndir, RT, p.UT);
    if (active && nu > 0) {
        if (s > 0) cprev = ((const float2*)p.Cs)[(native_tile(s - 1, dir, rt, ut, p.ndir, RT, p.UT) * 2 + khalf) * 64 + lane];
        else if (p.c0) { const float* c = p.c0 + (size_t)b * NH + dir * H + u0; cprev.x = c[0]; if (nu > 1) cprev.y = c[1]; }
    }
    STAMP(2);

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < G.nch; c += 2) {
        dma_wait_all();
        STAMP(3);
        __syncthreads();
        STAMP(4);
        for (int cc = c; cc < min(c + 2, G.nch); ++cc)
            mma_chunk<4, 4>(acc, lstm_smem + (size_t)(cc & 1) * G.bufsz, G, min(G.kch, KB - cc * G.kch), wave & 3, khalf, lane);
        if (c + 2 < G.nch) {
            __syncthreads();                         // everyone is done reading both buffers
            issue_chunk<4>(wsrc, wgs, KB, srow, c + 2, lstm_smem, G, wave, lane);
            if (c + 3 < G.nch) issue_chunk<4>(wsrc, wgs, KB, srow, c + 3, lstm_smem + G.bufsz, G, wave, lane);
        }
    }
    float z[4][2];
    reduce_scatter<4>(acc, (float*)(gx_lds + 4 * 256), wave, khalf, lane, z);
    STAMP(5);

    // ---- lane-local cell update for (utterance b, units u0, u0+1) -------------------------------
    if (b >= B || nu <= 0) return;
    if (active) {
        const size_t m = (size_t)t * B + b;
        const size_t e4 = m * NH + dir * H + (u0 - 2 * khalf);      // logical index of the lane's 4-unit group
        float dsc4[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.Ydrop && p.drop.rate > 0.f) drop_scale4(p.drop.rate, key, p.drop.stream, e4, dsc4);
        float hv[2], cv[2], hd[2];
        const float cp[2] = {cprev.x, cprev.y};
        // the lane's two units fq*4 + 2*khalf, +1 are ONE 16-B slot: [fetching wave (row tile, row half)][row][unit pair]
        const uint4 graw = gx_lds[((wave & 3) * 2 + (frow >> 3)) * 64 + (frow & 7) * 8 + fq * 2 + khalf];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const unsigned g01 = rr ? graw.z : graw.x, g23 = rr ? graw.w : graw.y;
            const float4 gx = make_float4(__uint_as_float(g01 << 16), __uint_as_float(g01 & 0xFFFF0000u), __uint_as_float(g23 << 16), __uint_as_float(g23 & 0xFFFF0000u));
            const float gi = fsigmoid(z[0][rr] + gx.x);
            const float gj = ftanh(z[1][rr] + gx.y);
            const float gf = fsigmoid(z[2][rr] + gx.z + p.forget_bias);
            const float go = fsigmoid(z[3][rr] + gx.w);
            cv[rr] = fmaf(gf, cp[rr], gi * gj);           // explicit: identical rounding in the step and persistent kernels
            hv[rr] = go * ftanh(cv[rr]);
            if (rr < nu) nt_store_u2(p.Gs + (((tile * 2 + khalf) * 64 + lane) * 2 + rr) * 4, gates_pack(gi, gj, gf, go));
            hd[rr] = hv[rr] * (khalf ? dsc4[2 + rr] : dsc4[rr]);
        }
        STAMP(6);
        ((float2*)p.Cs)[(tile * 2 + khalf) * 64 + lane] = make_float2(cv[0], cv[1]);
        bf16_t* yp = p.Yext + ((size_t)(t + 1) * B + b) * p.ldy + dir * p.H8 + u0;
        bf16_t* ydp = p.Ydrop ? p.Ydrop + m * p.ldy + dir * p.H8 + u0 : nullptr;
        if (nu == 2) {
            *(ushort2*)yp = make_ushort2(f2bf(hv[0]), f2bf(hv[1]));                // re-read next step: keep in L2
            if (ydp) *(ushort2*)ydp = make_ushort2(f2bf(hd[0]), f2bf(hd[1]));
        } else {
            yp[0] = f2bf(hv[0]); if (ydp) ydp[0] = f2bf(hd[0]);
        }
        STAMP(7);
        if (p.dbg && lane == 0) {
            long long* o = p.dbg + ((size_t)(blockIdx.x * 8 + wave)) * 8;
            for (int i = 0; i < 8; ++i) o[i] = ts[i];
        }
    } else if (s < p.S) {
        // padded position s of this utterance: emit zeros (dynamic_rnn semantics)
        bf16_t* yp = p.Yext + ((size_t)(s + 1) * B + b) * p.ldy + dir * p.H8 + u0;
        bf16_t* ydp = p.Ydrop ? p.Ydrop + ((size_t)s * B + b) * p.ldy + dir * p.H8 + u0 : nullptr;
        for (int r = 0; r < nu; ++r) { yp[r] = 0; if (ydp) ydp[r] = 0; }
    }
}

// ---------------------------------------------------------------------------
// Persistent forward recurrence: ONE launch runs all S steps of a layer (both directions).
// Ownership as in k_lstm_step_fwd (workgroup = 16 units x 64 utterances x direction) but with 4 waves at ONE
// wave per SIMD, so a wave may hold 512 registers:
//   * weight-stationary in REGISTERS: the wave keeps the W_h fragments of its 16 units, all four gates, the whole
//     K range (4 x KB x 4 registers) for the entire sequence; there is no LDS image and no workgroup barrier;
//   * h is handed from step to step through a small double-buffered exchange array `hx` laid out in MFMA operand
//     order and indexed by STEP PARITY (the state of an utterance at step s+1 is what its row produced at step s,
//     whatever its time index), so a consumer load instruction is one contiguous KiB (fragment-layout loads
//     from the row-major output array touch 16 cache lines per 16 lanes and are L1-tag bound: 2.4 us vs 0.5 us)
//     and a cluster's working set (51 KiB per buffer at H=400) lives in its XCD's L2.  Loads use the sc1 policy
//     (bypass the CU's L1: other CUs wrote the data during this launch).  Step 0 reads the initial state from the
//     row-major array.  The row-major, time-indexed copy for the next layer / BPTT is written off the critical path;
//   * two accumulator sets reproduce the K-halves of the launch-per-step kernel (pairs of k-blocks of equal
//     parity), so the sums are bit-identical to it; the lane then owns all FOUR gate sums of 4 consecutive units;
//   * c lives in registers across steps (still saved per step for BPTT);
//   * h_t is exchanged between the workgroups of a "cluster" (the UT unit tiles that share a row block and a
//     direction) WITHOUT flags: every bf16 in `hx` carries a 1-bit stamp in bit 14 (free, because |h| <= 1 keeps the
//     exponent below 128) that toggles each time its buffer is rewritten; a consumer loads its fragments and retries
//     (bounded) until every value it needs shows the expected stamp, then strips the stamps.  This removes the
//     producer's store-ack wait and the separate flag round trip (~0.9 us of a 3.4 us step).  Each 8-B producer store
//     is stamped per value, so a torn 16-B read is simply seen as stale.  A buffer is only rewritten by a producer
//     that has consumed the following step from EVERY cluster member, each of which had read the buffer before
//     publishing, so nothing is overwritten early.  The protocol is placement-independent; clusters are laid on XCDs
//     (workgroup id % 8 = XCC id, read back from HW_REG_XCC_ID by a one-off probe in round 1) only for speed;
//   * everything that is not on the h_t -> h_{t+1} critical path (gate / cell saves, the Philox mask and the
//     dropped copy for the next layer) is issued AFTER the exchange store and runs while the other workgroups' stores
//     are in flight; Gx is prefetched two steps ahead, right after a step's state has landed.
// All workgroups must be co-resident (one per CU: checked on the host against the CU count); every spin is
// bounded and raises err[0] instead of hanging.
// ---------------------------------------------------------------------------
struct LstmPersistArgs {
    LstmFwdArgs a;
    bf16_t* hx;             // [2 step parities][ndir][4*ceil(B/64)][KB][64 lanes][8]  h exchange, MFMA operand order; pad lanes stay zero
    int* err;               // [1] set to 1 if a bounded spin gave up
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// DEFER (round 6 experiment, diagnostics build only -- MEASURED AND REJECTED): the dropped copy of step s (Philox mask + store) is made
// at step s+1, its Philox evaluation (336 cycles per wave, scripts/probes/philox_probe.hip) UNDER the state loads of step s+1 instead
// of in front of them.  Same arithmetic, same bits.  Phase stamps (profiles/r06_rec_sidework_probe.txt): the mask does hide under the
// loads (state landed 0.72 vs 0.68 us, no retries) and the side phase shrinks 0.6 -> 0.4 us, but the pending store in front of the
// MFMAs lengthens that phase 0.56 -> 0.80: 2.44 vs 2.36 us from step top to step bottom, 2.81 vs 2.70 us per step by the clock, the
// cfg2 train step 1.671 vs 1.562 ms.  Second form (the pending store behind the exchange store instead; E2T_REC_VARIANT=16 selects the
// first): state landed 0.72, MFMA phase still 0.68 (the carried state costs register moves around the MFMAs), gates 0.54, side 0.5:
// 2.44 vs 2.36 us, the step 1.579 vs 1.576 ms over three same-box pairs -- no gain either.  (What the side work in front of the loads
// does buy: with NOTHING between the exchange store and the loads the first attempt always finds stale stamps and pays a second
// round trip -- 2.94 us per step.)
// PIPE (round 6 experiment, diagnostics build only -- MEASURED AND REJECTED): the state fragments consumed AS THEY LAND -- VMEM returns
// in issue order, so `s_waitcnt vmcnt(KB-1-kb)` retires exactly fragment kb; its stamps go into running AND / OR words, its four MFMAs
// are issued at once (behind a sched_barrier: hipcc otherwise sinks all 52 behind the last wait) -- and the freshness check is made
// ONCE, behind the last fragment: a product on stale state is discarded and the step taken again.  Same MFMAs on the same accumulators
// in the same order: same bits.  The idea was to run the 0.56-us MFMA phase under the 0.68-us load phase.  Measured
// (profiles/r06_rec_sidework_probe.txt): loads + MFMAs 1.36 us against 0.68 + 0.56 -- a wave's thirteen fragments land TOGETHER at the
// end of one L2 round trip, there is nothing to run under; 3.23 vs 2.70 us per step.
template <int KB, bool DEFER, bool PIPE>        // k-blocks of 32 over H8: compile-time, so every fragment sits in a fixed register
__global__ __launch_bounds__(256) void k_lstm_seq_fwd_persist(LstmPersistArgs pa) {
    const LstmFwdArgs& p = pa.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = p.B, H = p.H, S = p.S;
    const int RB = (B + 63) >> 6, RT = (B + 15) >> 4;
    const int ncl = RB * p.ndir;
    const int cl = blockIdx.x % ncl, ut = blockIdx.x / ncl;      // cluster-major ids: cluster c sits on XCD c % 8
    if (ut >= p.UT) return;
    const int rb = cl % RB, dir = cl / RB;
    const int frow = lane & 15, fq = lane >> 4;
    const int NH = p.ndir * H;
    const int rt = rb * 4 + wave;
    const int b = rt * 16 + frow;
    const int bc = min(b, B - 1);
    const int len = (b < B) ? p.lens[b] : 0;
    const int u0 = ut * 16 + fq * 4;                              // this lane's 4 consecutive units (H % 4 == 0)
    const bool own = (b < B) && (u0 < H);
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);
    constexpr int npr = KB >> 1;

    // ---- once: W_h fragments of (dir, ut): [gate][k-block] ----
    bf16x8 W[KB][4];
    {
        const uint4* wsrc = (const uint4*)p.WhF + ((size_t)(dir * 4) * p.UT + ut) * KB * 64 + lane;
        const size_t wgs = (size_t)p.UT * KB * 64;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) { const uint4 v = wsrc[g * wgs + (size_t)kb * 64]; W[kb][g] = *(const bf16x8*)&v; }
    }
    float cst[4] = {0.f, 0.f, 0.f, 0.f};                          // c of this lane's 4 cells, carried in registers
    if (own && len > 0 && p.c0) { const float4 c = *(const float4*)(p.c0 + (size_t)b * NH + dir * H + u0); cst[0] = c.x; cst[1] = c.y; cst[2] = c.z; cst[3] = c.w; }
    // Gx of the lane's 4 units at its utterance's time index: 64 contiguous bytes.  Prefetched TWO steps ahead by
    // LDS-DMA into a wave-private ring of three buffers ([buffer][r][lane] float4) -- no registers are involved, so the
    // compiler cannot pull the wait for it into the critical path.  It is issued right after a step's state has landed
    // (VMEM returns in order: issued in front of the state loads it would be waited for with them) and is covered by the
    // state waits of the two following steps, which is enough for an HBM miss (long sequences: Gx no longer fits the
    // 256-MB infinity cache; one step ahead cost 4.5 instead of 3.0 us per step at S = 167).
    uint4* gxl = lstm_smem + (size_t)wave * (3 * 4 * 64);
    auto gx_load = [&](int s) {
        const bool act = own && s < len;
        const int tt = act ? (dir ? (len - 1 - s) : s) : 0;
        const bf16_t* q = p.Gx + (((size_t)tt * B + bc) * NH + dir * H + (u0 < H ? u0 : 0)) * 4;
#pragma unroll
        // Gx is read exactly once.  Where a layer's Gx exceeds the infinity cache (cfg5: 273 MB) the stream goes out non-temporal:
        // with the default policy it displaced what the step lives on and the whole train step was 4 % slower (6.94 -> 6.65 ms same
        // box; the recurrence ALONE measures the same either way); where Gx is cache-resident (cfg2: 55 MB, written by the
        // projection GEMM just before) the default policy is the faster one (2.70 vs 3.43 us per step alone, step equal)
        for (int r = 0; r < 2; ++r) {
            if (p.gx_nt) dma16_to_lds_nt(q + r * 8, lds_addr_of(gxl + ((s % 3) * 2 + r) * 64));
            else dma16_to_lds(q + r * 8, lds_addr_of(gxl + ((s % 3) * 2 + r) * 64));
        }
    };
    gx_load(0);
    if (S > 1) gx_load(1);
    // Stamp convention (see the state loads below).  Every slot of an exchange buffer ends a launch on the same stamp
    // (all producers make the same number of writes); the new launch starts on the opposite one, so leftovers -- of the
    // previous launch or of the initial fill -- never look fresh, and no flag, counter or reset pass is needed.  The
    // leftover stamps (bit q = buffer q) live in one word per cluster behind the buffers; the cluster's first workgroup
    // updates it when it is done -- a cluster cannot finish before all its members have started and read the word.
    // (Reading the leftover from the wave's own slot instead fails for waves that own none but still check others' rows.)
    const size_t hx_slot = ((((size_t)p.ndir * 0 + dir) * (RB * 4) + rt) * KB + (ut >> 1)) * 512 + (((ut & 1) * 2 + (fq >> 1)) * 16 + frow) * 8 + (fq & 1) * 4;
    const size_t hx_buf = (size_t)p.ndir * (RB * 4) * KB * 512;          // elements per step-parity buffer
    unsigned* hxw = (unsigned*)(pa.hx + 2 * hx_buf) + cl;
    const unsigned left = __builtin_amdgcn_readfirstlane(__hip_atomic_load(hxw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned base[2] = {(left & 1u) ^ 1u, ((left >> 1) & 1u) ^ 1u};
    long long pts[8];
    const long long t_entry = p.dbg ? wall_clock64() : 0;
#define PSTAMP(i) do { if (p.dbg && s == S / 2) pts[i] = wall_clock64(); } while (0)      // 100 MHz, chip-wide

    // DEFER: the previous step's h (fp32, before the mask), where its dropped copy goes, and whether there is one to make
    float hvp[4] = {0.f, 0.f, 0.f, 0.f};
    size_t ydp_off = 0;
    unsigned long long ctrp = 0ull;          // Philox counter of the lane's 4-unit group (logical element index / 4)
    int prev = 0;                            // 0 nothing pending, 1 an active cell (mask), 2 a padded position (zeros)
    const unsigned dthresh = (unsigned)(p.drop.rate * 16777216.0f);
    const float dkeep = 1.0f / (1.0f - p.drop.rate);
    const bool masked = DEFER && p.Ydrop && p.drop.rate > 0.f;          // wave-uniform
    auto ydrop_put = [&](const float (&sc)[4]) {                          // the pending dropped copy (after the state has landed)
        if (prev == 1) nt_store_bf4(p.Ydrop + ydp_off, f2bf(hvp[0] * sc[0]), f2bf(hvp[1] * sc[1]), f2bf(hvp[2] * sc[2]), f2bf(hvp[3] * sc[3]));
        else if (prev == 2) *(unsigned long long*)(p.Ydrop + ydp_off) = 0ull;
    };

    for (int s = 0; s < S; ++s) {
        PSTAMP(0);
        // ---- h_{t-1} fragments of this lane's utterance (MFMA B operand: k = kb*32 + fq*8 .. +8) ----
        const bool active = s < len;
        const int t = dir ? (len - 1 - s) : s;
        u32x4 st[KB];
        f32x4 acc[2][4];
        float dscp[4] = {1.f, 1.f, 1.f, 1.f};
        {
            if (s == 0) {
                // initial state from the row-major array: block 0 (forward), the all-zero slack block S+1 (backward);
                // rows that are inactive or beyond B read block 0 (finite, result discarded)
                size_t tau = 0, srb = bc;
                if (active && dir == 1) { tau = (size_t)S + 1; srb = 0; }
                const bf16_t* src = p.Yext + (tau * B + srb) * p.ldy + dir * p.H8 + fq * 8;
    #pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(st[kb]) : "v"(src), "i"(kb * 64) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    #pragma unroll
                for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(st[kb]));         // uses stay behind the wait
            } else if (PIPE) {
                const bf16_t* src = pa.hx + ((((size_t)((s - 1) & 1) * p.ndir + dir) * (RB * 4) + rt) * KB * 64 + lane) * 8;
                const bool tag1 = ((((s - 1) >> 1) & 1) ^ ((s - 1) & 1 ? base[1] : base[0])) != 0;      // wave-uniform
                const bool chk_last = (KB - 1) * 32 + fq * 8 < H;            // the last k-block is partly padding (never written)
                int spins = 0;
                for (;;) {
    #pragma unroll
                    for (int h = 0; h < 2; ++h)
    #pragma unroll
                        for (int g = 0; g < 4; ++g) acc[h][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int kb = 0; kb < KB; ++kb)     // the immediate offset field is 13-bit signed: one base per 4 k-blocks
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(st[kb]) : "v"(src + (kb >> 2) * 2048), "i"((kb & 3) * 1024) : "memory");
                    unsigned mand = 0xFFFFFFFFu, mor = 0u;
    #pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        // fragment kb has landed when at most KB-1-kb vector-memory operations are outstanding (everything older --
                        // the previous step's stores and prefetches -- retired before it); the register is an OUTPUT of the wait, so
                        // hipcc cannot read it earlier
                        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(st[kb]) : "n"(KB - 1 - kb) : "memory");
                        const unsigned a = st[kb][0] & st[kb][1] & st[kb][2] & st[kb][3], o = st[kb][0] | st[kb][1] | st[kb][2] | st[kb][3];
                        const bool chk = kb < KB - 1 || chk_last;
                        mand &= chk ? a : 0xFFFFFFFFu; mor |= chk ? o : 0u;
                        const u32x4 x = st[kb] & 0xBFFFBFFFu;                // stamps off (a no-op where they are clear)
    #pragma unroll
                        for (int g = 0; g < 4; ++g) acc[(kb >> 1) & 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[kb][g], *(const bf16x8*)&x, acc[(kb >> 1) & 1][g], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);                  // (hipcc otherwise sinks all 52 MFMAs behind the last wait)
                    }
                    const bool fresh = tag1 ? ((mand & 0x40004000u) == 0x40004000u) : ((mor & 0x40004000u) == 0u);
                    if (__all(fresh || !active)) break;
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;        // bounded: never hang the GPU; once any wave has given up nobody waits any more
                    if ((spins & 255) == 0 && __hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > (1 << 17)) { __hip_atomic_store(pa.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                if (p.dbg && s == S / 2) pts[1] = spins;
            } else {
                // No flags: every bf16 in the exchange buffer carries a 1-bit stamp in bit 14 (free: |h| <= 1 keeps the
                // exponent below 128).  Buffer (s-1)&1 is rewritten every other step, so its stamp toggles with (s-1)>>1;
                // the consumer simply loads and retries until every value it needs shows the expected stamp.
                const bf16_t* src = pa.hx + ((((size_t)((s - 1) & 1) * p.ndir + dir) * (RB * 4) + rt) * KB * 64 + lane) * 8;
                const bool tag1 = ((((s - 1) >> 1) & 1) ^ ((s - 1) & 1 ? base[1] : base[0])) != 0;      // wave-uniform
                const bool chk_last = (KB - 1) * 32 + fq * 8 < H;            // the last k-block is partly padding (never written)
                int spins = 0;
                for (;;) {
                    // NOTHING that touches the load destinations may sit between the issue and the wait, and the loads stay inline
                    // (not in a lambda): a copy of a load destination taken while the load is in flight is garbage -- hipcc makes
                    // such copies when it parks registers in AGPRs or gives a by-reference capture a home (seen: stale stamps,
                    // timeout; scripts/check_inflight_regs.py walks the ISA for them).  Moving the Philox mask in front of the MFMAs
                    // was measured too: +0.2 us on the MFMA phase, more retries, slower.
    #pragma unroll
                    for (int kb = 0; kb < KB; ++kb)     // the immediate offset field is 13-bit signed: one base per 4 k-blocks
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(st[kb]) : "v"(src + (kb >> 2) * 2048), "i"((kb & 3) * 1024) : "memory");
                    if (masked && spins == 0) {
                        // DEFER: the previous step's Philox mask, pure VALU on registers of its own, while the state is on its way
                        // (the counter is pinned behind the issue, the result in front of the wait: hipcc may not move it out)
                        unsigned clo = (unsigned)ctrp, chi = (unsigned)(ctrp >> 32);
                        asm volatile("" : "+v"(clo), "+v"(chi));
                        unsigned r[4];
                        philox4x32_10(clo, chi, p.drop.stream, 0u, (unsigned)key, (unsigned)(key >> 32), r);
    #pragma unroll
                        for (int i = 0; i < 4; ++i) dscp[i] = (r[i] >> 8) >= dthresh ? dkeep : 0.0f;
                        asm volatile("" :: "v"(dscp[0]), "v"(dscp[1]), "v"(dscp[2]), "v"(dscp[3]));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    #pragma unroll
                    for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(st[kb]));     // uses stay behind the wait
                    bool fresh;
                    if (tag1) {                                              // all stamps must be set
                        unsigned m = 0xFFFFFFFFu;
    #pragma unroll
                        for (int kb = 0; kb < KB - 1; ++kb) m &= st[kb][0] & st[kb][1] & st[kb][2] & st[kb][3];
                        const unsigned l = st[KB - 1][0] & st[KB - 1][1] & st[KB - 1][2] & st[KB - 1][3];
                        m &= chk_last ? l : 0xFFFFFFFFu;
                        fresh = (m & 0x40004000u) == 0x40004000u;
                    } else {                                                 // all stamps must be clear
                        unsigned m = 0u;
    #pragma unroll
                        for (int kb = 0; kb < KB - 1; ++kb) m |= st[kb][0] | st[kb][1] | st[kb][2] | st[kb][3];
                        const unsigned l = st[KB - 1][0] | st[KB - 1][1] | st[KB - 1][2] | st[KB - 1][3];
                        m |= chk_last ? l : 0u;
                        fresh = (m & 0x40004000u) == 0u;
                    }
                    // rows that are inactive at this step (or beyond B) may hold anything: their results are discarded
                    if (__all(fresh || !active)) break;
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;        // bounded: never hang the GPU; once any wave has given up nobody waits any more
                    if ((spins & 255) == 0 && __hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > (1 << 17)) { __hip_atomic_store(pa.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                if (p.dbg && s == S / 2) pts[1] = spins;
                if (tag1) {
    #pragma unroll
                    for (int kb = 0; kb < KB; ++kb) st[kb] &= 0xBFFFBFFFu;   // strip the stamps before the MFMAs
                }
            }
            PSTAMP(2);
            if (s + 2 < S) gx_load(s + 2);
            if (DEFER && p.Ydrop && own && (E2T_DBGV(p) & 16)) ydrop_put(dscp);      // (first form of the experiment: the pending store in front of the MFMAs)

            if (!(PIPE && s > 0)) {
    #pragma unroll
            for (int h = 0; h < 2; ++h)
    #pragma unroll
                for (int g = 0; g < 4; ++g) acc[h][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    #pragma unroll
            for (int pp = 0; pp < npr; ++pp) {                           // k-block pairs; parity = the K half of the step kernel
    #pragma unroll
                for (int g = 0; g < 4; ++g) acc[pp & 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[2 * pp][g], *(bf16x8*)&st[2 * pp], acc[pp & 1][g], 0, 0, 0);
    #pragma unroll
                for (int g = 0; g < 4; ++g) acc[pp & 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[2 * pp + 1][g], *(bf16x8*)&st[2 * pp + 1], acc[pp & 1][g], 0, 0, 0);
            }
            if (KB & 1) {                                                // odd tail k-block: the half that owns pair index npr
    #pragma unroll
                for (int g = 0; g < 4; ++g) acc[npr & 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[KB - 1][g], *(bf16x8*)&st[KB - 1], acc[npr & 1][g], 0, 0, 0);
            }
            }
        }
        PSTAMP(3);

        // ---- lane-local cell update for (utterance b, units u0..u0+3): critical part ----
        float gi[4], gj[4], gf[4], go[4], hv[4];
        const uint4 gxq[2] = {gxl[((s % 3) * 2 + 0) * 64 + lane], gxl[((s % 3) * 2 + 1) * 64 + lane]};     // units u0, u0+1 | u0+2, u0+3: 4 gates of bf16 each
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint4 graw = gxq[r >> 1];
            const unsigned g01 = (r & 1) ? graw.z : graw.x, g23 = (r & 1) ? graw.w : graw.y;
            const float gxv[4] = {__uint_as_float(g01 << 16), __uint_as_float(g01 & 0xFFFF0000u), __uint_as_float(g23 << 16), __uint_as_float(g23 & 0xFFFF0000u)};
            gi[r] = fsigmoid((acc[0][0][r] + acc[1][0][r]) + gxv[0]);
            gj[r] = ftanh((acc[0][1][r] + acc[1][1][r]) + gxv[1]);
            gf[r] = fsigmoid((acc[0][2][r] + acc[1][2][r]) + gxv[2] + p.forget_bias);
            go[r] = fsigmoid((acc[0][3][r] + acc[1][3][r]) + gxv[3]);
            const float cv = fmaf(gf[r], cst[r], gi[r] * gj[r]);
            hv[r] = go[r] * ftanh(cv);
            if (active) cst[r] = cv;
        }
        unsigned long long hb = 0ull;
        if (active) hb = (unsigned long long)f2bf_pk(hv[0], hv[1]) | ((unsigned long long)f2bf_pk(hv[2], hv[3]) << 32);      // hardware converter: same bits as f2bf for every non-NaN
        if (own && s + 1 < S) {
            // write-through store into the exchange buffer: unit u0 sits in k-block ut/2, k-group (ut&1)*2 + fq/2,
            // half (fq&1) of the 16-B lane slot.  EVERY owned slot is rewritten EVERY step (padded positions: zeros),
            // so all stamps of a buffer move together.
            unsigned long long* hp = (unsigned long long*)(pa.hx + (s & 1) * hx_buf + hx_slot);
            const unsigned long long stamp = ((((s >> 1) & 1) ^ (s & 1 ? base[1] : base[0])) != 0) ? 0x4000400040004000ull : 0ull;
            __hip_atomic_store(hp, hb | stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PSTAMP(4);
        PSTAMP(5);
        // ---- off the critical path: saves for BPTT, dropped copy for the next layer, Gx of the next step ----
        const int dv = E2T_DBGV(p);      // (0 in the product build: folds away)
        // DEFER, second form: the previous step's dropped copy (its mask was computed under this step's state loads) leaves HERE, behind
        // the exchange store, where a store costs the critical chain nothing -- in front of the MFMAs it lengthened that phase by 0.24 us
        if (DEFER && p.Ydrop && own && !(dv & 16)) ydrop_put(dscp);
        if (own && !(dv & 8)) {            // row-major copy for the next layer and BPTT (time block t+1); padded positions emit zeros
            const size_t blk = active ? (size_t)(t + 1) : (size_t)(s + 1);
            *(unsigned long long*)(p.Yext + (blk * B + b) * p.ldy + dir * p.H8 + u0) = hb;
        }
        if (own) {
            if (active) {
                const size_t m = (size_t)t * B + b;
                const size_t tile = native_tile(s, dir, rt, ut, p.ndir, RT, p.UT);
                if (!(dv & 2)) {
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const uint2 a = gates_pack(gi[2 * rp], gj[2 * rp], gf[2 * rp], go[2 * rp]), c = gates_pack(gi[2 * rp + 1], gj[2 * rp + 1], gf[2 * rp + 1], go[2 * rp + 1]);
                    nt_store_u4(p.Gs + ((tile * 2 + rp) * 64 + lane) * 8, make_uint4(a.x, a.y, c.x, c.y));
                }
                nt_store_f2(p.Cs + ((tile * 2 + 0) * 64 + lane) * 2, cst[0], cst[1]);
                nt_store_f2(p.Cs + ((tile * 2 + 1) * 64 + lane) * 2, cst[2], cst[3]);
                }
                if (DEFER && p.Ydrop) {
                    // made at the next step, its Philox evaluation under that step's state loads (the launcher requires H % 8 == 0
                    // and u0 is a multiple of 4: the lane's 4 elements always share one counter)
                    prev = 1; ydp_off = m * p.ldy + dir * p.H8 + u0; ctrp = (m * NH + dir * H + u0) >> 2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hvp[r] = hv[r];
                } else if (p.Ydrop && !(dv & 4)) {
                    float dsc4[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.drop.rate > 0.f && !(dv & 1)) drop_scale4(p.drop.rate, key, p.drop.stream, m * NH + dir * H + u0, dsc4);
                    nt_store_bf4(p.Ydrop + m * p.ldy + dir * p.H8 + u0, f2bf(hv[0] * dsc4[0]), f2bf(hv[1] * dsc4[1]), f2bf(hv[2] * dsc4[2]), f2bf(hv[3] * dsc4[3]));
                }
            } else if (DEFER && p.Ydrop) {
                prev = 2; ydp_off = ((size_t)s * B + b) * p.ldy + dir * p.H8 + u0;
            } else if (p.Ydrop && !(dv & 4)) {
                *(unsigned long long*)(p.Ydrop + ((size_t)s * B + b) * p.ldy + dir * p.H8 + u0) = 0ull;
            }
        }
        PSTAMP(6);
        if (p.dbg && s == S / 2 && lane == 0)
            for (int i = 0; i < 7; ++i) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = pts[i];
        if (p.dbg && s == 0 && lane == 0) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + 7] = wall_clock64() - t_entry;   // prologue + step 0
    }
    if (DEFER && p.Ydrop && own) {           // the last step's dropped copy
        float sc[4] = {1.f, 1.f, 1.f, 1.f};
        if (masked && prev == 1) {
            unsigned r[4];
            philox4x32_10((unsigned)ctrp, (unsigned)(ctrp >> 32), p.drop.stream, 0u, (unsigned)key, (unsigned)(key >> 32), r);
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[i] = (r[i] >> 8) >= dthresh ? dkeep : 0.0f;
        }
        ydrop_put(sc);
    }
    if (ut == 0 && threadIdx.x == 0) {
        // stamps the buffers are left with: buffer q was written at steps q, q+2, ... <= S-2 with stamps base, !base, ...
        unsigned nl = left;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nwr = (S - 1 > q) ? (S - q) / 2 : 0;
            if (nwr > 0) nl = (nl & ~(1u << q)) | ((((unsigned)(nwr - 1) & 1u) ^ base[q]) << q);
        }
        __hip_atomic_store(hxw, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.dbg && lane == 0) p.dbg[(size_t)(gridDim.x * 4) * 8 + (size_t)blockIdx.x * 4 + wave] = wall_clock64() - t_entry;
#undef PSTAMP
}

// (A variant with the row tile's state pulled once per workgroup and shared through LDS -- 16 utterances x 64 units, L2 -> CU
// traffic 10.6 -> 3.0 MB per step -- was built, bit-identical, and measured SLOWER: 3.20 vs 2.89 us per step; the step is bound
// by the hand-off latency, and the LDS hop + barrier add more than the lighter load burst saves.  Removed in round 3;
// DESIGN.md keeps the numbers.)

// ---------------------------------------------------------------------------
// Persistent forward recurrence, WIDE layers (14 <= KB <= 26 k-blocks, i.e. 417 <= H <= 832: the decoder).
// The W_h slice of a 64-utterance x 16-unit workgroup no longer fits a wave's registers, so the tiling is
//   workgroup = 32 utterances (2 row tiles) x 32 units (2 unit tiles) x direction, 4 waves at one wave per SIMD;
//   wave w = (unit tile w&1, K half w>>1): it keeps the W_h fragments of its unit tile, all four gates, for its
//   half of the k-blocks (4 x KH x 4 registers), multiplies BOTH row tiles with them, hands the partial sums of the
//   row tile it does not finish to the wave with the other K half (through LDS, one barrier) and finishes the
//   cells of (row tile rb*2 + (w>>1), its unit tile): same lane <-> cell map and saves as the other kernels.
// The k-blocks of a K half are the ones k_lstm_step_fwd gives that half, in its order (the host passes the two
// lists: they follow from its chunked LDS geometry), so the result is bit-identical to the launch-per-step kernel.
// Exchange, stamps, prefetch and bounded retries exactly as in k_lstm_seq_fwd_persist.
// ---------------------------------------------------------------------------
struct LstmPersistWideArgs {
    LstmFwdArgs a;
    bf16_t* hx;                     // [2][ndir][>= 2*ceil(B/32)][KB][64][8]
    int* err;
    unsigned char kbl[2][16];       // k-blocks of K half h in accumulation order
    unsigned char nkb[2];
};

template <int KH>
__global__ __launch_bounds__(256) void k_lstm_seq_fwd_persist_wide(LstmPersistWideArgs pa) {
    const LstmFwdArgs& p = pa.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = p.B, H = p.H, S = p.S, KB = p.KB;
    const int RB = (B + 31) >> 5, RT = (B + 15) >> 4, UG = (p.UT + 1) >> 1, RTP = 2 * RB;
    const int ncl = RB * p.ndir;
    const int cl = blockIdx.x % ncl, ug = blockIdx.x / ncl;      // cluster-major ids: cluster c sits on XCD c % 8
    if (ug >= UG) return;
    const int rb = cl % RB, dir = cl / RB;
    const int khalf = wave >> 1;
    const int ut = ug * 2 + (wave & 1);
    const bool tile_ok = ut < p.UT;
    const int frow = lane & 15, fq = lane >> 4;
    const int NH = p.ndir * H;
    const int rt = rb * 2 + khalf;                                // row tile whose cells this wave finishes
    const int b = rt * 16 + frow;
    const int bc = min(b, B - 1);
    const int len = (b < B) ? p.lens[b] : 0;
    int len2[2];                                                  // lengths of the rows whose state this lane loads
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) { const int rr = (rb * 2 + r2) * 16 + frow; len2[r2] = (rr < B) ? p.lens[rr] : 0; }
    const int u0 = ut * 16 + fq * 4;
    const bool own = (b < B) && tile_ok && (u0 < H);
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);
    const int nk = pa.nkb[khalf];

    // ---- once: W_h fragments of (dir, ut), this wave's K half ----
    bf16x8 W[KH][4];
    int kbj[KH];
#pragma unroll
    for (int j = 0; j < KH; ++j) {
        kbj[j] = (j < nk) ? pa.kbl[khalf][j] : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (j < nk && tile_ok) v = ((const uint4*)p.WhF)[(((size_t)(dir * 4 + g) * p.UT + ut) * KB + kbj[j]) * 64 + lane];
            W[j][g] = *(bf16x8*)&v;
        }
    }
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    if (own && len > 0 && p.c0) { const float4 c = *(const float4*)(p.c0 + (size_t)b * NH + dir * H + u0); cst[0] = c.x; cst[1] = c.y; cst[2] = c.z; cst[3] = c.w; }
    uint4* gxl = lstm_smem + (size_t)wave * (3 * 4 * 64);         // Gx ring: two steps ahead (see the narrow kernel)
    float4* xbuf0 = (float4*)(lstm_smem + 4 * 3 * 4 * 64);        // [step parity][wave][gate][lane]
    auto gx_load = [&](int s) {
        const bool act = own && s < len;
        const int tt = act ? (dir ? (len - 1 - s) : s) : 0;
        const bf16_t* q = p.Gx + (((size_t)tt * B + bc) * NH + dir * H + (own ? u0 : 0)) * 4;
#pragma unroll
        // Gx is read exactly once.  Where a layer's Gx exceeds the infinity cache (cfg5: 273 MB) the stream goes out non-temporal:
        // with the default policy it displaced what the step lives on and the whole train step was 4 % slower (6.94 -> 6.65 ms same
        // box; the recurrence ALONE measures the same either way); where Gx is cache-resident (cfg2: 55 MB, written by the
        // projection GEMM just before) the default policy is the faster one (2.70 vs 3.43 us per step alone, step equal)
        for (int r = 0; r < 2; ++r) {
            if (p.gx_nt) dma16_to_lds_nt(q + r * 8, lds_addr_of(gxl + ((s % 3) * 2 + r) * 64));
            else dma16_to_lds(q + r * 8, lds_addr_of(gxl + ((s % 3) * 2 + r) * 64));
        }
    };
    gx_load(0);
    if (S > 1) gx_load(1);
    const size_t hx_slot = ((((size_t)dir) * RTP + rt) * KB + (ut >> 1)) * 512 + (((ut & 1) * 2 + (fq >> 1)) * 16 + frow) * 8 + (fq & 1) * 4;
    const size_t hx_buf = (size_t)p.ndir * RTP * KB * 512;       // elements per step-parity buffer
    unsigned* hxw = (unsigned*)(pa.hx + 2 * hx_buf) + cl;       // leftover stamps of this cluster's buffers (see the narrow kernel)
    const unsigned left = __builtin_amdgcn_readfirstlane(__hip_atomic_load(hxw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned base[2] = {(left & 1u) ^ 1u, ((left >> 1) & 1u) ^ 1u};

    for (int s = 0; s < S; ++s) {
        const bool active = s < len;
        const int t = dir ? (len - 1 - s) : s;
        u32x4 st[2][KH];
        if (s == 0) {
            // initial state from the row-major array (block 0; the backward direction of a bidirectional layer: slack)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const int rr = min((rb * 2 + r2) * 16 + frow, B - 1);
                size_t tau = 0, srb = rr;
                if (0 < len2[r2] && dir == 1) { tau = (size_t)S + 1; srb = 0; }
                const bf16_t* src = p.Yext + (tau * B + srb) * p.ldy + dir * p.H8 + fq * 8;
#pragma unroll
                for (int j = 0; j < KH; ++j)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st[r2][j]) : "v"(src + kbj[j] * 32) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                for (int j = 0; j < KH; ++j) asm volatile("" : "+v"(st[r2][j]));
        } else {
            const bf16_t* src = pa.hx + ((s - 1) & 1) * hx_buf + ((((size_t)dir) * RTP + rb * 2) * KB * 64 + lane) * 8;
            const bool tag1 = ((((s - 1) >> 1) & 1) ^ ((s - 1) & 1 ? base[1] : base[0])) != 0;      // wave-uniform
            int spins = 0;
            for (;;) {
                // (nothing between the issue and the wait: see k_lstm_seq_fwd_persist)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int j = 0; j < KH; ++j)
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(st[r2][j]) : "v"(src + ((size_t)r2 * KB + kbj[j]) * 512) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int j = 0; j < KH; ++j) asm volatile("" : "+v"(st[r2][j]));
                bool fresh = true;
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    unsigned ma = 0xFFFFFFFFu, mo = 0u;
#pragma unroll
                    for (int j = 0; j < KH; ++j) {
                        // k-blocks beyond this half's list and the padding k-groups of the last block are never written
                        const bool chk = (j < nk) && (kbj[j] * 32 + fq * 8 < H);
                        const unsigned a = st[r2][j][0] & st[r2][j][1] & st[r2][j][2] & st[r2][j][3];
                        const unsigned o = st[r2][j][0] | st[r2][j][1] | st[r2][j][2] | st[r2][j][3];
                        ma &= chk ? a : 0xFFFFFFFFu; mo |= chk ? o : 0u;
                    }
                    const bool f = tag1 ? ((ma & 0x40004000u) == 0x40004000u) : ((mo & 0x40004000u) == 0u);
                    // rows inactive at this step may hold anything; a wave whose unit tile does not exist (odd number of
                    // unit tiles) owns no slot to take its stamp base from and multiplies zero weights: it never waits
                    fresh = fresh && (f || !(s < len2[r2]) || !tile_ok);
                }
                if (__all(fresh)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;        // bounded: never hang the GPU; once any wave has given up nobody waits any more
                if ((spins & 255) == 0 && __hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1 << 17)) { __hip_atomic_store(pa.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            if (tag1) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int j = 0; j < KH; ++j) st[r2][j] &= 0xBFFFBFFFu;   // strip the stamps before the MFMAs
            }
        }
        // fragments beyond this half's list are padding (their weights are zero) -- and must BE zero: they were loaded
        // from k-block 0 without a freshness check, and a stale value still carries the opposite stamp in bit 14, which
        // turns a saturated h = +-1.0 into +-Inf; Inf x 0 = NaN (seen after ~50 train steps, once the LSTM saturates)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int j = 0; j < KH; ++j)
                if (j >= nk) st[r2][j] = (u32x4){0u, 0u, 0u, 0u};
        if (s + 2 < S) gx_load(s + 2);       // (after the state has landed: VMEM returns in order)
        f32x4 acc[2][4];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[r2][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KH; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[r2][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[j][g], *(bf16x8*)&st[r2][j], acc[r2][g], 0, 0, 0);
        // ---- the other K half of "my" row tile comes from wave ^ 2; mine of the other row tile goes there ----
        // Double-buffered by step parity: a wave only waits for the k-blocks of ITS K half, so it may enter step s+1 (and
        // write its partials) while its partner still reads those of step s; the barrier of step s+1 then keeps it from
        // reaching step s+2's write before the partner has left step s.  (With one buffer this write-after-read race
        // corrupted the sums as soon as other kernels' traffic skewed the waves: NaN after ~50 train steps.)
        float4* xbuf = xbuf0 + (s & 1) * (4 * 4 * 64);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = khalf ? acc[0][g] : acc[1][g];
            xbuf[(wave * 4 + g) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
        }
        __syncthreads();
        float z[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 o = xbuf[((wave ^ 2) * 4 + g) * 64 + lane];
            const f32x4 a = khalf ? acc[1][g] : acc[0][g];
            z[g][0] = a[0] + o.x; z[g][1] = a[1] + o.y; z[g][2] = a[2] + o.z; z[g][3] = a[3] + o.w;
        }
        // ---- lane-local cell update for (utterance b, units u0..u0+3) ----
        float gi[4], gj[4], gf[4], go[4], hv[4];
        const uint4 gxq[2] = {gxl[((s % 3) * 2 + 0) * 64 + lane], gxl[((s % 3) * 2 + 1) * 64 + lane]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint4 graw = gxq[r >> 1];
            const unsigned g01 = (r & 1) ? graw.z : graw.x, g23 = (r & 1) ? graw.w : graw.y;
            gi[r] = fsigmoid(z[0][r] + __uint_as_float(g01 << 16));
            gj[r] = ftanh(z[1][r] + __uint_as_float(g01 & 0xFFFF0000u));
            gf[r] = fsigmoid(z[2][r] + __uint_as_float(g23 << 16) + p.forget_bias);
            go[r] = fsigmoid(z[3][r] + __uint_as_float(g23 & 0xFFFF0000u));
            const float cv = fmaf(gf[r], cst[r], gi[r] * gj[r]);
            hv[r] = go[r] * ftanh(cv);
            if (active) cst[r] = cv;
        }
        unsigned long long hb = 0ull;
        if (active) hb = (unsigned long long)f2bf_pk(hv[0], hv[1]) | ((unsigned long long)f2bf_pk(hv[2], hv[3]) << 32);      // hardware converter: same bits as f2bf for every non-NaN
        if (own && s + 1 < S) {
            const unsigned long long stamp = ((((s >> 1) & 1) ^ (s & 1 ? base[1] : base[0])) != 0) ? 0x4000400040004000ull : 0ull;
            __hip_atomic_store((unsigned long long*)(pa.hx + (s & 1) * hx_buf + hx_slot), hb | stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- off the critical path: saves for BPTT, dropped copy for the next layer, Gx of the next step ----
        if (own) {
            const size_t blk = active ? (size_t)(t + 1) : (size_t)(s + 1);
            *(unsigned long long*)(p.Yext + (blk * B + b) * p.ldy + dir * p.H8 + u0) = hb;
            if (active) {
                const size_t m = (size_t)t * B + b;
                const size_t tile = native_tile(s, dir, rt, ut, p.ndir, RT, p.UT);
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const uint2 a = gates_pack(gi[2 * rp], gj[2 * rp], gf[2 * rp], go[2 * rp]), c = gates_pack(gi[2 * rp + 1], gj[2 * rp + 1], gf[2 * rp + 1], go[2 * rp + 1]);
                    nt_store_u4(p.Gs + ((tile * 2 + rp) * 64 + lane) * 8, make_uint4(a.x, a.y, c.x, c.y));
                }
                nt_store_f2(p.Cs + ((tile * 2 + 0) * 64 + lane) * 2, cst[0], cst[1]);
                nt_store_f2(p.Cs + ((tile * 2 + 1) * 64 + lane) * 2, cst[2], cst[3]);
                if (p.Ydrop) {
                    float dsc4[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.drop.rate > 0.f) drop_scale4(p.drop.rate, key, p.drop.stream, m * NH + dir * H + u0, dsc4);
                    nt_store_bf4(p.Ydrop + m * p.ldy + dir * p.H8 + u0, f2bf(hv[0] * dsc4[0]), f2bf(hv[1] * dsc4[1]), f2bf(hv[2] * dsc4[2]), f2bf(hv[3] * dsc4[3]));
                }
            } else if (p.Ydrop) {
                *(unsigned long long*)(p.Ydrop + ((size_t)s * B + b) * p.ldy + dir * p.H8 + u0) = 0ull;
            }
        }
    }
    if (ug == 0 && threadIdx.x == 0) {
        unsigned nl = left;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nwr = (S - 1 > q) ? (S - q) / 2 : 0;
            if (nwr > 0) nl = (nl & ~(1u << q)) | ((((unsigned)(nwr - 1) & 1u) ^ base[q]) << q);
        }
        __hip_atomic_store(hxw, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct LstmBwdArgs {
    const bf16_t* WhB;      // [ndir][UT][KB4][64][8]  fragment-packed W_h^T operand (K = 4H gate columns)
    bf16_t* dG;             // [(S+1)*B][lddg]  (dir, unit, gate) interleaved, bf16, time-major rows; block S = zero slack
    const float* dY;        // [S*B][lddy] gradient wrt the (dropped) layer output, or null
    const bf16_t* Gs; const float* Cs;    // lane-native saves of the forward pass (gates bf16, cell state fp32)
    const int* lens;
    const float* c0;        // [B][ndir*H] or null
    const float* dh_final;  // [B][ndir*H] or null: gradient wrt final state h
    const float* dc_final;  // [B][ndir*H] or null
    float* dc_carry;        // [B][ndir*H] workspace (in/out)
    float* dh0;             // [B][ndir*H] out, written when step == -1 (else untouched)
    float* dc0;             // [B][ndir*H] out, written when step == -1
    int S, B, H, H8, ndir, lddg, lddy, UT, KB4, step, rb_begin, rb_count;
    DropCfg drop;
    long long* dbg;         // diagnostic timeline (E2T_LSTM_DBG), null in production
    int dbgv;               // diagnostics build only (E2T_REC_VARIANT): see LstmFwdArgs
};

// NU = unit tiles (of 16 units) per workgroup.  1: the general form.  2: large hidden sizes (H % 32 == 0, H >= 1024: the H_d = 2048
// decoder of config 4): K = 4H makes the dG rows of a row block (64 x 4H bf16 = 1 MiB at H = 2048) the bulk of what a workgroup
// pulls through LDS, and every row block is pulled by H / (16 NU) workgroups.  The chunk loop keeps at most two 64-KiB chunks in
// flight per CU (~48 GB/s per CU at the ~2 us a chunk takes to land), so the step time is the bytes per workgroup over that rate
// times the rounds of workgroups: NU = 1 -> 512 workgroups x 1.25 MiB: 62.7 us; NU = 2 -> 256 x 1.5 MiB: 41.4 us; NU = 4 -> 128 x 2
// MiB on half the CUs: 53.6 us (all measured at H = 2048, B = 256).
template <int NU>
__global__ __launch_bounds__(512) void k_lstm_step_bwd(LstmBwdArgs p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RB = p.rb_count, RT = (p.B + 15) >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int utg = (slot / (RB * p.ndir)) * 8 + xcd;         // same XCD-aware map as the forward kernel (over groups of NU unit tiles)
    if (utg * NU >= p.UT) return;
    const int ut0 = utg * NU;
    const int rem = slot % (RB * p.ndir);
    const int rb = p.rb_begin + rem % RB, dir = rem / RB;
    const int s = p.step, B = p.B, H = p.H, KB = p.KB4;
    const int frow = lane & 15, fq = lane >> 4;
    const int K4 = 4 * H;
    const int NH = p.ndir * H;
    const StepGeom G = step_geom(KB, NU, NU * 256);

    const int fb = (rb * 4 + (wave >> 1)) * 16 + (wave & 1) * 8 + (lane >> 3);       // row this lane fetches for
    const int flen = (fb < B) ? p.lens[fb] : 0;
    const int rt = rb * 4 + (wave & 3);
    const int b = rt * 16 + frow;
    const int len = (b < B) ? p.lens[b] : 0;
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);
    asm volatile("" :: "v"(flen), "v"(len) : "memory");

    // state side: dG of the step processed just before in the sweep (s+1); rows without a successor
    // step (last valid step, inactive, beyond B) read the all-zero slack block S
    size_t trow = (size_t)p.S * B;
    if (fb < B && s + 1 < flen) trow = (size_t)(dir ? (flen - 2 - s) : (s + 1)) * B + fb;
    const bf16_t* srow = p.dG + trow * p.lddg + (size_t)dir * K4;
    const uint4* wsrc = (const uint4*)p.WhB + ((size_t)dir * p.UT + ut0) * KB * 64;
    const size_t wts = (size_t)KB * 64;                       // the NU unit tiles' images follow each other
    issue_chunk<NU>(wsrc, wts, KB, srow, 0, lstm_smem, G, wave, lane);
    if (G.nch > 1) issue_chunk<NU>(wsrc, wts, KB, srow, 1, lstm_smem + G.bufsz, G, wave, lane);

    // ---- epilogue operands (lane-native, coalesced), requested in the same round trip ----------
    const int khalf = wave >> 2;
    const bool active = s >= 0 && s < len;
    const int t = dir ? (len - 1 - s) : s;
    const size_t m = active ? ((size_t)t * B + b) : 0;
    auto unit0 = [&](int j) { return (ut0 + j) * 16 + fq * 4 + 2 * khalf; };      // first of this lane's 2 units of tile j
    float4 g4[NU][2];
    float2 c_t[NU], cprev[NU], dyv[NU], dcin[NU], dhf[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        c_t[j] = cprev[j] = dyv[j] = dcin[j] = dhf[j] = make_float2(0.f, 0.f);
        g4[j][0] = g4[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int u0 = unit0(j), nu = min(2, H - u0);
        const size_t su = (size_t)b * NH + dir * H + u0;      // state index [B][ndir*H]
        auto ld2 = [&](const float* q) { float2 v = make_float2(q[0], 0.f); if (nu > 1) v.y = q[1]; return v; };
        if (active && nu > 0) {
            const size_t tile = native_tile(s, dir, rt, ut0 + j, p.ndir, RT, p.UT);
            { const uint4 graw = ((const uint4*)p.Gs)[(tile * 2 + khalf) * 64 + lane]; g4[j][0] = gates_unpack(graw.x, graw.y); g4[j][1] = gates_unpack(graw.z, graw.w); }
            c_t[j] = ((const float2*)p.Cs)[(tile * 2 + khalf) * 64 + lane];
            if (s > 0) cprev[j] = ((const float2*)p.Cs)[(native_tile(s - 1, dir, rt, ut0 + j, p.ndir, RT, p.UT) * 2 + khalf) * 64 + lane];
            else if (p.c0) cprev[j] = ld2(p.c0 + su);
            if (p.dY) dyv[j] = ld2(p.dY + m * p.lddy + dir * p.H8 + u0);
            if (s == len - 1) {
                if (p.dh_final) dhf[j] = ld2(p.dh_final + su);
                if (p.dc_final) dcin[j] = ld2(p.dc_final + su);
            } else {
                dcin[j] = ld2(p.dc_carry + su);
            }
        }
    }

    f32x4 accv[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) accv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < G.nch; c += 2) {
        dma_wait_all();
        __syncthreads();
        for (int cc = c; cc < min(c + 2, G.nch); ++cc)
            mma_chunk<NU, (NU == 1 ? 8 : 4)>(accv, lstm_smem + (size_t)(cc & 1) * G.bufsz, G, min(G.kch, KB - cc * G.kch), wave & 3, khalf, lane);
        if (c + 2 < G.nch) {
            __syncthreads();
            issue_chunk<NU>(wsrc, wts, KB, srow, c + 2, lstm_smem, G, wave, lane);
            if (c + 3 < G.nch) issue_chunk<NU>(wsrc, wts, KB, srow, c + 3, lstm_smem + G.bufsz, G, wave, lane);
        }
    }
    float rec[NU][2];
    reduce_scatter<NU>(accv, (float*)(lstm_smem + (size_t)G.nbuf * G.bufsz), wave, khalf, lane, rec);

    if (b >= B) return;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int u0 = unit0(j), nu = min(2, H - u0);
        if (nu <= 0) continue;
        const size_t su = (size_t)b * NH + dir * H + u0;
        auto ld2 = [&](const float* q) { float2 v = make_float2(q[0], 0.f); if (nu > 1) v.y = q[1]; return v; };
        if (s < 0) {
            // pseudo-step -1: gradient into the initial state (decoder <- encoder seam)
            float2 o_h, o_c;
            if (len > 0) { o_h = make_float2(rec[j][0], rec[j][1]); o_c = ld2(p.dc_carry + su); }
            else {
                o_h = p.dh_final ? ld2(p.dh_final + su) : make_float2(0.f, 0.f);
                o_c = p.dc_final ? ld2(p.dc_final + su) : make_float2(0.f, 0.f);
            }
            p.dh0[su] = o_h.x; p.dc0[su] = o_c.x;
            if (nu > 1) { p.dh0[su + 1] = o_h.y; p.dc0[su + 1] = o_c.y; }
            continue;
        }
        if (active) {
            const size_t e4 = m * NH + dir * H + (u0 - 2 * khalf);
            float dsc4[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.dY && p.drop.rate > 0.f) drop_scale4(p.drop.rate, key, p.drop.stream, e4, dsc4);
            const float ct[2] = {c_t[j].x, c_t[j].y}, cp[2] = {cprev[j].x, cprev[j].y};
            const float dy[2] = {dyv[j].x, dyv[j].y}, dci[2] = {dcin[j].x, dcin[j].y}, dhfv[2] = {dhf[j].x, dhf[j].y};
            bf16_t og[8]; float dcn[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float4 g = g4[j][rr];
                const float dh = rec[j][rr] + dhfv[rr] + dy[rr] * (khalf ? dsc4[2 + rr] : dsc4[rr]);
                const float tc = ftanh(ct[rr]);
                const float dct = dci[rr] + dh * g.w * (1.f - tc * tc);
                og[rr * 4 + 3] = f2bf(dh * tc * g.w * (1.f - g.w));            // d_o
                og[rr * 4 + 0] = f2bf(dct * g.y * g.x * (1.f - g.x));          // d_i
                og[rr * 4 + 1] = f2bf(dct * g.x * (1.f - g.y * g.y));          // d_j
                og[rr * 4 + 2] = f2bf(dct * cp[rr] * g.z * (1.f - g.z));       // d_f
                dcn[rr] = dct * g.z;
            }
            bf16_t* gp = p.dG + m * p.lddg + (size_t)dir * K4 + u0 * 4;
            if (nu == 2) {
                uint4 v;
                v.x = og[0] | ((unsigned)og[1] << 16); v.y = og[2] | ((unsigned)og[3] << 16);
                v.z = og[4] | ((unsigned)og[5] << 16); v.w = og[6] | ((unsigned)og[7] << 16);
                *(uint4*)gp = v;
            } else {
                for (int i = 0; i < 4; ++i) gp[i] = og[i];
            }
            p.dc_carry[su] = dcn[0];
            if (nu > 1) p.dc_carry[su + 1] = dcn[1];
        } else if (s < p.S) {
            bf16_t* gp = p.dG + ((size_t)s * B + b) * p.lddg + (size_t)dir * K4 + u0 * 4;
            if (nu == 2) *(uint4*)gp = make_uint4(0, 0, 0, 0);
            else { for (int i = 0; i < 4; ++i) gp[i] = 0; }
        }
    }
}

// ---------------------------------------------------------------------------
// Persistent backward recurrence (BPTT over all S steps of a layer in ONE launch).
// The recurrent product dh_rec = dG_{s+1} . W_h^T has K = 4H, so the per-step hand-off between CUs is 4x the
// forward's and a 64-utterance x 16-unit patch would pull 200 KiB per step into every CU.  Tiling here:
//   workgroup = 16 utterances x 64 units (4 unit tiles) x direction, 4 waves at one wave per SIMD;
//   MFMA phase : wave q holds K-QUARTER q of W_h^T for all 4 unit tiles in registers (4 x KQ x 4 registers) and
//                loads only its quarter of the 16 dG rows (KQ KiB per step, one contiguous KiB per instruction from
//                the step-parity exchange buffer `dgx` in MFMA operand order, sc1);
//   reduction  : the 4 partial 16x16 tiles per unit tile go through LDS (one barrier), summed as (P0+P1)+(P2+P3);
//   cell phase : wave w finishes unit tile ug*4 + w: lane (frow, fq) = utterance rt*16+frow, units u0..u0+3.
// Everything that does not depend on dh_rec -- the saved gates and cells, dY, tanh(c) and the gate-derivative
// factors -- is fetched by LDS-DMA two steps ahead (ring of three buffers) and folded into 7 factors per cell after
// the publish, so the critical path per step is: KQ stamped loads (retried until fresh), 4*KQ MFMAs, LDS reduce,
// ~12 FMAs per cell, stamped store.  The hand-off protocol is the forward kernel's (a 1-bit stamp in bit 14 of every
// bf16, bounded retries, err[0] on timeout; the exchange copy saturates at |x| < 2); dc is
// carried in registers; the row-major time-indexed dG for the weight-gradient GEMMs is written off the critical
// path.  The summation order differs from k_lstm_step_bwd (4 K-quarters vs 2 interleaved halves), so results
// agree with it to fp32 round-off, not bit for bit.
// ---------------------------------------------------------------------------
struct LstmBwdPersistArgs {
    LstmBwdArgs a;
    bf16_t* dgx;            // [2 step parities][ndir][RT][4*KQ][64 lanes][8]  dG exchange, MFMA operand order; zero-filled once
    unsigned* flags;        // [clusters][fstride]: only the last word of a row is used: the stamps the cluster's buffers were left with
    int* err;
    int fstride;            // words per cluster row
};
#define E2T_BWD_PRE16 (6 * 64)              // 16-B units of one prefetch buffer: Gs 4 KiB, Cs 1 KiB, dY 1 KiB

// DEFER (round 6 experiment, diagnostics build only -- MEASURED AND REJECTED): the factors of step s (LDS reads of the prefetched
// saves, Philox mask of dY, 4 tanh, ~60 flops per cell: 0.45 us of the wave's one instruction stream) computed UNDER the state loads
// of step s instead of behind the exchange store of step s+1, i.e. in front of those loads.  Same arithmetic, same bits.  The loads
// then leave 0.1 instead of 0.8 us behind the exchange store, ALWAYS find stale stamps and pay a second round trip: state landed
// 2.0 instead of 0.8 us, 3.92 vs 3.40 us per step (profiles/r06_rec_sidework_probe.txt).  The side work is the spacer that makes the
// first attempt succeed; it is not idle time that could be given back.
template <int KQ, bool WIDE, bool DEFER>
__global__ __launch_bounds__(256) void k_lstm_seq_bwd_persist(LstmBwdPersistArgs pa) {
    const LstmBwdArgs& p = pa.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = p.B, H = p.H, S = p.S, KB4 = p.KB4;
    constexpr int RW = WIDE ? 2 : 1, UW = WIDE ? 2 : 4;         // row tiles x unit tiles of a workgroup (RW * UW = 4 waves)
    const int RT = (B + 15) >> 4, RTG = (RT + RW - 1) / RW, RTD = RTG * RW, UG = (p.UT + UW - 1) / UW;
    const int ncl = RTG * p.ndir;
    const int cl = blockIdx.x % ncl, ug = blockIdx.x / ncl;      // cluster-major ids: cluster c sits on XCD c % 8
    if (ug >= UG) return;
    const int rg = cl % RTG, dir = cl / RTG;
    const int rt = WIDE ? rg * 2 + (wave >> 1) : rg;             // row tile whose cells this wave finishes
    const int frow = lane & 15, fq = lane >> 4;
    const int NH = p.ndir * H, K4 = 4 * H;
    const int b = rt * 16 + frow;
    const int bc = min(b, B - 1);
    const int len = (b < B) ? p.lens[b] : 0;
    const int ut = WIDE ? ug * 2 + (wave & 1) : ug * 4 + wave;    // unit tile this wave finishes
    const bool tile_ok = ut < p.UT;
    const int u0 = ut * 16 + fq * 4;
    const bool own = (b < B) && tile_ok && (u0 < H);
    const int u0c = own ? u0 : 0;
    const size_t su = (size_t)bc * NH + dir * H + u0c;            // state index [B][ndir*H]
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);
    constexpr int KBP = 4 * KQ;

    // ---- once: W_h^T fragments: K quarter `wave` for the UW unit tiles of this group ----
    bf16x8 W[UW][KQ];
#pragma unroll
    for (int u = 0; u < UW; ++u)
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int utu = ug * UW + u, kb = wave * KQ + i;
            if (utu < p.UT && kb < KB4) v = ((const uint4*)p.WhB)[(((size_t)dir * p.UT + utu) * KB4 + kb) * 64 + lane];
            W[u][i] = *(bf16x8*)&v;
        }
    uint4* pre = lstm_smem + (size_t)wave * (3 * E2T_BWD_PRE16);              // wave-private prefetch ring (two steps ahead)
    float4* part = (float4*)(lstm_smem + 4 * 3 * E2T_BWD_PRE16);              // [unit tile][source wave][lane]
    // Hand-off without flags (the forward kernels' protocol): every bf16 of the exchange copy carries a 1-bit stamp in
    // bit 14 (the top exponent bit: free for |x| < 2; the exchange copy saturates there, the row-major dG does not) that
    // toggles whenever its slot is rewritten.  A consumer loads its rows and retries until all stamps are the expected
    // ones, then strips them: no store-ack wait and no flag round trip per step.  The stamps a launch starts from are
    // the inverse of the ones the previous launch left behind, kept per buffer in one word per cluster (last word of the
    // cluster's row in `flags`) that the cluster's first workgroup updates when it is done -- a cluster cannot finish
    // before all its members have started and read the word (clusters are independent of each other).
    unsigned* state = pa.flags + (size_t)cl * pa.fstride + (pa.fstride - 1);
    const unsigned left = __builtin_amdgcn_readfirstlane(__hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned sbase[2] = {(left & 1u) ^ 1u, ((left >> 1) & 1u) ^ 1u};
    const int s_min = (p.dh0 != nullptr) ? 0 : 1;                              // steps S-1 .. s_min publish
    // stamp of the publish of step s: writes to buffer q = s&1 happen at s = smax_q, smax_q - 2, ...
    auto stamp_of = [&](int s) -> unsigned {
        const int q = s & 1, smax = (S - 1) - ((((S - 1) & 1) != q) ? 1 : 0);
        return sbase[q] ^ ((unsigned)((smax - s) >> 1) & 1u);
    };

    // operands of step s that do not depend on the recurrence, by LDS-DMA into buffer s&1 (full exec, clamped addresses)
    const unsigned pre_lds = lds_addr_of(lstm_smem) + (unsigned)wave * (3 * E2T_BWD_PRE16 * 16);    // LDS byte address (integer math:
    auto prefetch = [&](int s) {                                                                     //  no generic->LDS casts in the loop)
        if (!tile_ok) return;
        const unsigned dst = pre_lds + (unsigned)(s % 3) * (E2T_BWD_PRE16 * 16);
        // (a wide workgroup covers TWO row tiles: with an odd number of them the second one of the last group does not
        //  exist -- its lanes own nothing, but the DMA runs with full exec: clamp the tile so it reads inside the arrays.
        //  Found by tests/test_gpu_fullsize_parity.py: cfg2's decoder at B = 16 read 200 KiB past the end of Gs.)
        const int rtp = min(rt, RT - 1);
        const size_t tile = native_tile(s, dir, rtp, ut, p.ndir, RT, p.UT);
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) dma16_to_lds(p.Gs + ((tile * 2 + rp) * 64 + lane) * 8, dst + rp * 1024);
        if (s > 0) dma16_to_lds(p.Cs + native_tile(s - 1, dir, rtp, ut, p.ndir, RT, p.UT) * 256 + lane * 4, dst + 4 * 1024);
        if (p.dY) {
            const int t = (s < len) ? (dir ? (len - 1 - s) : s) : 0;
            dma16_to_lds(p.dY + ((size_t)t * B + bc) * p.lddy + dir * p.H8 + u0c, dst + 5 * 1024);
        }
    };
    // The rarely used per-utterance operands (gradients into the final state, consumed once per utterance at its own last
    // step; the initial cell state, at step 0) must not be compiler-visible global loads inside the step loop: hipcc then puts
    // `s_waitcnt vmcnt(0)` on every path of the side work, which drains this wave's exchange stores, the row-major dG stores and
    // the operand prefetch of two steps ahead EVERY step (seen in the ISA; side work 1.1 -> 0.8 us per step).  They are
    // fetched where they are needed by a load that waits for itself (one asm statement, invisible to hipcc's wait insertion):
    // a drain only at the steps that do consume them.  (Registers are not the place: 12 more push the kernel past the 368 that
    // let a 128 x 128 K-major GEMM workgroup share the CU; nor is LDS: 8 KiB more and the two no longer fit 160 KiB together.)
    auto ld4_now = [](const float* q) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
        return v;
    };
    // factors of step s (see the cell backward below): f_add = dh_final + dy*mask, k1..k5, f; c_t is carried
    float ct[4] = {0.f, 0.f, 0.f, 0.f}, dcc[4] = {0.f, 0.f, 0.f, 0.f};
    float f_add[4], k1[4], k2[4], k3[4], k4[4], k5[4], k6[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) f_add[r] = k1[r] = k2[r] = k3[r] = k4[r] = k5[r] = k6[r] = 0.f;
    auto precompute = [&](int s) {
        if (!own) return;
        const uint4* src = pre + (s % 3) * E2T_BWD_PRE16;
        float cp[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const float2* cs = (const float2*)(src + 4 * 64);
            const float2 a = cs[lane], c = cs[64 + lane];
            cp[0] = a.x; cp[1] = a.y; cp[2] = c.x; cp[3] = c.y;
        } else if (p.c0) {
            const f32x4 c = ld4_now(p.c0 + su);
            cp[0] = c[0]; cp[1] = c[1]; cp[2] = c[2]; cp[3] = c[3];
        }
        if (s < len) {
            const int t = dir ? (len - 1 - s) : s;
            const size_t m = (size_t)t * B + b;
            float dhf[4] = {0.f, 0.f, 0.f, 0.f}, dy[4] = {0.f, 0.f, 0.f, 0.f}, dsc4[4] = {1.f, 1.f, 1.f, 1.f};
            if (s == len - 1) {               // the utterance's last time step: gradients into the final state
                if (p.dh_final) { const f32x4 v = ld4_now(p.dh_final + su); dhf[0] = v[0]; dhf[1] = v[1]; dhf[2] = v[2]; dhf[3] = v[3]; }
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.dc_final) v = ld4_now(p.dc_final + su);
                dcc[0] = v[0]; dcc[1] = v[1]; dcc[2] = v[2]; dcc[3] = v[3];
            }
            if (p.dY) {
                const uint4 raw = src[5 * 64 + lane];
                dy[0] = __uint_as_float(raw.x); dy[1] = __uint_as_float(raw.y); dy[2] = __uint_as_float(raw.z); dy[3] = __uint_as_float(raw.w);
                if (p.drop.rate > 0.f && !(E2T_DBGV(p) & 1)) drop_scale4(p.drop.rate, key, p.drop.stream, m * NH + dir * H + u0, dsc4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint4 raw = src[(r >> 1) * 64 + lane];
                const float4 g4 = (r & 1) ? gates_unpack(raw.z, raw.w) : gates_unpack(raw.x, raw.y);
                const float gi = g4.x, gj = g4.y, gf = g4.z, go = g4.w;
                const float tc = ftanh(ct[r]);
                f_add[r] = dhf[r] + dy[r] * dsc4[r];
                k1[r] = go * (1.f - tc * tc);             // d c_t     += dh  * k1
                k2[r] = tc * go * (1.f - go);             // d o (pre) =  dh  * k2
                k3[r] = gj * gi * (1.f - gi);             // d i (pre) =  dct * k3
                k4[r] = gi * (1.f - gj * gj);             // d j (pre) =  dct * k4
                k5[r] = cp[r] * gf * (1.f - gf);          // d f (pre) =  dct * k5
                k6[r] = gf;                               // d c_{t-1} =  dct * f
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ct[r] = cp[r];        // c_t of step s-1 is c_{t-1} of step s, active or not
    };

    // ---- prologue: operands of the first step (s = S-1) ----
    prefetch(S - 1);
    if (own) {
        const size_t tile = native_tile(S - 1, dir, rt, ut, p.ndir, RT, p.UT);
        const float2 a = ((const float2*)p.Cs)[(tile * 2 + 0) * 64 + lane], c = ((const float2*)p.Cs)[(tile * 2 + 1) * 64 + lane];
        ct[0] = a.x; ct[1] = a.y; ct[2] = c.x; ct[3] = c.y;
    }
    dma_wait_all();
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(ct[r]));     // hipcc's wait for the prologue's loads HERE, not at a first use inside the loop
    precompute(S - 1);
    if (S > 1) prefetch(S - 2);
    long long pts[8];
#define PSTAMP(i) do { if (p.dbg && s == S / 2) pts[i] = wall_clock64(); } while (0)

    const bool want0 = p.dh0 != nullptr;          // pseudo-step -1: gradient into the initial state (decoder <- encoder seam)
    for (int k = 0; k < S + (want0 ? 1 : 0); ++k) {
        const int s = S - 1 - k;
        PSTAMP(0);
        const bool active = s >= 0 && s < len;
        f32x4 rec = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (k > 0) {
            // ---- this wave's K quarter of the dG rows of step s+1 (rows without a successor step hold zeros), one row
            //      tile at a time (a wide layer's quarter is 25 KiB per tile: both would not fit the registers); loaded
            //      until every stamp is the one step s+1 was published with ----
            const unsigned E = stamp_of(s + 1) ? 0x40004000u : 0u;
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r2 = 0; r2 < RW; ++r2) {
                u32x4 st[KQ];
                const bf16_t* src = pa.dgx + (((((size_t)((s + 1) & 1) * p.ndir + dir) * RTD + rg * RW + r2) * KBP + wave * KQ) * 64 + lane) * 8;
                const bool rowok = (rg * RW + r2) * 16 + frow < B;             // rows beyond the batch are never published
                int spins = 0;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < KQ; ++i)        // the immediate offset field is 13-bit signed: one base per 4 k-blocks
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(st[i]) : "v"(src + (i >> 2) * 2048), "i"((i & 3) * 1024) : "memory");
                    if (DEFER && r2 == 0 && spins == 0 && s >= 0) {
                        // this step's factors while the state is on its way: LDS reads + VALU on registers of their own (their operands
                        // were prefetched two steps ago and have landed: the previous step's wait covered them); the results are
                        // pinned in front of the wait so that hipcc cannot sink the arithmetic behind it
                        precompute(s);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            asm volatile("" :: "v"(f_add[r]), "v"(k1[r]), "v"(k2[r]), "v"(k3[r]), "v"(k4[r]), "v"(k5[r]), "v"(k6[r]), "v"(ct[r]));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < KQ; ++i) asm volatile("" : "+v"(st[i]));
                    unsigned bad = 0u;
#pragma unroll
                    for (int i = 0; i < KQ; ++i) {
                        if (wave * KQ + i < KB4) {      // (k-blocks beyond 4H are never published: zero weights, zero data)
                            st[i] = (u32x4){st[i][0] ^ E, st[i][1] ^ E, st[i][2] ^ E, st[i][3] ^ E};      // fresh: bit 14 now clear
                            bad |= st[i][0] | st[i][1] | st[i][2] | st[i][3];
                        } else {
                            st[i] = (u32x4){0u, 0u, 0u, 0u};
                        }
                    }
                    if (!__any(rowok && (bad & 0x40004000u) != 0u)) break;
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;        // bounded: never hang the GPU; once any wave has given up nobody waits any more
                    if ((spins & 255) == 0 && __hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > (1 << 16)) {
                        // err[0] = code; err[1..] = who waited at which step (first reporter wins): diagnostics for the host
                        if (lane == 0 && atomicCAS((int*)pa.err, 0, 3) == 0) {
                            pa.err[1] = blockIdx.x; pa.err[2] = wave; pa.err[3] = k; pa.err[4] = r2; pa.err[5] = (int)E; pa.err[6] = (int)left;
                        }
                        break;
                    }
                }
                if (r2 == 0) { PSTAMP(1); PSTAMP(2); if (s > 1) prefetch(s - 2); }    // two steps ahead: covered by the state waits of the next two steps
#pragma unroll
                for (int i = 0; i < KQ; ++i)
#pragma unroll
                    for (int u = 0; u < UW; ++u) acc[r2 * UW + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[u][i], *(bf16x8*)&st[i], acc[r2 * UW + u], 0, 0, 0);
            }
            // ---- reduce the 4 K-quarter partials of every (row tile, unit tile) through LDS; tile index == finishing wave ----
#pragma unroll
            for (int u = 0; u < 4; ++u) part[(u * 4 + wave) * 64 + lane] = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
            __syncthreads();
            const float4 p0 = part[(wave * 4 + 0) * 64 + lane], p1 = part[(wave * 4 + 1) * 64 + lane];
            const float4 p2 = part[(wave * 4 + 2) * 64 + lane], p3 = part[(wave * 4 + 3) * 64 + lane];
            rec = (f32x4){(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)};
        } else {
            if (s > 1) prefetch(s - 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // first step only: the operands of step S-2 (issued in the prologue)
        }
        PSTAMP(3);
        if (s < 0) {
            if (own) {
                float4 oh = make_float4(rec[0], rec[1], rec[2], rec[3]), oc = make_float4(dcc[0], dcc[1], dcc[2], dcc[3]);
                if (len == 0) {
                    oh = make_float4(0.f, 0.f, 0.f, 0.f); oc = oh;
                    if (p.dh_final) { const f32x4 v = ld4_now(p.dh_final + su); oh = make_float4(v[0], v[1], v[2], v[3]); }
                    if (p.dc_final) { const f32x4 v = ld4_now(p.dc_final + su); oc = make_float4(v[0], v[1], v[2], v[3]); }
                }
                *(float4*)(p.dh0 + su) = oh; *(float4*)(p.dc0 + su) = oc;
            }
            break;
        }
        // ---- cell backward for (utterance b, units u0..u0+3): the part that needs dh_rec ----
        uint4 og0 = make_uint4(0u, 0u, 0u, 0u), og1 = make_uint4(0u, 0u, 0u, 0u);      // (i,j,f,o) x units 0,1 / units 2,3
        if (own && active) {
            unsigned ogp[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = rec[r] + f_add[r];
                const float dct = fmaf(dh, k1[r], dcc[r]);
                ogp[r * 2 + 0] = f2bf_pk(dct * k3[r], dct * k4[r]);          // hardware converter: same bits as f2bf for every non-NaN
                ogp[r * 2 + 1] = f2bf_pk(dct * k5[r], dh * k2[r]);
                dcc[r] = dct * k6[r];
            }
            og0 = make_uint4(ogp[0], ogp[1], ogp[2], ogp[3]);
            og1 = make_uint4(ogp[4], ogp[5], ogp[6], ogp[7]);
        }
        // nothing of this step is ever waited for from here on: the stores below are fire-and-forget, and the operands the
        // side work reads were fetched two steps ago
        if (s > 0 || want0) {
            if (own) {
                // exchange copy for the next step's consumers: gate columns u0*4 .. u0*4+15 = k-block ut*2 + fq/2,
                // k-groups (fq&1)*2 and +1; padded positions publish zeros.  Write-through (sc1) stores.  |x| >= 2 (or
                // inf / nan) saturates to +-1.992 in THIS copy so that bit 14 is free for the stamp.
                u32x4* hp = (u32x4*)(pa.dgx + (((((size_t)(s & 1) * p.ndir + dir) * RTD + rt) * KBP + ut * 2 + (fq >> 1)) * 64 + (fq & 1) * 32 + frow) * 8);
                const unsigned PS = stamp_of(s) ? 0x40004000u : 0u;
                unsigned x[8] = {og0.x, og0.y, og0.z, og0.w, og1.x, og1.y, og1.z, og1.w};
                unsigned big = 0u;
#pragma unroll
                for (int i = 0; i < 8; ++i) big |= x[i];
                if (__any((big & 0x40004000u) != 0u)) {
                    // |x| >= 2 somewhere in this wave's publish.  Finite values saturate in the exchange copy (counted in err[8]:
                    // the host warns); a NaN or an infinity must not be turned into a finite gradient -- it invalidates the step
                    // like a timed-out wait does (err[0]: the optimiser kernels skip their update, check_sync() raises)
                    unsigned nonfinite = 0u;
#pragma unroll
                    for (int i = 0; i < 8; ++i) nonfinite |= ((x[i] & 0x7F807F80u) + 0x00800080u) & 0x80008000u;     // a half with all exponent bits set carries into its sign bit
                    if (__any(nonfinite != 0u)) { if (lane == 0) __hip_atomic_fetch_or((unsigned*)pa.err, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (1 | 7 = 7: reported as non-finite)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = (x[i] | (((x[i] & 0x40004000u) >> 14) * 0x3FFFu)) & 0xBFFFBFFFu;
                    if (lane == 0) atomicAdd(pa.err + 8, 1);     // visible to the host: recurrent gate gradients were clipped
                }
                const u32x4 v0 = (u32x4){x[0] | PS, x[1] | PS, x[2] | PS, x[3] | PS}, v1 = (u32x4){x[4] | PS, x[5] | PS, x[6] | PS, x[7] | PS};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:256 sc1" :: "v"(hp), "v"(v0), "v"(v1) : "memory");
            }
            PSTAMP(4);
        }
        PSTAMP(5);
        // ---- off the critical path: row-major dG for the weight-gradient GEMMs, factors of the next step ----
        if (own && !(E2T_DBGV(p) & 2)) {
            const int t = dir ? (len - 1 - s) : s;
            uint4* gp = (uint4*)(p.dG + ((size_t)(active ? t : s) * B + b) * p.lddg + (size_t)dir * K4 + u0 * 4);
            gp[0] = og0; gp[1] = og1;
        }
        if (!DEFER && s > 0 && !(E2T_DBGV(p) & 4)) precompute(s - 1);        // (a first look at the next step's flags from here was measured: slower,
                                             //  hipcc waits for the loads at once when registers are this tight)
        PSTAMP(6);
        if (p.dbg && s == S / 2 && lane == 0)
            for (int i = 0; i < 7; ++i) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = pts[i];
    }
    if (ug == 0 && threadIdx.x == 0) {
        // stamps the buffers are left with (= those of their last publish, steps s_min and s_min + 1)
        unsigned nl = left;
        for (int sl = s_min; sl <= s_min + 1 && sl <= S - 1; ++sl) nl = (nl & ~(1u << (sl & 1))) | (stamp_of(sl) << (sl & 1));
        __hip_atomic_store(state, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef PSTAMP
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static int set_big_lds(const void* fn) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return E2T_ERR_HIP; }
    return E2T_OK;
}

extern "C" int e2t_lstm_seq_fwd(const e2t_lstm_desc* d, const void* Gx, const void* WhF, void* Yext,
                                void* Ydrop, float* Cs, void* Gs, const int32_t* lens, const float* c0,
                                int step_begin, int step_end, void* stream) {
    E2T_CHECK_ARG(d && Gx && WhF && Yext && Cs && Gs && lens);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && d->H > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(d->H % 2 == 0 && d->ldy % 8 == 0 && d->ldy >= d->ndir * ((d->H + 7) / 8) * 8);
    E2T_CHECK_ARG(0 <= step_begin && step_begin <= step_end && step_end <= d->S);
    static const int attr_rc = set_big_lds((const void*)k_lstm_step_fwd);          // (thread-safe one-time init)
    if (attr_rc) return attr_rc;
    LstmFwdArgs p{};
    p.Gx = (const bf16_t*)Gx; p.WhF = (const bf16_t*)WhF; p.Yext = (bf16_t*)Yext; p.Ydrop = (bf16_t*)Ydrop;
    p.Cs = Cs; p.Gs = (bf16_t*)Gs; p.lens = lens; p.c0 = c0;
    p.S = d->S; p.B = d->B; p.H = d->H; p.H8 = (d->H + 7) / 8 * 8; p.ndir = d->ndir; p.ldy = d->ldy;
    p.UT = (d->H + 15) / 16; p.KB = (p.H8 + 31) / 32;
    p.dbg = (long long*)e2t_dbg_ptr("E2T_LSTM_DBG");
    p.forget_bias = d->forget_bias;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    const StepGeom G = step_geom(p.KB, 4, 4 * 256 + 1024);
    const size_t lds = ((size_t)G.nbuf * G.bufsz + 4 * 256 + 1024) * 16;
    const int nrb = (d->B + 63) / 64;
    p.rb_begin = d->rb_count > 0 ? d->rb_begin : 0;
    p.rb_count = d->rb_count > 0 ? d->rb_count : nrb;
    E2T_CHECK_ARG(p.rb_begin >= 0 && p.rb_begin + p.rb_count <= nrb);
    dim3 grid(8 * ((p.UT + 7) / 8) * p.rb_count * d->ndir);      // XCD-major tile map, see kernel
    for (int s = step_begin; s < step_end; ++s) {
        p.step = s;
        hipLaunchKernelGGL(k_lstm_step_fwd, grid, dim3(512), lds, (hipStream_t)stream, p);
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

extern "C" int e2t_lstm_seq_fwd_persistent(const e2t_lstm_desc* d, const void* Gx, const void* WhF, void* Yext, void* Ydrop,
                                           float* Cs, void* Gs, const int32_t* lens, const float* c0, void* hx,
                                           int32_t* err, int num_cus, void* stream) {
    E2T_CHECK_ARG(d && Gx && WhF && Yext && Cs && Gs && lens && hx && err);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && d->H > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(d->H % 2 == 0 && d->ldy % 8 == 0 && d->ldy >= d->ndir * ((d->H + 7) / 8) * 8);
    LstmPersistArgs pa{};
    LstmFwdArgs& p = pa.a;
    p.Gx = (const bf16_t*)Gx; p.WhF = (const bf16_t*)WhF; p.Yext = (bf16_t*)Yext; p.Ydrop = (bf16_t*)Ydrop;
    p.Cs = Cs; p.Gs = (bf16_t*)Gs; p.lens = lens; p.c0 = c0;
    p.S = d->S; p.B = d->B; p.H = d->H; p.H8 = (d->H + 7) / 8 * 8; p.ndir = d->ndir; p.ldy = d->ldy;
    p.UT = (d->H + 15) / 16; p.KB = (p.H8 + 31) / 32;
    p.forget_bias = d->forget_bias;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    pa.hx = (bf16_t*)hx; pa.err = err;
    p.dbg = (long long*)e2t_dbg_ptr("E2T_LSTM_DBG");
    p.dbgv = e2t_dbg_int("E2T_REC_VARIANT", 0);
    p.gx_nt = (size_t)d->S * d->B * d->ndir * d->H * 4 * sizeof(bf16_t) > ((size_t)128 << 20);     // beyond half the 256-MB infinity cache
    if (p.KB > 13 && p.KB <= 26 && d->H % 8 == 0) {
        // wide layer: 32 x 32 workgroups, K halves in the accumulation order of k_lstm_step_fwd (its LDS chunk geometry)
        LstmPersistWideArgs pw{};
        pw.a = p; pw.hx = (bf16_t*)hx; pw.err = err;
        const StepGeom G = step_geom(p.KB, 4, 4 * 256 + 1024);
        int n[2] = {0, 0};
        for (int c = 0; c < G.nch; ++c) {
            const int kb0 = c * G.kch, kc = std::min(G.kch, p.KB - kb0), npr = kc >> 1;
            for (int pp = 0; pp < npr; ++pp) { pw.kbl[pp & 1][n[pp & 1]++] = (unsigned char)(kb0 + 2 * pp); pw.kbl[pp & 1][n[pp & 1]++] = (unsigned char)(kb0 + 2 * pp + 1); }
            if (kc & 1) pw.kbl[npr & 1][n[npr & 1]++] = (unsigned char)(kb0 + kc - 1);
        }
        pw.nkb[0] = (unsigned char)n[0]; pw.nkb[1] = (unsigned char)n[1];
        const int KH = std::max(n[0], n[1]);
        const int nwgw = ((d->B + 31) / 32) * d->ndir * ((p.UT + 1) / 2);
        if (KH > 13 || nwgw > num_cus) {
            e2t_set_error("persistent recurrence not applicable (H=%d, %d workgroups, %d CUs)", d->H, nwgw, num_cus);
            return E2T_ERR_ARG;
        }
        const size_t ldsw = (size_t)(4 * 3 * 4 * 64 + 2 * 4 * 4 * 64) * 16;
#define E2T_PERSIST_CASE(K) case K: hipLaunchKernelGGL(k_lstm_seq_fwd_persist_wide<K>, dim3(nwgw), dim3(256), ldsw, (hipStream_t)stream, pw); break;
        switch (KH) { E2T_PERSIST_CASE(7) E2T_PERSIST_CASE(8) E2T_PERSIST_CASE(9) E2T_PERSIST_CASE(10) E2T_PERSIST_CASE(11) E2T_PERSIST_CASE(12) E2T_PERSIST_CASE(13) }
#undef E2T_PERSIST_CASE
        E2T_LAUNCH_CHECK();
        return E2T_OK;
    }
    const int ncl = ((d->B + 63) / 64) * d->ndir;
    const int nwg = ncl * p.UT;
    // every workgroup must be resident at once (1 per CU), and the W_h fragments of a unit tile must fit the
    // wave's registers (13 k-blocks x 4 gates x 4 registers)
    if (p.KB > 13 || d->H % 8 != 0 || nwg > num_cus) {
        e2t_set_error("persistent recurrence not applicable (H=%d, %d workgroups, %d CUs)", d->H, nwg, num_cus);
        return E2T_ERR_ARG;
    }
#define E2T_PERSIST_CASE(K) case K: hipLaunchKernelGGL((k_lstm_seq_fwd_persist<K, E2T_FWD_DEFER, E2T_FWD_PIPE>), dim3(nwg), dim3(256), 4 * 3 * 4 * 64 * 16, (hipStream_t)stream, pa); break;
#define E2T_PERSIST_CASES switch (p.KB) { \
        E2T_PERSIST_CASE(1) E2T_PERSIST_CASE(2) E2T_PERSIST_CASE(3) E2T_PERSIST_CASE(4) E2T_PERSIST_CASE(5) \
        E2T_PERSIST_CASE(6) E2T_PERSIST_CASE(7) E2T_PERSIST_CASE(8) E2T_PERSIST_CASE(9) E2T_PERSIST_CASE(10) \
        E2T_PERSIST_CASE(11) E2T_PERSIST_CASE(12) E2T_PERSIST_CASE(13) }
    // (DEFER is a measured REJECTION, kept in the diagnostics build so that it can be re-taken: scripts/probe_rec_sidework.py)
#define E2T_FWD_PIPE_PRODUCT false
#ifdef E2T_DEBUG
    if (e2t_dbg_int("E2T_FWD_DEFER", 0) != 0) {
#define E2T_FWD_DEFER true
#define E2T_FWD_PIPE false
        E2T_PERSIST_CASES
#undef E2T_FWD_PIPE
#undef E2T_FWD_DEFER
    } else if (e2t_dbg_int("E2T_FWD_PIPE", E2T_FWD_PIPE_PRODUCT) != (E2T_FWD_PIPE_PRODUCT ? 1 : 0)) {
#define E2T_FWD_DEFER false
#define E2T_FWD_PIPE (!E2T_FWD_PIPE_PRODUCT)
        E2T_PERSIST_CASES
#undef E2T_FWD_PIPE
#undef E2T_FWD_DEFER
    } else
#endif
    {
#define E2T_FWD_DEFER false
#define E2T_FWD_PIPE E2T_FWD_PIPE_PRODUCT
        E2T_PERSIST_CASES
#undef E2T_FWD_PIPE
#undef E2T_FWD_DEFER
    }
#undef E2T_PERSIST_CASES
#undef E2T_PERSIST_CASE
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

extern "C" int e2t_lstm_seq_bwd(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY,
                                int lddy, const void* Gs, const float* Cs, const int32_t* lens, const float* c0,
                                const float* dh_final, const float* dc_final, float* dc_carry, float* dh0,
                                float* dc0, void* stream) {
    E2T_CHECK_ARG(d && WhB && dG && Gs && Cs && lens && dc_carry);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && d->H > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(d->H % 2 == 0 && lddg % 8 == 0 && lddg >= d->ndir * 4 * d->H);
    E2T_CHECK_ARG((dh0 == nullptr) == (dc0 == nullptr));
    static const int attr_rc = set_big_lds((const void*)k_lstm_step_bwd<1>) | set_big_lds((const void*)k_lstm_step_bwd<2>);
    if (attr_rc) return attr_rc;
    LstmBwdArgs p{};
    p.WhB = (const bf16_t*)WhB; p.dG = (bf16_t*)dG; p.dY = dY; p.Gs = (const bf16_t*)Gs; p.Cs = Cs; p.lens = lens; p.c0 = c0;
    p.dh_final = dh_final; p.dc_final = dc_final; p.dc_carry = dc_carry; p.dh0 = dh0; p.dc0 = dc0;
    p.S = d->S; p.B = d->B; p.H = d->H; p.H8 = (d->H + 7) / 8 * 8; p.ndir = d->ndir;
    p.lddg = lddg; p.lddy = lddy;
    p.UT = (d->H + 15) / 16; p.KB4 = (4 * d->H + 31) / 32;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    // 32 units per workgroup where the dG rows dominate the traffic (see the kernel)
    const bool wide_units = d->H % 32 == 0 && d->H >= 1024;
    const int NU = wide_units ? 2 : 1;
    const StepGeom G = step_geom(p.KB4, NU, NU * 256);
    const size_t lds = ((size_t)G.nbuf * G.bufsz + NU * 256) * 16;
    const int nrb = (d->B + 63) / 64;
    p.rb_begin = d->rb_count > 0 ? d->rb_begin : 0;
    p.rb_count = d->rb_count > 0 ? d->rb_count : nrb;
    E2T_CHECK_ARG(p.rb_begin >= 0 && p.rb_begin + p.rb_count <= nrb);
    const int UG = (p.UT + NU - 1) / NU;
    dim3 grid(8 * ((UG + 7) / 8) * p.rb_count * d->ndir);        // XCD-major tile map, see kernel
    for (int s = d->S - 1; s >= (dh0 ? -1 : 0); --s) {
        p.step = s;
        if (NU == 2) hipLaunchKernelGGL(k_lstm_step_bwd<2>, grid, dim3(512), lds, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(k_lstm_step_bwd<1>, grid, dim3(512), lds, (hipStream_t)stream, p);
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

extern "C" int e2t_lstm_seq_bwd_persistent(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY,
                                           int lddy, const void* Gs, const float* Cs, const int32_t* lens,
                                           const float* c0, const float* dh_final, const float* dc_final, float* dh0,
                                           float* dc0, void* dgx, uint32_t* flags, int32_t* err, int num_cus, void* stream) {
    E2T_CHECK_ARG(d && WhB && dG && Gs && Cs && lens && dgx && flags && err);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && d->H > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(lddg % 8 == 0 && lddg >= d->ndir * 4 * d->H);
    E2T_CHECK_ARG((dh0 == nullptr) == (dc0 == nullptr));
    LstmBwdPersistArgs pa{};
    LstmBwdArgs& p = pa.a;
    p.WhB = (const bf16_t*)WhB; p.dG = (bf16_t*)dG; p.dY = dY; p.Gs = (const bf16_t*)Gs; p.Cs = Cs; p.lens = lens; p.c0 = c0;
    p.dh_final = dh_final; p.dc_final = dc_final; p.dh0 = dh0; p.dc0 = dc0;
    p.S = d->S; p.B = d->B; p.H = d->H; p.H8 = (d->H + 7) / 8 * 8; p.ndir = d->ndir;
    p.lddg = lddg; p.lddy = lddy;
    p.UT = (d->H + 15) / 16; p.KB4 = (4 * d->H + 31) / 32;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    p.dbg = (long long*)e2t_dbg_ptr("E2T_LSTM_DBG");
    p.dbgv = e2t_dbg_int("E2T_REC_VARIANT", 0);
    // (the BPTT's prefetch of the saved gates / cells / dY with the non-temporal policy, and its row-major dG stores, were measured
    //  at cfg5: no difference)
    pa.dgx = (bf16_t*)dgx; pa.flags = flags; pa.err = err;
    const int RT = (d->B + 15) / 16;
    const int kq = (p.KB4 + 3) / 4;
    const bool wide = kq > 13;
    const int KQ = e2t_bwd_persist_kq(d->H);
    // every workgroup must be resident at once (1 per CU); a K quarter of W_h^T for the workgroup's unit tiles must fit a
    // wave's registers (4 tiles x 13 k-blocks, or 2 tiles x 25)
    const int nwg = wide ? ((RT + 1) / 2) * d->ndir * ((p.UT + 1) / 2) : RT * d->ndir * ((p.UT + 3) / 4);
    if (KQ == 0 || d->H % 8 != 0 || nwg > num_cus) {
        e2t_set_error("persistent BPTT not applicable (H=%d, %d workgroups, %d CUs)", d->H, nwg, num_cus);
        return E2T_ERR_ARG;
    }
    pa.fstride = wide ? 128 : 32;
    const size_t lds = (size_t)(4 * 3 * E2T_BWD_PRE16 + 16 * 64) * 16;        // prefetch rings, K-quarter partials
#define E2T_PERSIST_CASE(K, W) case K: hipLaunchKernelGGL((k_lstm_seq_bwd_persist<K, W, E2T_BWD_DEFER>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, pa); break;
#define E2T_PERSIST_CASES \
    if (!wide) { \
        switch (KQ) { \
            E2T_PERSIST_CASE(1, false) E2T_PERSIST_CASE(2, false) E2T_PERSIST_CASE(3, false) E2T_PERSIST_CASE(4, false) E2T_PERSIST_CASE(5, false) \
            E2T_PERSIST_CASE(6, false) E2T_PERSIST_CASE(7, false) E2T_PERSIST_CASE(8, false) E2T_PERSIST_CASE(9, false) E2T_PERSIST_CASE(10, false) \
            E2T_PERSIST_CASE(11, false) E2T_PERSIST_CASE(12, false) E2T_PERSIST_CASE(13, false) \
        } \
    } else { \
        switch (KQ) { E2T_PERSIST_CASE(16, true) E2T_PERSIST_CASE(20, true) E2T_PERSIST_CASE(25, true) } \
    }
    // (DEFER is a measured REJECTION, kept in the diagnostics build so that it can be re-taken: scripts/probe_rec_sidework.py)
#ifdef E2T_DEBUG
    if (e2t_dbg_int("E2T_BWD_DEFER", 0) != 0) {
#define E2T_BWD_DEFER true
        E2T_PERSIST_CASES
#undef E2T_BWD_DEFER
    } else
#endif
    {
#define E2T_BWD_DEFER false
        E2T_PERSIST_CASES
#undef E2T_BWD_DEFER
    }
#undef E2T_PERSIST_CASES
#undef E2T_PERSIST_CASE
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

// k-blocks per K quarter the persistent BPTT kernel is instantiated for (dgx holds 4x this many per row tile); 0 = none
extern "C" int e2t_bwd_persist_kq(int H) {
    const int kq = ((4 * H + 31) / 32 + 3) / 4;
    if (kq <= 13) return kq;
    if (kq <= 16) return 16;
    if (kq <= 20) return 20;
    if (kq <= 25) return 25;
    return 0;
}
