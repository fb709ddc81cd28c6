// Weight-stationary persistent recurrences for LARGE hidden sizes (BASELINE.json config 4: H = 1024, B = 256 --
// "MFMA-bound recurrent GEMMs"; reference rows a7: SequenceNetwork._encode_sequences, ecog2txt/trainers.py:821-823,
// 4-gate packing trainers.py:527-529).
//
// At H = 1024 one direction's W_h is 8 MiB of bf16: it fits the chip's register files exactly once (256 CUs x
// 256 KiB of the 512 KiB register file each), but only if NO weight is replicated -- the narrow kernels of lstm.hip keep
// a copy of a unit tile's weights in each of a workgroup's four waves (one per row tile) and therefore stop at H = 416
// (832 for the 32 x 32 tiling).  Here the four waves of a workgroup hold DIFFERENT weights and share the state:
//
//   forward   workgroup = 64 utterances (4 row tiles) x 32 units x direction; wave w owns units w*8 .. w*8+7, all four
//             gates, the WHOLE K range: 2 A-tiles x (H/32) k-blocks x 4 registers (256 at H = 1024).  The A-tile rows are
//             (unit, gate) interleaved -- 16 consecutive columns of the gate-interleaved master -- so a lane's
//             accumulator f32x4 holds the four gates of ONE cell: lane (frow, fq), row tile rt, tile a  <->  utterance
//             rt*16 + frow, unit w*8 + a*4 + fq.  The state of the workgroup's 64 utterances (64 x H bf16 = 128 KiB) is
//             pulled ONCE per step into LDS by LDS-DMA (each wave one row tile) and read by all four waves: L2 -> CU
//             traffic per step is (#unit groups) x state = 32 x 0.5 MiB per direction instead of 64 x.
//   backward  workgroup = 64 utterances x 32 units x direction; wave q owns K-QUARTER q (K = 4H gate columns) of W_h^T
//             for the two unit tiles: 2 x (H/32) k-blocks x 4 registers.  The dG rows (64 x 4H bf16 = 512 KiB per step and
//             workgroup -- K = 4H makes the hand-off 4x the forward's) are streamed through wave-private LDS rings of
//             16-KiB chunks; the four partial sums per cell are reduced through LDS.
//
// Hand-off between the CUs inside the launch: every producer wave owns ONE flag word that counts the steps it has
// published (over all launches: nothing is ever reset); data goes out with write-through (sc1) stores, the wave then
// waits for their acknowledgement and bumps its flag; a consumer polls the 128 words of its cluster (32 unit groups x
// 4 waves) with two coalesced sc1 loads and a wave vote, then pulls the state with sc1 LDS-DMA.  (The narrow kernels
// carry a 1-bit stamp in every bf16 instead and save the flag round trip; with 128-512 KiB per step to move, the
// stamp check would cost a pass over the data.)  All workgroups must be co-resident, one per CU (checked on the host);
// every spin is bounded and raises err[0].
//
// The K sum of the forward kernel is split in the two interleaved halves of k_lstm_step_fwd (the host passes that
// kernel's k-block lists) and everything downstream uses the same expressions, so outputs, saves and dropped copies are
// BIT-IDENTICAL to the launch-per-step path; saves use its lane-native layout, so either BPTT kernel can follow.
#include "common.h"
#include "ecog2txt_hip.h"
#include <stdlib.h>
#include <algorithm>

extern __shared__ __attribute__((aligned(16))) uint4 big_smem[];
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return fmaf(-2.0f, fast_rcp(__expf(2.0f * x) + 1.0f), 1.0f); }
__device__ __forceinline__ size_t native_tile(int s, int dir, int rt, int ut, int ndir, int RT, int UT) {
    return ((size_t)(s * ndir + dir) * RT + rt) * UT + ut;
}
__device__ __forceinline__ void philox_group(float rate, unsigned long long key, unsigned stream, unsigned long long group, float (&sc)[4]) {
    const unsigned thresh = (unsigned)(rate * 16777216.0f);
    const float keep = 1.0f / (1.0f - rate);
    unsigned r[4];
    philox4x32_10((unsigned)group, (unsigned)(group >> 32), stream, 0u, (unsigned)key, (unsigned)(key >> 32), r);
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[i] = (r[i] >> 8) >= thresh ? keep : 0.0f;
}
}  // namespace

// k-block that k_lstm_step_fwd gives K half h at position j of its accumulation order (step_geom / mma_chunk in lstm.hip:
// up to 16 k-blocks are one LDS chunk whose k-block pairs alternate between the halves; longer K ranges are cut in chunks
// of 8 k-blocks = 4 pairs).  The host checks this closed form against the list it derives from the geometry.
__host__ __device__ constexpr int big_kb_of(int KB, int h, int j) {
    return KB <= 16 ? 2 * (h + 2 * (j >> 1)) + (j & 1) : 8 * (j >> 2) + 2 * (h + 2 * ((j >> 1) & 1)) + (j & 1);
}

// Workgroup id -> (cluster, unit group).  The dispatcher places workgroup id on XCD id % 8 (observed; speed only).  A
// cluster's members are laid on `spread` XCDs (1, 2, 4 or 8): on one XCD the hand-off data is re-read from that XCD's own
// L2 by all members; on several, the cluster's bursts of saves and exchange reads use that many fabric links.
__device__ __forceinline__ void big_block_map(int b, int ncl, int UG, int spread, int& cl, int& ug) {
    const int G = 8 / spread;                         // XCD groups; clusters are dealt to the groups round-robin
    if (spread >= 8 || ncl % G != 0) { cl = b / UG; ug = b % UG; return; }
    const int xcd = b & 7, slot = b >> 3;
    const int g = xcd / spread, i = slot * spread + (xcd % spread);
    cl = g + G * (i / UG); ug = i % UG;
}

struct BigFwdArgs {
    const bf16_t* Gx;       // [S*B][ndir*H*4] bf16, (dir, unit, gate) interleaved, bias included
    const bf16_t* WhG;      // [ndir][4H/16 column tiles][KB][64][8]: MFMA fragment image of Bn[n = gate column][k] (e2t_pack_frag)
    bf16_t* Yext;           // [(S+3)*B][ldy]
    bf16_t* Ydrop;          // [S*B][ldy] or null
    float* Cs; bf16_t* Gs;  // lane-native saves (layout: lstm.hip; gates bf16)
    const int* lens;
    const float* c0;
    bf16_t* hx;             // [2 step parities][ndir][4*RB row tiles][KB][64 lanes][8]  h exchange, MFMA operand order
    unsigned* flags;        // [clusters][128]: steps published so far by (unit group, wave), over all launches
    int* err;
    int S, B, H, ndir, ldy, UT, KB;
    float forget_bias;
    DropCfg drop;
    unsigned char kbl[2][16];   // k-blocks of K half h in the accumulation order of k_lstm_step_fwd
    unsigned char nkb[2];
    long long* dbg;             // diagnostic phase stamps (E2T_LSTM_DBG), null in production
    int spread;
};

// KH = k-blocks per K half (KB = 2 * KH for the supported sizes: H % 64 == 0)
// FULL: both halves hold exactly KH k-blocks (H = 64 * KH): no run-time guards in the unrolled loops, so the MFMA phase is
// straight-line code and hipcc can run the LDS fragment reads ahead of the MFMAs (with the guards every k-block was a
// basic block of its own: read, wait, multiply -- 5.9 us instead of 1.9 per step at H = 1024)
template <int KH, bool FULL>
__global__ __launch_bounds__(256) void k_lstm_seq_fwd_big(BigFwdArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = p.B, H = p.H, S = p.S, KB = p.KB;
    const int RB = (B + 63) >> 6, RT = (B + 15) >> 4, RTP = RB * 4;
    const int UG = H >> 5;
    const int ncl = RB * p.ndir;
    int cl, ug;
    big_block_map(blockIdx.x, ncl, UG, p.spread, cl, ug);
    if (ug >= UG) return;
    const int rb = cl % RB, dir = cl / RB;
    const int frow = lane & 15, fq = lane >> 4;
    const int NH = p.ndir * H;
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);

    // LDS: state image [4 row tiles][KB][64 lanes] x 16 B (the B operand of every MFMA), then a per-wave 2-KiB transposer
    uint4* st_lds = big_smem;
    float* tr_lds = (float*)(big_smem + 4 * KB * 64) + wave * 512;

    // ---- once: W_h fragments of this wave's 8 units: column tiles ug*8 + wave*2 + a, K halves in list order ----
    bf16x8 W[2][2][KH];                                           // [a][half][j]
    int kbj[2][KH];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            kbj[h][j] = FULL ? big_kb_of(2 * KH, h, j) : ((j < p.nkb[h]) ? p.kbl[h][j] : 0);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (FULL || j < p.nkb[h]) v = ((const uint4*)p.WhG)[(((size_t)dir * (4 * H / 16) + ug * 8 + wave * 2 + a) * KB + kbj[h][j]) * 64 + lane];
                W[a][h][j] = *(bf16x8*)&v;
            }
        }
    // cells of this lane: (rt, a) -> utterance rb*64 + rt*16 + frow, unit ug*32 + wave*8 + a*4 + fq
    int len4[4];
    float cst[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int b = rb * 64 + rt * 16 + frow;
        len4[rt] = (b < B) ? p.lens[b] : 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            cst[rt][a] = 0.f;
            if (p.c0 && len4[rt] > 0) cst[rt][a] = p.c0[(size_t)b * NH + dir * H + ug * 32 + wave * 8 + a * 4 + fq];
        }
    }
    const int b2 = rb * 64 + fq * 16 + frow;                       // the utterance whose 8 units this lane holds after the transposition
    const int len2 = (b2 < B) ? p.lens[b2] : 0;
    // the flag word this wave publishes through, and where the launch starts counting
    unsigned* myflag = p.flags + (size_t)cl * 128 + ug * 4 + wave;
    const unsigned base = __builtin_amdgcn_readfirstlane(__hip_atomic_load(myflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned* clflags = p.flags + (size_t)cl * 128;
    const size_t hx_buf = (size_t)p.ndir * RTP * KB * 512;       // elements per step-parity buffer
    // row tile this wave PULLS (rt = wave) and the slot it PUBLISHES after the transposition (row tile fq, lane wave*16 + frow)
    const bf16_t* pull0 = p.hx + (((size_t)dir * RTP + rb * 4 + wave) * KB * 64 + lane) * 8;
    bf16_t* push0 = p.hx + ((((size_t)dir * RTP + rb * 4 + fq) * KB + ug) * 64 + wave * 16 + frow) * 8;

    long long pts[8];
#define PSTAMP(i) do { if (p.dbg && s == S / 2) pts[i] = wall_clock64(); } while (0)
    for (int s = 0; s < S; ++s) {
        PSTAMP(0);
        // ---- Gx of this step's cells: 8 x 8 B per lane (4 gates of bf16), requested first (used after the MFMAs) ----
        uint2 gxr[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int b = min(rb * 64 + rt * 16 + frow, B - 1);
            const bool act = s < len4[rt];
            const int tt = act ? (dir ? (len4[rt] - 1 - s) : s) : 0;
            const bf16_t* q = p.Gx + (((size_t)tt * B + b) * NH + dir * H + ug * 32 + wave * 8 + fq) * 4;
#pragma unroll
            for (int a = 0; a < 2; ++a) gxr[rt][a] = *(const uint2*)(q + a * 16);
        }
        // ---- state of row tile `wave` -> LDS ----
        if (s == 0) {
            // initial state from the row-major array: block 0 (forward), the all-zero slack block S+1 (backward direction);
            // rows beyond B read row B-1 (finite; their results are never stored)
            const int b = min(rb * 64 + wave * 16 + frow, B - 1);
            const int lw = (rb * 64 + wave * 16 + frow < B) ? p.lens[b] : 0;
            size_t tau = 0, srb = b;
            if (lw > 0 && dir == 1) { tau = (size_t)S + 1; srb = 0; }
            const bf16_t* src = p.Yext + (tau * B + srb) * p.ldy + dir * H + fq * 8;
            for (int kb = 0; kb < KB; ++kb) dma16_to_lds(src + kb * 32, lds_addr_of(st_lds + (wave * KB + kb) * 64));
        } else {
            const unsigned target = base + (unsigned)s;            // every producer has published step s-1
            int spins = 0;
            for (;;) {
                const unsigned f0 = __hip_atomic_load(clflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned f1 = __hip_atomic_load(clflags + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (unit groups beyond UG do not exist: their words are never written)
                const bool ok0 = (lane >> 2) >= UG || (int)(f0 - target) >= 0;
                const bool ok1 = ((64 + lane) >> 2) >= UG || (int)(f1 - target) >= 0;
                if (__all(ok0 && ok1)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;        // bounded: never hang the GPU; once any wave has given up nobody waits any more
                if ((spins & 255) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1 << 17)) { __hip_atomic_store(p.err, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            PSTAMP(1);
            const bf16_t* src = pull0 + ((s - 1) & 1) * hx_buf;
            for (int kb = 0; kb < KB; ++kb) dma16_to_lds_sc1(src + (size_t)kb * 512, lds_addr_of(st_lds + (wave * KB + kb) * 64));
        }
        dma_wait_all();
        __syncthreads();
        PSTAMP(2);

        // ---- recurrent product: acc[half][rt][a] += W[a][half][j] (16 (unit, gate) rows x 32 k) . state(32 k x 16 utterances) ----
        f32x4 acc[2][4][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[h][rt][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (FULL) {
            // Software pipeline, pinned: the 4 state fragments of k-block i+1 are requested BEFORE the 8 MFMAs of k-block i
            // (left alone, hipcc recycles one register quad: read, wait, two MFMAs -- the LDS latency 128 times per step).
            // The k-block of (half h, position j) is a compile-time function of the step kernel's LDS geometry, so every
            // LDS offset is an immediate.
            const uint4* sp = st_lds + lane;
            uint4 cur[4], nxt[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) cur[rt] = sp[(rt * (2 * KH) + big_kb_of(2 * KH, 0, 0)) * 64];
#pragma unroll
            for (int idx = 0; idx < 2 * KH; ++idx) {
                const int j = idx >> 1, h = idx & 1;
                if (idx + 1 < 2 * KH) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) nxt[rt] = sp[(rt * (2 * KH) + big_kb_of(2 * KH, (idx + 1) & 1, (idx + 1) >> 1)) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        acc[h][rt][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[a][h][j], *(const bf16x8*)&cur[rt], acc[h][rt][a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) cur[rt] = nxt[rt];
            }
        } else {
#pragma unroll
            for (int j = 0; j < KH; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (j < p.nkb[h]) {
                        const uint4* sp = st_lds + kbj[h][j] * 64 + lane;
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) {
                            const uint4 sv = sp[rt * KB * 64];
#pragma unroll
                            for (int a = 0; a < 2; ++a)
                                acc[h][rt][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[a][h][j], *(const bf16x8*)&sv, acc[h][rt][a], 0, 0, 0);
                        }
                    }
                }
        }
        __syncthreads();            // everybody is done reading the state image (the next step's DMA overwrites it)
        PSTAMP(3);

        // ---- lane-local cell update, 8 cells ----
        float gi[4][2], gj[4][2], gf[4][2], go[4][2], hv[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const bool active = s < len4[rt];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 z = (f32x4){acc[0][rt][a][0] + acc[1][rt][a][0], acc[0][rt][a][1] + acc[1][rt][a][1],
                                        acc[0][rt][a][2] + acc[1][rt][a][2], acc[0][rt][a][3] + acc[1][rt][a][3]};
                const uint2 gr = gxr[rt][a];
                gi[rt][a] = fsigmoid(z[0] + __uint_as_float(gr.x << 16));
                gj[rt][a] = ftanh(z[1] + __uint_as_float(gr.x & 0xFFFF0000u));
                gf[rt][a] = fsigmoid(z[2] + __uint_as_float(gr.y << 16) + p.forget_bias);
                go[rt][a] = fsigmoid(z[3] + __uint_as_float(gr.y & 0xFFFF0000u));
                const float cv = fmaf(gf[rt][a], cst[rt][a], gi[rt][a] * gj[rt][a]);
                hv[rt][a] = active ? go[rt][a] * ftanh(cv) : 0.f;
                if (active) cst[rt][a] = cv;
                tr_lds[(rt * 16 + frow) * 8 + a * 4 + fq] = hv[rt][a];       // -> [row tile][utterance][8 consecutive units]
            }
        }
        // transposed view: this lane now holds the 8 consecutive units wave*8 .. +7 of utterance rb*64 + fq*16 + frow
        const float4 t0 = *(const float4*)(tr_lds + (fq * 16 + frow) * 8), t1 = *(const float4*)(tr_lds + (fq * 16 + frow) * 8 + 4);
        const float h8[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        u32x4 hb;
#pragma unroll
        for (int i = 0; i < 4; ++i) hb[i] = (unsigned)f2bf(h8[2 * i]) | ((unsigned)f2bf(h8[2 * i + 1]) << 16);
        PSTAMP(4);
        if (s + 1 < S) {
            // publish: one write-through 16-B store per lane, acknowledged, then this wave's flag
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(push0 + (s & 1) * hx_buf), "v"(hb) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(myflag, base + (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PSTAMP(5);
        // ---- off the critical path: row-major copies for the next layer / BPTT, saves ----
        {
            if (b2 < B) {
                const bool act2 = s < len2;
                const int t2 = dir ? (len2 - 1 - s) : s;
                const size_t blk = act2 ? (size_t)(t2 + 1) : (size_t)(s + 1);
                const int u8 = ug * 32 + wave * 8;
                *(u32x4*)(p.Yext + (blk * B + b2) * p.ldy + dir * H + u8) = hb;           // padded positions: zeros
                if (p.Ydrop) {
                    u32x4 hd = (u32x4){0u, 0u, 0u, 0u};
                    const size_t m = (size_t)(act2 ? t2 : s) * B + b2;
                    if (act2) {
                        float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                        if (p.drop.rate > 0.f) {
                            const unsigned long long e0 = m * NH + dir * H + u8;          // multiple of 4 (H, u8 are)
                            float s4[4];
                            philox_group(p.drop.rate, key, p.drop.stream, e0 >> 2, s4);
                            sc[0] = s4[0]; sc[1] = s4[1]; sc[2] = s4[2]; sc[3] = s4[3];
                            philox_group(p.drop.rate, key, p.drop.stream, (e0 >> 2) + 1, s4);
                            sc[4] = s4[0]; sc[5] = s4[1]; sc[6] = s4[2]; sc[7] = s4[3];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) hd[i] = (unsigned)f2bf(h8[2 * i] * sc[2 * i]) | ((unsigned)f2bf(h8[2 * i + 1] * sc[2 * i + 1]) << 16);
                    }
                    __builtin_nontemporal_store(hd, (u32x4*)(p.Ydrop + m * p.ldy + dir * H + u8));
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int b = rb * 64 + rt * 16 + frow;
            if (b < B && s < len4[rt]) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    // lane-native layout of the step kernels: unit tile ug*2 + wave/2, lane group (wave&1)*2 + a, register fq
                    const size_t tile = native_tile(s, dir, rb * 4 + rt, ug * 2 + (wave >> 1), p.ndir, RT, p.UT);
                    const int ln = ((wave & 1) * 2 + a) * 16 + frow;
                    {   // (i, j, f, o) of unit register fq as 4 x bf16: half (fq & 1) of the 16-B slot of unit pair fq >> 1 (layout: lstm.hip)
                        const unsigned long long g = (unsigned long long)f2bf_pk(gi[rt][a], gj[rt][a]) | ((unsigned long long)f2bf_pk(gf[rt][a], go[rt][a]) << 32);
                        __builtin_nontemporal_store(g, (unsigned long long*)(p.Gs + (((tile * 2 + (fq >> 1)) * 64 + ln) * 2 + (fq & 1)) * 4));
                    }
                    __builtin_nontemporal_store(cst[rt][a], &p.Cs[((tile * 2 + (fq >> 1)) * 64 + ln) * 2 + (fq & 1)]);    // (next read by the BPTT)
                }
            }
        }
        PSTAMP(6);
        if (p.dbg && s == S / 2 && lane == 0)
            for (int i = 0; i < 7; ++i) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = pts[i];
    }
#undef PSTAMP
}

static int set_lds(const void* fn, size_t bytes) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { e2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return E2T_ERR_HIP; }
    return E2T_OK;
}

// k-block lists of the two K halves of k_lstm_step_fwd (its chunked LDS geometry: chunks of `kch` k-blocks, pairs of
// k-blocks alternate between the halves) -- see step_geom / mma_chunk in lstm.hip
static void step_fwd_halves(int KB, unsigned char (&kbl)[2][16], unsigned char (&nkb)[2]) {
    const int per_kb = 4 * 64 + 4 * 64, budget = (160 * 1024) / 16 - (4 * 256 + 1024);
    int kch, nch;
    const int kb_pad = (KB + 1) & ~1;
    if (kb_pad * per_kb <= budget) { kch = KB; nch = 1; }
    else { int k = (budget / 2) / per_kb; k &= ~1; if (k < 2) k = 2; kch = k; nch = (KB + k - 1) / k; }
    int n[2] = {0, 0};
    for (int c = 0; c < nch; ++c) {
        const int kb0 = c * kch, kc = std::min(kch, KB - kb0), npr = kc >> 1;
        for (int pp = 0; pp < npr; ++pp) { kbl[pp & 1][n[pp & 1]++] = (unsigned char)(kb0 + 2 * pp); kbl[pp & 1][n[pp & 1]++] = (unsigned char)(kb0 + 2 * pp + 1); }
        if (kc & 1) kbl[npr & 1][n[npr & 1]++] = (unsigned char)(kb0 + kc - 1);
    }
    nkb[0] = (unsigned char)n[0]; nkb[1] = (unsigned char)n[1];
}

extern "C" int e2t_lstm_big_ok(int H) { return (H % 64 == 0 && H >= 448 && H <= 1024) ? 1 : 0; }

extern "C" int e2t_lstm_seq_fwd_big(const e2t_lstm_desc* d, const void* Gx, const void* WhG, void* Yext, void* Ydrop, float* Cs,
                                    void* Gs, const int32_t* lens, const float* c0, void* hx, uint32_t* flags, int32_t* err,
                                    int num_cus, void* stream) {
    E2T_CHECK_ARG(d && Gx && WhG && Yext && Cs && Gs && lens && hx && flags && err);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(e2t_lstm_big_ok(d->H) && d->ldy % 8 == 0 && d->ldy >= d->ndir * d->H);
    BigFwdArgs p{};
    p.Gx = (const bf16_t*)Gx; p.WhG = (const bf16_t*)WhG; p.Yext = (bf16_t*)Yext; p.Ydrop = (bf16_t*)Ydrop; p.Cs = Cs; p.Gs = (bf16_t*)Gs;
    p.lens = lens; p.c0 = c0; p.hx = (bf16_t*)hx; p.flags = flags; p.err = err;
    p.S = d->S; p.B = d->B; p.H = d->H; p.ndir = d->ndir; p.ldy = d->ldy; p.UT = d->H / 16; p.KB = d->H / 32;
    p.forget_bias = d->forget_bias;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    p.spread = e2t_dbg_int("E2T_BIG_SPREAD_FWD", 1); if (p.spread != 2 && p.spread != 4 && p.spread != 8) p.spread = 1;
    step_fwd_halves(p.KB, p.kbl, p.nkb);
    p.dbg = (long long*)e2t_dbg_ptr("E2T_LSTM_DBG");
    const int KH = std::max(p.nkb[0], p.nkb[1]);
    const int nwg = ((d->B + 63) / 64) * d->ndir * (d->H / 32);
    if (KH > 16 || nwg > num_cus) {
        e2t_set_error("big persistent recurrence not applicable (H=%d, %d workgroups, %d CUs)", d->H, nwg, num_cus);
        return E2T_ERR_ARG;
    }
    const size_t lds = (size_t)4 * p.KB * 64 * 16 + 4 * 2048;
    static const int rc8 = set_lds((const void*)k_lstm_seq_fwd_big<8, true>, 160 * 1024);
    static const int rc12 = set_lds((const void*)k_lstm_seq_fwd_big<12, true>, 160 * 1024);
    static const int rc16 = set_lds((const void*)k_lstm_seq_fwd_big<16, true>, 160 * 1024);
    static const int rcg = set_lds((const void*)k_lstm_seq_fwd_big<16, false>, 160 * 1024);
    if (rc8 || rc12 || rc16 || rcg) return E2T_ERR_HIP;
    bool full = p.nkb[0] == p.nkb[1] && p.nkb[0] * 2 == p.KB;
    for (int h = 0; h < 2 && full; ++h)
        for (int j = 0; j < p.nkb[h]; ++j) full = full && p.kbl[h][j] == big_kb_of(p.KB, h, j);
    if (full && KH == 8) hipLaunchKernelGGL((k_lstm_seq_fwd_big<8, true>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, p);
    else if (full && KH == 12) hipLaunchKernelGGL((k_lstm_seq_fwd_big<12, true>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, p);
    else if (full && KH == 16) hipLaunchKernelGGL((k_lstm_seq_fwd_big<16, true>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_lstm_seq_fwd_big<16, false>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, p);
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// BPTT, large hidden sizes (see the file header).  Cell ownership is the step kernels' lane-native one: wave q finishes
// row tile q of the workgroup's 64 utterances for its two unit tiles; lane (frow, fq) owns utterance q*16 + frow and
// units ut*16 + fq*4 .. +3, so the saved gates / cells are coalesced float4 / float2 reads and the gate gradients of a
// unit tile are 32 contiguous bytes per lane, in the row-major dG as well as in the exchange image.
// ---------------------------------------------------------------------------------------------------------------------
struct BigBwdArgs {
    const bf16_t* WhB;      // [ndir][UT][KB4][64][8]  fragment image of W_h^T (rows = units, K = 4H gate columns)
    bf16_t* dG;             // [(S+1)*B][lddg] row-major, (dir, unit, gate) interleaved
    const float* dY;        // [S*B][lddy] or null
    const bf16_t* Gs; const float* Cs;
    const int* lens;
    const float* c0; const float* dh_final; const float* dc_final;
    bf16_t* dgx;            // [2 step parities][ndir][4*RB row tiles][KB4][64][8]  dG exchange, MFMA operand order
    unsigned* flags;        // [clusters][128]
    int* err;
    int S, B, H, ndir, lddg, lddy, UT, KB4;
    DropCfg drop;
    long long* dbg;
    int spread;
};
#define BIG_CH 4                    // k-blocks per streamed chunk (x 4 row tiles = 16 KiB), two chunks per wave in LDS

template <int KQ>                   // k-blocks per K quarter = H / 32
__global__ __launch_bounds__(256) void k_lstm_seq_bwd_big(BigBwdArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = p.B, H = p.H, S = p.S, KB4 = p.KB4;
    const int RB = (B + 63) >> 6, RT = (B + 15) >> 4, RTP = RB * 4;
    const int UG = H >> 5;
    const int ncl = RB * p.ndir;
    int cl, ug;
    big_block_map(blockIdx.x, ncl, UG, p.spread, cl, ug);
    if (ug >= UG) return;
    const int rb = cl % RB, dir = cl / RB;
    const int frow = lane & 15, fq = lane >> 4;
    const int NH = p.ndir * H, K4 = 4 * H;
    const unsigned long long key = p.drop.seed + ((p.drop.rate > 0.f && p.drop.step) ? (unsigned long long)(*p.drop.step) : 0ull);
    constexpr int NCH = KQ / BIG_CH;

    // ---- once: W_h^T fragments: K quarter `wave` for the two unit tiles of this group ----
    bf16x8 W[2][KQ];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const uint4 v = ((const uint4*)p.WhB)[(((size_t)dir * p.UT + ug * 2 + u) * KB4 + wave * KQ + i) * 64 + lane];
            W[u][i] = *(const bf16x8*)&v;
        }
    uint4* ring = big_smem + (size_t)wave * (2 * BIG_CH * 4 * 64);           // [2 buffers][4 row tiles][BIG_CH][64 lanes]
    const unsigned ring_lds = lds_addr_of(ring);
    float4* part = (float4*)big_smem;                                         // aliases the rings between the two barriers

    // cells of this lane: row tile `wave`, unit tiles ug*2 + u
    const int rt = rb * 4 + wave;
    const int b = rb * 64 + wave * 16 + frow;
    const bool own = b < B;
    const int bc = min(b, B - 1);
    const int len = own ? p.lens[b] : 0;
    float ct[2][4], dcc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ct[u][r] = 0.f; dcc[u][r] = 0.f; }
    if (own) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t tile = native_tile(S - 1, dir, rt, ug * 2 + u, p.ndir, RT, p.UT);
            const float2 a = ((const float2*)p.Cs)[(tile * 2 + 0) * 64 + lane], c = ((const float2*)p.Cs)[(tile * 2 + 1) * 64 + lane];
            ct[u][0] = a.x; ct[u][1] = a.y; ct[u][2] = c.x; ct[u][3] = c.y;
        }
    }
    unsigned* myflag = p.flags + (size_t)cl * 128 + ug * 4 + wave;
    const unsigned base = __builtin_amdgcn_readfirstlane(__hip_atomic_load(myflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // this wave multiplies gate columns [wave*KQ*32, (wave+1)*KQ*32) = the columns of unit groups wave*UG/4 .. : their 4 waves each
    const unsigned* qflags = p.flags + (size_t)cl * 128 + wave * (UG / 4) * 4;
    const int nq = (UG / 4) * 4;                                              // words to poll (<= 32)
    const size_t dgx_buf = (size_t)p.ndir * RTP * KB4 * 512;
    const bf16_t* pull0 = p.dgx + ((((size_t)dir * RTP + rb * 4) * KB4 + wave * KQ) * 64 + lane) * 8;

    long long pts[8];
#define PSTAMP(i) do { if (p.dbg && s == S / 2) pts[i] = wall_clock64(); } while (0)
    for (int k = 0; k < S; ++k) {
        const int s = S - 1 - k;
        PSTAMP(0);
        const bool active = s < len;
        const int t = active ? (dir ? (len - 1 - s) : s) : 0;
        const size_t m = (size_t)t * B + bc;
        // ---- operands of this step's cells, requested first (they arrive while the state is streamed) ----
        float4 g4[2][4];
        float cp[2][4], dy[2][4], dhf[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ut = ug * 2 + u;
            const size_t tile = native_tile(s, dir, rt, ut, p.ndir, RT, p.UT);
            const size_t su = (size_t)bc * NH + dir * H + ut * 16 + fq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { g4[u][r] = make_float4(0.f, 0.f, 0.f, 0.f); cp[u][r] = 0.f; dy[u][r] = 0.f; dhf[u][r] = 0.f; }
            if (own && active) {
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const uint4 raw = ((const uint4*)p.Gs)[(tile * 2 + rp) * 64 + lane];
                    g4[u][2 * rp] = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xFFFF0000u), __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xFFFF0000u));
                    g4[u][2 * rp + 1] = make_float4(__uint_as_float(raw.z << 16), __uint_as_float(raw.z & 0xFFFF0000u), __uint_as_float(raw.w << 16), __uint_as_float(raw.w & 0xFFFF0000u));
                }
                if (p.dY) { const float4 v = *(const float4*)(p.dY + m * p.lddy + dir * H + ut * 16 + fq * 4); dy[u][0] = v.x; dy[u][1] = v.y; dy[u][2] = v.z; dy[u][3] = v.w; }
                if (s == len - 1) {
                    if (p.dh_final) { const float4 v = *(const float4*)(p.dh_final + su); dhf[u][0] = v.x; dhf[u][1] = v.y; dhf[u][2] = v.z; dhf[u][3] = v.w; }
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.dc_final) v = *(const float4*)(p.dc_final + su);
                    dcc[u][0] = v.x; dcc[u][1] = v.y; dcc[u][2] = v.z; dcc[u][3] = v.w;
                }
            }
            if (own) {
                if (s > 0) {
                    const size_t tp = native_tile(s - 1, dir, rt, ut, p.ndir, RT, p.UT);
                    const float2 a = ((const float2*)p.Cs)[(tp * 2 + 0) * 64 + lane], c = ((const float2*)p.Cs)[(tp * 2 + 1) * 64 + lane];
                    cp[u][0] = a.x; cp[u][1] = a.y; cp[u][2] = c.x; cp[u][3] = c.y;
                } else if (p.c0) {
                    const float4 c = *(const float4*)(p.c0 + su);
                    cp[u][0] = c.x; cp[u][1] = c.y; cp[u][2] = c.z; cp[u][3] = c.w;
                }
            }
        }
        f32x4 rec[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (k > 0) {
            // ---- wait until the producers of this wave's K quarter have published step s+1 ----
            const unsigned target = base + (unsigned)k;
            int spins = 0;
            for (;;) {
                const unsigned f0 = (lane < nq) ? __hip_atomic_load(qflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                if (__all((int)(f0 - target) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 255) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1 << 17)) { __hip_atomic_store(p.err, 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            PSTAMP(1);
            // ---- stream this quarter of the 64 dG rows through the wave's LDS ring, 16 KiB per chunk, one chunk ahead ----
            const bf16_t* src = pull0 + ((s + 1) & 1) * dgx_buf;
            auto issue = [&](int c) {
                const unsigned dst = ring_lds + (unsigned)(c & 1) * (BIG_CH * 4 * 1024);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int i = 0; i < BIG_CH; ++i)
                        dma16_to_lds_sc1(src + ((size_t)r4 * KB4 + c * BIG_CH + i) * 512, dst + (unsigned)(r4 * BIG_CH + i) * 1024);
            };
            f32x4 acc[4][2];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) { acc[r4][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[r4][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            issue(0);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + 1 < NCH) {
                    issue(c + 1);
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * BIG_CH) : "memory");     // chunk c has landed, chunk c+1 is in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const uint4* buf = ring + (c & 1) * (BIG_CH * 4 * 64) + lane;
                // pinned software pipeline inside the chunk: the 4 row-tile fragments of k-block i+1 are requested before the
                // 8 MFMAs of k-block i (left alone, hipcc recycles one register quad: read, wait, two MFMAs)
                uint4 cur[4], nxt[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) cur[r4] = buf[(r4 * BIG_CH) * 64];
#pragma unroll
                for (int i = 0; i < BIG_CH; ++i) {
                    if (i + 1 < BIG_CH) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) nxt[r4] = buf[(r4 * BIG_CH + i + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            acc[r4][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[u][c * BIG_CH + i], *(const bf16x8*)&cur[r4], acc[r4][u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) cur[r4] = nxt[r4];
                }
                // (the fragments are in registers: the buffer may be refilled by the issue two iterations on)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            PSTAMP(2);
            // ---- reduce the 4 K-quarter partials of every (row tile, unit tile) through LDS ----
            __syncthreads();                                   // every wave is done with its ring (part aliases it)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    part[((r4 * 2 + u) * 4 + wave) * 64 + lane] = make_float4(acc[r4][u][0], acc[r4][u][1], acc[r4][u][2], acc[r4][u][3]);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4 p0 = part[((wave * 2 + u) * 4 + 0) * 64 + lane], p1 = part[((wave * 2 + u) * 4 + 1) * 64 + lane];
                const float4 p2 = part[((wave * 2 + u) * 4 + 2) * 64 + lane], p3 = part[((wave * 2 + u) * 4 + 3) * 64 + lane];
                rec[u] = (f32x4){(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)};
            }
            __syncthreads();                                   // partials consumed: the rings may be refilled
        }
        PSTAMP(3);
        // ---- cell backward, 8 cells ----
        uint4 og[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            og[u][0] = og[u][1] = make_uint4(0u, 0u, 0u, 0u);
            if (own && active) {
                float dsc4[4] = {1.f, 1.f, 1.f, 1.f};
                if (p.dY && p.drop.rate > 0.f) philox_group(p.drop.rate, key, p.drop.stream, (m * NH + dir * H + (ug * 2 + u) * 16 + fq * 4) >> 2, dsc4);
                bf16_t o[16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 g = g4[u][r];
                    const float tc = ftanh(ct[u][r]);
                    const float dh = rec[u][r] + (dhf[u][r] + dy[u][r] * dsc4[r]);
                    const float dct = fmaf(dh, g.w * (1.f - tc * tc), dcc[u][r]);
                    o[r * 4 + 0] = f2bf(dct * (g.y * g.x * (1.f - g.x)));
                    o[r * 4 + 1] = f2bf(dct * (g.x * (1.f - g.y * g.y)));
                    o[r * 4 + 2] = f2bf(dct * (cp[u][r] * g.z * (1.f - g.z)));
                    o[r * 4 + 3] = f2bf(dh * (tc * g.w * (1.f - g.w)));
                    dcc[u][r] = dct * g.z;
                }
                og[u][0] = make_uint4(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16), o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16));
                og[u][1] = make_uint4(o[8] | ((unsigned)o[9] << 16), o[10] | ((unsigned)o[11] << 16), o[12] | ((unsigned)o[13] << 16), o[14] | ((unsigned)o[15] << 16));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) ct[u][r] = cp[u][r];          // c_t of step s-1 is c_{t-1} of step s, active or not
        }
        PSTAMP(4);
        if (s > 0) {
            // publish (rows beyond B / padded positions: zeros): gate columns ut*64 + fq*16 .. +15 = k-block ut*2 + fq/2,
            // lanes (fq&1)*32 + frow and +16; write-through, acknowledged, then the flag
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ut = ug * 2 + u;
                bf16_t* hp = p.dgx + (s & 1) * dgx_buf + ((((size_t)dir * RTP + rt) * KB4 + ut * 2 + (fq >> 1)) * 64 + (fq & 1) * 32 + frow) * 8;
                const u32x4 v0 = (u32x4){og[u][0].x, og[u][0].y, og[u][0].z, og[u][0].w}, v1 = (u32x4){og[u][1].x, og[u][1].y, og[u][1].z, og[u][1].w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:256 sc1" :: "v"(hp), "v"(v0), "v"(v1) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(myflag, base + (unsigned)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PSTAMP(5);
        // ---- off the critical path: row-major dG for the weight / input gradient GEMMs ----
        if (own) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4* gp = (uint4*)(p.dG + ((size_t)(active ? t : s) * B + b) * p.lddg + (size_t)dir * K4 + ((ug * 2 + u) * 16 + fq * 4) * 4);
                gp[0] = og[u][0]; gp[1] = og[u][1];
            }
        }
        PSTAMP(6);
        if (p.dbg && s == S / 2 && lane == 0)
            for (int i = 0; i < 7; ++i) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = pts[i];
    }
#undef PSTAMP
}

extern "C" int e2t_lstm_seq_bwd_big(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY, int lddy,
                                    const void* Gs, const float* Cs, const int32_t* lens, const float* c0, const float* dh_final,
                                    const float* dc_final, void* dgx, uint32_t* flags, int32_t* err, int num_cus, void* stream) {
    E2T_CHECK_ARG(d && WhB && dG && Gs && Cs && lens && dgx && flags && err);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(e2t_lstm_big_ok(d->H) && d->H % 128 == 0 && lddg % 8 == 0 && lddg >= d->ndir * 4 * d->H);
    BigBwdArgs p{};
    p.WhB = (const bf16_t*)WhB; p.dG = (bf16_t*)dG; p.dY = dY; p.Gs = (const bf16_t*)Gs; p.Cs = Cs; p.lens = lens; p.c0 = c0;
    p.dh_final = dh_final; p.dc_final = dc_final; p.dgx = (bf16_t*)dgx; p.flags = flags; p.err = err;
    p.S = d->S; p.B = d->B; p.H = d->H; p.ndir = d->ndir; p.lddg = lddg; p.lddy = lddy; p.UT = d->H / 16; p.KB4 = d->H / 8;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    p.dbg = (long long*)e2t_dbg_ptr("E2T_LSTM_DBG");
    p.spread = e2t_dbg_int("E2T_BIG_SPREAD_BWD", 1); if (p.spread != 2 && p.spread != 4 && p.spread != 8) p.spread = 1;
    const int KQ = d->H / 32;
    const int nwg = ((d->B + 63) / 64) * d->ndir * (d->H / 32);
    if (nwg > num_cus) {
        e2t_set_error("big persistent BPTT not applicable (H=%d, %d workgroups, %d CUs)", d->H, nwg, num_cus);
        return E2T_ERR_ARG;
    }
    const size_t lds = (size_t)4 * 2 * BIG_CH * 4 * 64 * 16;      // 4 waves x 2 buffers x 16 KiB
    static const int rc16 = set_lds((const void*)k_lstm_seq_bwd_big<16>, 160 * 1024);
    static const int rc24 = set_lds((const void*)k_lstm_seq_bwd_big<24>, 160 * 1024);
    static const int rc32 = set_lds((const void*)k_lstm_seq_bwd_big<32>, 160 * 1024);
    if (rc16 || rc24 || rc32) return E2T_ERR_HIP;
    switch (KQ) {
        case 16: hipLaunchKernelGGL(k_lstm_seq_bwd_big<16>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, p); break;
        case 24: hipLaunchKernelGGL(k_lstm_seq_bwd_big<24>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, p); break;
        case 32: hipLaunchKernelGGL(k_lstm_seq_bwd_big<32>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, p); break;
        default: e2t_set_error("big persistent BPTT: H=%d not instantiated (512, 768, 1024)", d->H); return E2T_ERR_ARG;
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}
