// HBM-bound / small kernels around the matmuls of the ECoG->text hot path:
// length extraction, time reversal + decimation gathers, conv im2row packing,
// transposes, weight packing, embedding, softmax cross-entropy, squared-error
// head, Adam+EMA.  Reference rows: a4, a5, a6, a8, a9, a10.
#include "common.h"
#include "ecog2txt_hip.h"

// ---------------------------------------------------------------------------
// a4: valid length of a zero-padded batch = number of non-zero rows
// (nn.sequences_tools usage at ecog2txt/trainers.py:789-790, 806-807).
// One workgroup per utterance, one wave per time row, 16-B coalesced loads.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_seq_lengths_f32(const float* x, int T, int C, int div, int* lens, int* lens_div) {
    // one 16-wave workgroup per utterance, one wave per time row: 4096 waves in flight for B = 256
    // (no atomics, no pre-zeroing: the result is written once, so the launch is replay-safe)
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (size_t)b * T * C;
    int cnt = 0;
    if ((C & 3) == 0 && C <= 256) {
        // the common shape (one 16-B load per lane covers a row): 8 rows in flight per wave -- a load / vote / next-row
        // chain is a memory round trip per row (33 us for 105 MB), the loads of 8 rows are independent
        const int c = lane * 4;
        for (int t0 = wave; t0 < T; t0 += 16 * 8) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = t0 + 16 * i;
                v[i] = (t < T && c < C) ? *(const float4*)(xb + (size_t)t * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (__any((v[i].x != 0.f) | (v[i].y != 0.f) | (v[i].z != 0.f) | (v[i].w != 0.f))) cnt++;      // wave-uniform
        }
    } else if (C <= 32) {
        // narrow rows (the 13 MFCCs of the auxiliary targets): a wave per row is one dependent ~1-us load per row
        // (54 us for 5 MB, slowing the GEMM it ran next to); here a LANE owns a row and its C loads are independent
        for (int t = threadIdx.x; t < T; t += 1024) {
            const float* row = xb + (size_t)t * C;
            bool nz = false;
#pragma unroll 8
            for (int c = 0; c < C; ++c) nz |= row[c] != 0.f;
            cnt += __popcll(__ballot(nz)) ;            // wave-uniform count of this pass's non-zero rows
        }
    } else {
        for (int t = wave; t < T; t += 16) {
            const float* row = xb + (size_t)t * C;
            bool nz = false;
            if ((C & 3) == 0) {
                for (int c = lane * 4; c < C; c += 256) {
                    float4 v = *(const float4*)(row + c);
                    nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
                }
            } else {
                for (int c = lane; c < C; c += 64) nz |= row[c] != 0.f;
            }
            if (__any(nz)) cnt++;            // wave-uniform
        }
    }
    __shared__ int sc[16];
    if (lane == 0) sc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int i = 0; i < 16; ++i) n += sc[i];
        lens[b] = n;
        if (lens_div) lens_div[b] = (n + div - 1) / div;
    }
}

// Same result for END-padded data (the only padding the reference produces: subjects.py:386-390) without reading the
// whole batch: the last non-zero row is searched from the tail, 32 rows per pass (16 waves x 2 rows, 16-B loads), so an
// utterance costs (padding + <= 32 rows) instead of T rows -- 105 MB -> ~9 MB at cfg2, 2.1 GB -> ~35 MB at cfg5.
// length = index of the last non-zero row + 1; interior all-zero rows (not padding) would make this differ from the
// non-zero-row COUNT of k_seq_lengths_f32: the host rejects such utterances when it stages them.
__global__ __launch_bounds__(1024) void k_seq_lengths_tail_f32(const float* x, int T, int C, int div, int* lens, int* lens_div) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (size_t)b * T * C;
    __shared__ int best;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    for (int hi = T; hi > 0; hi -= 32) {
        int found = 0;                                   // (1 + row index) of a non-zero row of this pass, wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = hi - 1 - (wave * 2 + i);
            bool nz = false;
            if (t >= 0)
                for (int c = lane * 4; c < C; c += 256) {
                    const float4 v = *(const float4*)(xb + (size_t)t * C + c);
                    nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
                }
            if (__any(nz)) found = max(found, t + 1);
        }
        if (lane == 0 && found) atomicMax(&best, found);
        __syncthreads();
        const int n = best;
        if (n > 0) break;                                // uniform: every thread reads the same word after the barrier
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int n = best;
        lens[b] = n;
        if (lens_div) lens_div[b] = (n + div - 1) / div;
    }
}

__global__ void k_seq_lengths_i32(const int* x, int B, int L, int pad, int div, int* lens, int* lens_div) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int l = 0; l < L; ++l) n += x[(size_t)b * L + l] != pad;
    lens[b] = n;
    if (lens_div) lens_div[b] = (n + div - 1) / div;
}

// out[0] = sum_b x[b]   (single block, deterministic)
__global__ __launch_bounds__(256) void k_sum_i32(const int* x, int n, int* out) {
    __shared__ int sh[256];
    // 4 loads in flight per thread (a single dependent chain over 8704 elements took 20 us); the association is fixed
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int i = threadIdx.x;
    for (; i + 768 < n; i += 1024) { a0 += x[i]; a1 += x[i + 256]; a2 += x[i + 512]; a3 += x[i + 768]; }
    for (; i < n; i += 256) a0 += x[i];
    int acc = (a0 + a1) + (a2 + a3);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// out[slot] = scale * sum_i x[i] / max(*count,1)  (single block, fixed order => deterministic)
// (blockIdx.x = 1: the second of two sums made in one launch, e2t_sum2_f32 -- same association, same bits)
__global__ __launch_bounds__(256) void k_sum_f32(const float* x, int n, const int* count, float scale, float* out,
                                                  const float* x1 = nullptr, float scale1 = 0.f, float* out1 = nullptr) {
    __shared__ float sh[256];
    if (blockIdx.x == 1) { x = x1; scale = scale1; out = out1; }
    // 4 loads in flight per thread (a single dependent chain over 8704 elements took 20 us); the association is fixed
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = threadIdx.x;
    for (; i + 768 < n; i += 1024) { a0 += x[i]; a1 += x[i + 256]; a2 += x[i + 512]; a3 += x[i + 768]; }
    for (; i < n; i += 256) a0 += x[i];
    float acc = (a0 + a1) + (a2 + a3);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float d = count ? (float)max(*count, 1) : 1.f;
        out[0] = scale * sh[0] / d;
    }
}

// ---------------------------------------------------------------------------
// a5 + a6 staging: A[(t',b)][(w,c)] = bf16( Xrev[b][t'*N + w][c] ), where Xrev
// is tf.reverse_sequence over the valid length (trainers.py:808-810) and the
// ragged tail is zero-padded.  Every source row is one contiguous C*4-byte
// read; every destination row one contiguous N*C*2-byte write.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_pack(const float* x, const int* lens, int B, int T, int C, int N,
                                                    int S, bf16_t* A, int lda, int G) {
    // row m <-> (t', b): time-major (G == 1), or grouped m = (tg*B + b)*G + g with t' = tg*G + g -- the order in which the G
    // consecutive steps that a later conv layer folds into one are adjacent rows (its im2row is then a plain view)
    const int m = blockIdx.x;
    const int tgb = m / G, g = m - tgb * G;
    const int tp = (tgb / B) * G + g, b = tgb % B;
    const int len = lens[b];
    bf16_t* arow = A + (size_t)m * lda;
    const int K = N * C;
    if ((C & 3) == 0) {
        for (int k4 = threadIdx.x * 4; k4 < lda; k4 += 1024) {
            ushort4 o = make_ushort4(0, 0, 0, 0);
            if (k4 == K) o.x = 0x3F80;          // column K (when lda > K) = 1.0: the ones column of the TN weight-gradient GEMM
            if (k4 < K) {
                const int w = k4 / C, c = k4 - w * C;
                const int tt = tp * N + w;
                if (tt < len) {
                    float4 v = *(const float4*)(x + ((size_t)b * T + (len - 1 - tt)) * C + c);
                    o.x = f2bf(v.x); o.y = f2bf(v.y); o.z = f2bf(v.z); o.w = f2bf(v.w);
                }
            }
            *(ushort4*)(arow + k4) = o;
        }
    } else {
        for (int k = threadIdx.x; k < lda; k += 256) {
            bf16_t o = (k == K) ? (bf16_t)0x3F80 : (bf16_t)0;
            if (k < K) {
                const int w = k / C, c = k - w * C;
                const int tt = tp * N + w;
                if (tt < len) o = f2bf(x[((size_t)b * T + (len - 1 - tt)) * C + c]);
            }
            arow[k] = o;
        }
    }
}

// scatter of d(loss)/dA back to the batch-major, un-reversed input (a12 saliency)
__global__ __launch_bounds__(256) void k_conv_unpack_grad(const float* dA, int ldda, const int* lens, int B, int T, int C,
                                                           int N, int S, float* dx, int G) {
    const int bt = blockIdx.x;              // (b, t) batch-major destination row
    const int b = bt / T, t = bt % T;
    const int len = lens[b];
    float* drow = dx + (size_t)bt * C;
    if (t >= len) { for (int c = threadIdx.x; c < C; c += 256) drow[c] = 0.f; return; }
    const int tt = len - 1 - t;             // reversed time
    const int tp = tt / N, w = tt - tp * N;
    const float* src = dA + ((size_t)((tp / G) * B + b) * G + tp % G) * ldda + (size_t)w * C;      // (grouped row order, see k_conv_pack)
    for (int c = threadIdx.x; c < C; c += 256) drow[c] = src[c];
}

// time-major, reversed, every-N-th-sample gather of encoder targets
// (trainers.py:791-795: reverse_sequence then [:, 0::N, :])
__global__ void k_gather_rev_decim_f32(const float* a, const int* tlens, int B, int T, int Kf, int N, int S, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)S * B * Kf) return;
    const int k = i % Kf; const size_t m = i / Kf;
    const int b = m % B, tp = m / B;
    const int len = tlens[b], tt = tp * N;
    out[i] = tt < len ? a[((size_t)b * T + (len - 1 - tt)) * Kf + k] : 0.f;
}
__global__ void k_gather_rev_decim_i32(const int* a, const int* tlens, int B, int T, int N, int S, int* out) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (size_t)S * B) return;
    const int b = m % B, tp = m / B;
    const int len = tlens[b], tt = tp * N;
    out[m] = tt < len ? a[(size_t)b * T + (len - 1 - tt)] : 0;
}

// decoder inputs/targets, time-major: U[l][b] = l==0 ? <EOS> : Y[b][l-1];  Tg[l][b] = Y[b][l]
__global__ void k_decoder_tokens(const int* y, int B, int L, int eos, int* U, int* Tg) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= L * B) return;
    const int b = m % B, l = m / B;
    U[m] = l == 0 ? eos : y[(size_t)b * L + l - 1];
    Tg[m] = y[(size_t)b * L + l];
}

// ---------------------------------------------------------------------------
// bf16 transpose through LDS: out[c][r] = in[r][c], r < R, c < Ccols; columns
// R..ld_out-1 of every written output row are zero-filled (K padding for the GEMM).
// ---------------------------------------------------------------------------
// 64x64 tile per workgroup: 16-B global loads along the input rows, 2-B scatter into a padded LDS tile,
// 16-B LDS reads along the transposed direction, 16-B global stores (needs ld_out % 8 == 0; falls back to
// element stores otherwise).
__global__ __launch_bounds__(256) void k_transpose_bf16(const bf16_t* in, int ld_in, int R, int Ccols, bf16_t* out, int ld_out) {
    __shared__ bf16_t tile[64][72];                    // [c][r], row pitch 144 B (16-B aligned, conflict-light)
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const bool in_vec = (ld_in & 7) == 0 && (((uintptr_t)in) & 15) == 0;
    // load: thread -> (row = tid/8 + 32*i, 8 columns starting at (tid%8)*8)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = (tid >> 3) + 32 * i, cl = (tid & 7) * 8;
        const int r = r0 + rl, c = c0 + cl;
        bf16_t v[8];
        if (r < R && in_vec && c + 7 < Ccols) {
            const uint4 q = *(const uint4*)(in + (size_t)r * ld_in + c);
            v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
            v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r < R && c + j < Ccols) ? in[(size_t)r * ld_in + c + j] : (bf16_t)0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[cl + j][rl] = v[j];
    }
    __syncthreads();
    // store: thread -> (out row c = tid/8 + 32*i, 8 consecutive r starting at (tid%8)*8)
    const bool out_vec = (ld_out & 7) == 0 && (((uintptr_t)out) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cl = (tid >> 3) + 32 * i, rl = (tid & 7) * 8;
        const int c = c0 + cl, r = r0 + rl;
        if (c >= Ccols || r >= ld_out) continue;
        if (out_vec && r + 7 < ld_out) {
            *(uint4*)(out + (size_t)c * ld_out + r) = *(const uint4*)&tile[cl][rl];
        } else {
            for (int j = 0; j < 8 && r + j < ld_out; ++j) out[(size_t)c * ld_out + r + j] = tile[cl][rl + j];
        }
    }
}

// ---------------------------------------------------------------------------
// weight packing (fp32 master -> bf16 operand images), run after every Adam step
// ---------------------------------------------------------------------------
// dst[r][c] = bf16(src[r*sr + c*sc]) for r<R, c<C
__global__ void k_cast_pack(const float* src, long sr, long sc, int R, int C, bf16_t* dst, int ld_dst) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c < C && r < R) dst[(size_t)r * ld_dst + c] = f2bf(src[(size_t)r * sr + (size_t)c * sc]);
}
// MFMA B-fragment image of the logical matrix Bn[n][k] = src[n*sn + k*sk]  (n < Nn, k < Kk):
// dst[(nt*KB + kb)*64 + lane][j] = Bn[nt*16 + (lane&15)][kb*32 + (lane>>4)*8 + j]   (0 outside)
__global__ __launch_bounds__(64) void k_pack_frag(const float* src, long sn, long sk, int Nn, int Kk, int KB, bf16_t* dst) {
    const int nt = blockIdx.y, kb = blockIdx.x, lane = threadIdx.x;
    const int n = nt * 16 + (lane & 15);
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb * 32 + (lane >> 4) * 8 + j;
        o[j] = (n < Nn && k < Kk) ? f2bf(src[(size_t)n * sn + (size_t)k * sk]) : (bf16_t)0;
    }
    uint4 v;
    v.x = o[0] | ((unsigned)o[1] << 16); v.y = o[2] | ((unsigned)o[3] << 16);
    v.z = o[4] | ((unsigned)o[5] << 16); v.w = o[6] | ((unsigned)o[7] << 16);
    ((uint4*)dst)[((size_t)nt * KB + kb) * 64 + lane] = v;
}

// All operand images in ONE launch: a device table of descriptors, each owning a contiguous range of
// 256-thread workgroups (prefix in `first_block`).  kind 0 = strided cast (one row x 256 columns per
// workgroup; for sources that are contiguous along the destination's columns), kind 1 = MFMA fragment image
// (4 fragments per workgroup), kind 2 = transposing cast (source contiguous along the destination's ROWS, s0 == 1):
// 64x64 tiles through LDS so that both the fp32 reads and the bf16 writes are coalesced.
// E2T_PACK_UNITS work units per workgroup: the descriptor search and the launch overhead of a workgroup are paid once
// per 8 units (42 k single-unit workgroups spent most of their life searching)
__device__ __forceinline__ int pack_units(const e2t_pack_desc& d) {
    const int NT = (d.d0 + 15) / 16;
    switch (d.kind) {
        case 0: return d.d0 * ((d.d1 + 255) / 256);
        case 3: return d.d0 * ((d.d1 + 1023) / 1024);
        case 2: case 4: return ((d.d0 + 63) / 64) * ((d.d1 + 63) / 64);
        case 5: return ((NT + 3) / 4) * ((d.ld + 1) / 2);
        case 6: return NT * ((d.ld + 1) / 2);
        default: return (NT * d.ld + 3) / 4;
    }
}
__device__ __forceinline__ void pack_unit(const e2t_pack_desc& d, const int lb, const float* src, bf16_t* dst, float* lds);
__global__ __launch_bounds__(256) void k_pack_batch(const e2t_pack_desc* descs, int ndesc, const float* base) {
    // ONE staging buffer for every kind that goes through LDS (a 64 x 64 transpose tile, or two [32 k][64 n] blocks): the kinds
    // used to declare their own arrays -- 50 KB of LDS together, three workgroups per CU for a kernel that lives on loads in flight
    __shared__ __attribute__((aligned(16))) float pk_lds[64 * 68];
    // binary search for the descriptor that owns this workgroup (uniform)
    int lo = 0, hi = ndesc - 1;
    const int bid = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].first_block <= bid) lo = mid; else hi = mid - 1; }
    const e2t_pack_desc d = descs[lo];
    const int units = pack_units(d);
    const float* src = base + d.src_off;
    bf16_t* dst = (bf16_t*)d.dst;
    if (d.kind == 3) {
        // E2T_PACK_UNITS units per workgroup with all their loads in flight at once (a workgroup per unit is bound by its
        // own start-up + one memory round trip)
        const int u0 = (bid - d.first_block) * E2T_PACK_UNITS;
        const int cb = (d.d1 + 1023) / 1024;
        float4 v[E2T_PACK_UNITS]; size_t o[E2T_PACK_UNITS]; bool ok[E2T_PACK_UNITS];
#pragma unroll
        for (int j = 0; j < E2T_PACK_UNITS; ++j) {
            const int u = u0 + j, r = u / cb, c = (u - r * cb) * 1024 + threadIdx.x * 4;
            ok[j] = u < units && c < d.d1;
            o[j] = (size_t)r * d.ld + c;
            if (ok[j]) v[j] = *(const float4*)(src + (size_t)r * d.s0 + c);
        }
#pragma unroll
        for (int j = 0; j < E2T_PACK_UNITS; ++j)
            if (ok[j]) *(ushort4*)(dst + o[j]) = make_ushort4(f2bf(v[j].x), f2bf(v[j].y), f2bf(v[j].z), f2bf(v[j].w));
        return;
    }
    if (d.kind == 1 && d.s1 == 1 && ((d.s0 | d.src_off | d.d1) & 3) == 0) {
        // fragment images of a source contiguous along k: two 16-B loads per lane and fragment, E2T_PACK_UNITS x 4 fragments
        // per workgroup in flight
        const int u0 = (bid - d.first_block) * E2T_PACK_UNITS;
        const int KB = d.ld, lane = threadIdx.x & 63, NT = (d.d0 + 15) / 16;
        float4 a[E2T_PACK_UNITS], b2[E2T_PACK_UNITS]; bool ok[E2T_PACK_UNITS];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < E2T_PACK_UNITS; ++j) {
            const int f = (u0 + j) * 4 + (threadIdx.x >> 6);
            ok[j] = f < NT * KB;
            const int nt = f / KB, kb = f - nt * KB;
            const int n = nt * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 8;
            // (out-of-range pieces load the source's first 16 B and are zeroed afterwards: `cond ? *p : zero` made hipcc select
            //  between p and a zero kept in SCRATCH and load through flat addressing)
            const bool va = ok[j] && n < d.d0 && k0 < d.d1, vb = ok[j] && n < d.d0 && k0 + 4 < d.d1;
            const float* sp = src + (size_t)n * d.s0 + k0;
            a[j] = *(const float4*)(va ? sp : src);
            b2[j] = *(const float4*)(vb ? sp + 4 : src);
            if (!va) a[j] = z;
            if (!vb) b2[j] = z;
        }
#pragma unroll
        for (int j = 0; j < E2T_PACK_UNITS; ++j) {
            if (!ok[j]) continue;
            const int f = (u0 + j) * 4 + (threadIdx.x >> 6);
            uint4 v;
            v.x = f2bf(a[j].x) | ((unsigned)f2bf(a[j].y) << 16); v.y = f2bf(a[j].z) | ((unsigned)f2bf(a[j].w) << 16);
            v.z = f2bf(b2[j].x) | ((unsigned)f2bf(b2[j].y) << 16); v.w = f2bf(b2[j].z) | ((unsigned)f2bf(b2[j].w) << 16);
            ((uint4*)dst)[(size_t)f * 64 + lane] = v;
        }
        return;
    }
    const int G = (d.kind == 1) ? E2T_PACK_UNITS : 1;
    const int u0 = (bid - d.first_block) * G;
    if (d.kind == 5 || d.kind == 6) {
        // fragment images of sources contiguous along n.  Kind 6: the four per-gate images of a gate-interleaved LSTM kernel (TF
        // layout [k][unit*4 + gate]) from ONE pass over the source; kind 5: four n tiles.  A workgroup takes TWO k-blocks: two
        // [32 k][64 columns] blocks staged with 16-B loads (256 B contiguous per k-row), all four loads of a thread in flight
        // together (one block per workgroup moved 8 KiB per launch-and-round-trip: 0.7 TB/s); wave g then emits the fragments of
        // gate g (kind 6) / n tile ntq*4 + g (kind 5) of both k-blocks: 2 KiB contiguous.  d0 = units (kind 6) or n (kind 5),
        // d1 = K, s1 = k stride, ld = KB; kind 6: image g starts at fragment g*UT*KB.
        const int lb = bid - d.first_block;
        const int KB = d.ld, KBG = (KB + 1) >> 1, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int NT = (d.d0 + 15) / 16;
        const int ut = lb / KBG, kb0 = (lb - ut * KBG) * 2;
        const int n4 = (threadIdx.x & 15) * 4;
        const int ncols = d.kind == 6 ? 4 * d.d0 : d.d0;
        float4 v[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kl = (threadIdx.x >> 4) + 16 * i;
                const int np = ut * 64 + n4, k = (kb0 + u) * 32 + kl;
                const bool in = np < ncols && k < d.d1;
                v[u][i] = *(const float4*)(in ? src + (size_t)np + (size_t)k * d.s1 : src);
                if (!in) v[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) *(float4*)&pk_lds[(u * 32 + (threadIdx.x >> 4) + 16 * i) * 68 + n4] = v[u][i];
        __syncthreads();
        const int nt = ut * 4 + g;
        if (d.kind == 5 && nt >= NT) return;
        const int col = d.kind == 6 ? (lane & 15) * 4 + g : g * 16 + (lane & 15);
        const size_t f0 = d.kind == 6 ? ((size_t)g * NT + ut) * KB + kb0 : (size_t)nt * KB + kb0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (kb0 + u >= KB) break;
            const float* b = &pk_lds[(u * 32 + (lane >> 4) * 8) * 68 + col];
            uint4 o;
            o.x = f2bf(b[0]) | ((unsigned)f2bf(b[68]) << 16); o.y = f2bf(b[2 * 68]) | ((unsigned)f2bf(b[3 * 68]) << 16);
            o.z = f2bf(b[4 * 68]) | ((unsigned)f2bf(b[5 * 68]) << 16); o.w = f2bf(b[6 * 68]) | ((unsigned)f2bf(b[7 * 68]) << 16);
            ((uint4*)dst)[(f0 + u) * 64 + lane] = o;
        }
        return;
    }
    for (int u = u0; u < min(units, u0 + G); ++u) pack_unit(d, u, src, dst, pk_lds);
}
__device__ __forceinline__ void pack_unit(const e2t_pack_desc& d, const int lb, const float* src, bf16_t* dst, float* lds) {
    if (d.kind == 4) {
        // kind 2 with 16-B loads along the source's contiguous direction and 8-B stores (d0 % 4 == 0, d1 % 4 == 0, s1 % 4 == 0)
        float (*tile)[65] = (float (*)[65])lds;              // tile[c][r]
        const int tcn = (d.d1 + 63) / 64;
        const int tr = lb / tcn, tc = lb - tr * tcn;
        const int a4 = (threadIdx.x & 15) * 4, b = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tc * 64 + b + 16 * i, r = tr * 64 + a4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < d.d0 && c < d.d1) v = *(const float4*)(src + (size_t)r + (size_t)c * d.s1);
            float* t = &tile[b + 16 * i][a4];
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = tr * 64 + b + 16 * i, c = tc * 64 + a4;
            if (r < d.d0 && c < d.d1)
                *(ushort4*)(dst + (size_t)r * d.ld + c) = make_ushort4(f2bf(tile[a4][b + 16 * i]), f2bf(tile[a4 + 1][b + 16 * i]),
                                                                         f2bf(tile[a4 + 2][b + 16 * i]), f2bf(tile[a4 + 3][b + 16 * i]));
        }
    } else if (d.kind == 2) {
        float (*tile)[65] = (float (*)[65])lds;
        const int tcn = (d.d1 + 63) / 64;
        const int tr = lb / tcn, tc = lb - tr * tcn;
        const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = tc * 64 + q * 16 + i, r = tr * 64 + l;
            tile[q * 16 + i][l] = (r < d.d0 && c < d.d1) ? src[(size_t)r * d.s0 + (size_t)c * d.s1] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = tr * 64 + q * 16 + i, c = tc * 64 + l;
            if (r < d.d0 && c < d.d1) dst[(size_t)r * d.ld + c] = f2bf(tile[l][q * 16 + i]);
        }
    } else if (d.kind == 0) {
        const int cb = (d.d1 + 255) / 256;                 // column blocks per row
        const int r = lb / cb, c = (lb - r * cb) * 256 + threadIdx.x;
        if (r < d.d0 && c < d.d1) dst[(size_t)r * d.ld + c] = f2bf(src[(size_t)r * d.s0 + (size_t)c * d.s1]);
    } else {
        const int KB = d.ld, lane = threadIdx.x & 63;
        const int f = lb * 4 + (threadIdx.x >> 6);          // fragment index nt*KB + kb
        const int NT = (d.d0 + 15) / 16;
        if (f >= NT * KB) return;
        const int nt = f / KB, kb = f - nt * KB;
        const int n = nt * 16 + (lane & 15);
        bf16_t o[8];
        const int k0 = kb * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            o[j] = (n < d.d0 && k < d.d1) ? f2bf(src[(size_t)n * d.s0 + (size_t)k * d.s1]) : (bf16_t)0;
        }
        uint4 v;
        v.x = o[0] | ((unsigned)o[1] << 16); v.y = o[2] | ((unsigned)o[3] << 16);
        v.z = o[4] | ((unsigned)o[5] << 16); v.w = o[6] | ((unsigned)o[7] << 16);
        ((uint4*)dst)[(size_t)f * 64 + lane] = v;
    }
}

// ---------------------------------------------------------------------------
// a9: decoder embedding gather (+FF dropout) and its scatter-add gradient
// ---------------------------------------------------------------------------
__global__ void k_embed_fwd(const bf16_t* emb, int ld_emb, const int* tok, int M, int E, bf16_t* out, int ld_out, DropCfg drop,
                            int row0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * E) return;
    const int e = i % E; const size_t m = i / E + row0;
    float v = bf2f(emb[(size_t)tok[m] * ld_emb + e]);
    v *= drop_scale(drop, (unsigned long long)(m * E + e));
    out[m * ld_out + e] = f2bf(v);
}
__global__ void k_embed_bwd(const float* de, int ld_de, const int* tok, int M, int E, float* demb, int ld_demb, DropCfg drop) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * E) return;
    const int e = i % E; const size_t m = i / E;
    float g = de[m * ld_de + e] * drop_scale(drop, (unsigned long long)(m * E + e));
    if (g != 0.f) atomicAdd(demb + (size_t)tok[m] * ld_demb + e, g);
}

// ---------------------------------------------------------------------------
// a9: masked softmax cross-entropy over the vocabulary, one wave per row, all
// reductions by wavefront shuffles.  Emits per-row loss, arg-max, and the bf16
// gradient dlogits = (softmax - onehot) * w / ntok for valid rows, 0 otherwise.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax_ce(const float* logits, int ldl, int M, int V, const int* tgt, const int* lens,
                                                     int rowsB, const int* ntok, float w, float* rowloss, int* pred,
                                                     float* correct, bf16_t* dl, int lddl) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* row = logits + (size_t)m * ldl;
    const bool valid = lens ? ((m / rowsB) < lens[m % rowsB]) : true;
    // the row stays in registers when it fits (V <= 2048: 32 values per lane): ONE pass over memory with all loads in
    // flight instead of three dependent passes; same reduction order either way
    constexpr int NR = 32;
    const bool inreg = V <= NR * 64;
    float xr[NR];
    if (inreg) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { const int v = lane + 64 * i; xr[i] = (v < V) ? row[v] : -INFINITY; }
    }
    float mx = -INFINITY; int arg = 0;
    if (inreg) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { if (xr[i] > mx) { mx = xr[i]; arg = lane + 64 * i; } }
    } else {
        for (int v = lane; v < V; v += 64) { float x = row[v]; if (x > mx) { mx = x; arg = v; } }
    }
    // wave arg-max with lowest-index tie-break (matches numpy argmax)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float omx = __shfl_xor(mx, o, 64); int oarg = __shfl_xor(arg, o, 64);
        if (omx > mx || (omx == mx && oarg < arg)) { mx = omx; arg = oarg; }
    }
    float se = 0.f;
    if (inreg) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { if (lane + 64 * i < V) se += __expf(xr[i] - mx); }
    } else {
        for (int v = lane; v < V; v += 64) se += __expf(row[v] - mx);
    }
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    const int t = tgt ? tgt[m] : 0;
    if (lane == 0) {
        if (rowloss) rowloss[m] = (valid && tgt) ? (lse - row[t]) : 0.f;
        if (pred) pred[m] = arg;
        if (correct) correct[m] = (valid && tgt && arg == t) ? 1.f : 0.f;
    }
    if (dl) {
        const float sc = valid ? w / (float)max(ntok ? *ntok : 1, 1) : 0.f;
        bf16_t* drow = dl + (size_t)m * lddl;
        if (inreg) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int v = lane + 64 * i;
                if (v < V) drow[v] = f2bf((__expf(xr[i] - lse) - (v == t ? 1.f : 0.f)) * sc);
            }
        } else {
            for (int v = lane; v < V; v += 64) {
                float pr = __expf(row[v] - lse);
                drow[v] = f2bf((pr - (v == t ? 1.f : 0.f)) * sc);
            }
        }
    }
}

// a8 Gaussian head: rowloss[m] = sum_k (P - A)^2 on valid rows; dP = 2 (P - A) w / (nval K)
__global__ void k_mse(const float* P, int ldp, const float* At, int M, int Kf, const int* lens, int rowsB, const int* nval,
                      float w, float* rowloss, bf16_t* dP, int lddp) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const bool valid = (m / rowsB) < lens[m % rowsB];
    const float sc = valid ? 2.f * w / ((float)max(*nval, 1) * Kf) : 0.f;
    float acc = 0.f;
    for (int k = 0; k < Kf; ++k) {
        const float d = valid ? (P[(size_t)m * ldp + k] - At[(size_t)m * Kf + k]) : 0.f;
        acc += d * d;
        if (dP) dP[(size_t)m * lddp + k] = f2bf(d * sc);
    }
    rowloss[m] = acc;
}

// ---------------------------------------------------------------------------
// encoder -> decoder seam: final (c,h) of the last encoder layer at each
// utterance's own last valid step (plotters.py:1388)
// ---------------------------------------------------------------------------
__global__ void k_final_state(const bf16_t* Yext, int ldy, const float* Cs, const int* lens, int B, int H, int H8,
                              bf16_t* h0, int ldh0, float* c0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * H) return;
    const int b = i / (2 * H), r = i % (2 * H), d = r / H, u = r % H;
    const int len = lens[b];
    bf16_t h = 0; float c = 0.f;
    if (len > 0) {
        const int t = d ? 0 : len - 1;             // time index of the direction's last processed step
        h = Yext[((size_t)(t + 1) * B + b) * ldy + d * H8 + u];
        // lane-native c save (lstm.hip): processing step len-1, tile (rt, ut), half (u&3)>>1, lane (fq, frow), component u&1
        const int RT = (B + 15) >> 4, UT = (H + 15) >> 4;
        const size_t tile = ((size_t)((len - 1) * 2 + d) * RT + (b >> 4)) * UT + (u >> 4);
        const int lane = (((u & 15) >> 2) << 4) + (b & 15);
        c = Cs[((tile * 2 + ((u & 3) >> 1)) * 64 + lane) * 2 + (u & 1)];
    }
    h0[(size_t)b * ldh0 + r] = h;
    c0[(size_t)b * 2 * H + r] = c;
}

// greedy decoding bookkeeping (beam_width 1): record token, latch <EOS>, feed next input
__global__ void k_greedy_update(const int* pred, int B, int l, int Lmax, int eos, int pad, int* done, int* out, int* next_tok) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int tok = pred[b];
    const int dn = done[b];
    out[(size_t)b * Lmax + l] = dn ? pad : tok;
    done[b] = dn | (tok == eos);
    if (next_tok) next_tok[b] = tok;
}

// the same in one launch straight from the logits: greedy search needs the arg-max only (lowest index on ties, as k_softmax_ce and
// numpy), not the normalised probabilities -- one wave per utterance
__global__ __launch_bounds__(256) void k_greedy_step(const float* logits, int ldl, int B, int V, int l, int Lmax, int eos, int pad, int* done,
                                                     int* out, int* next_tok) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* row = logits + (size_t)b * ldl;
    float mx = -INFINITY; int arg = 0;
    if (V <= 32 * 64) {                                  // the row in registers: all loads in flight together (k_softmax_ce does the same)
        float xr[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { const int v = lane + 64 * i; xr[i] = (v < V) ? row[v] : -INFINITY; }
#pragma unroll
        for (int i = 0; i < 32; ++i) if (xr[i] > mx) { mx = xr[i]; arg = lane + 64 * i; }
    } else {
        for (int v = lane; v < V; v += 64) { const float x = row[v]; if (x > mx) { mx = x; arg = v; } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float omx = __shfl_xor(mx, o, 64); const int oarg = __shfl_xor(arg, o, 64);
        if (omx > mx || (omx == mx && oarg < arg)) { mx = omx; arg = oarg; }
    }
    if (lane == 0) {
        const int dn = done[b];
        out[(size_t)b * Lmax + l] = dn ? pad : arg;
        done[b] = dn | (arg == eos);
        if (next_tok) next_tok[b] = arg;
    }
}

// ---------------------------------------------------------------------------
// ABI 9: greedy decoding of FEW utterances (the online predictor decodes ONE per call, ecog2txt/trainers.py:925-949).  At B = 1 a
// decoder step was four launches -- row gather, recurrence step, a 15-tile projection GEMM whose 128-row tiles hold one real row
// (12.8 us), arg-max (4.6 us) -- on an otherwise idle chip: 30 us per token, all of it launch latency and padding.  Here the head
// of a step is ONE launch: a matrix-vector product on the vector units (4 lanes per vocabulary row, 16 rows per wave pass, the
// state in LDS as fp32), the arg-max through a two-level reduction (wave shuffles, LDS, then one 8-byte {value, index} per
// workgroup and utterance written through to L2 and a ticket: the LAST workgroup to arrive reduces them), the bookkeeping of
// k_greedy_step, and the copy of the chosen token's input-projection row for the next step.
// ---------------------------------------------------------------------------
#define E2T_HEAD_MAXB 8
__global__ __launch_bounds__(256) void k_decode_init(int* done, int* hyp, int* tok0, int* dlens, int B, int Lmax, int eos, int pad) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * Lmax; i += gridDim.x * 256) hyp[i] = pad;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) { done[b] = 0; tok0[b] = eos; dlens[b] = Lmax; }
}
__device__ __forceinline__ unsigned long long head_key(float v, int idx) { return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)idx; }
__device__ __forceinline__ bool head_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }
__global__ __launch_bounds__(256) void k_greedy_head_small(const bf16_t* h, int ldh, const bf16_t* WT, int ldw, const float* bias, int B, int V, int K,
                                                           int l, int Lmax, int eos, int pad, int* done, int* out, int* next_tok,
                                                           const unsigned* table, size_t row_words, unsigned* gx_next, unsigned* scratch) {
    extern __shared__ uint4 head_lds[];                 // [B][K8 / 8] the state as it lies (bf16), then [4 waves][B] {value, index}
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K8 = (K + 7) & ~7, nchunk = K8 >> 3;
    {
        bf16_t* hs = (bf16_t*)head_lds;
        for (int i = threadIdx.x; i < B * K8; i += 256) {
            const int b = i / K8, k = i - b * K8;
            hs[i] = k < K ? h[(size_t)b * ldh + k] : (bf16_t)0;
        }
    }
    __syncthreads();
    float bestv[E2T_HEAD_MAXB]; int besti[E2T_HEAD_MAXB];
#pragma unroll
    for (int b = 0; b < E2T_HEAD_MAXB; ++b) { bestv[b] = -INFINITY; besti[b] = 0x7fffffff; }
    // 8 lanes per vocabulary row (lane q takes the 16-B chunks q, q + 8, ...), 8 rows per wave pass; a lane's loads of a pass are
    // issued together, eight at a time (one dependent load per loop trip made the launch 20 us: 25 round trips to L2 in a row)
    const int q = lane & 7, r = lane >> 3;
    for (int v0 = (blockIdx.x * 4 + wave) * 8; v0 < V; v0 += gridDim.x * 32) {
        const int v = v0 + r;
        const uint4* wrow = (const uint4*)(WT + (size_t)min(v, V - 1) * ldw);
        float acc[E2T_HEAD_MAXB];
#pragma unroll
        for (int b = 0; b < E2T_HEAD_MAXB; ++b) acc[b] = 0.f;
        for (int c0 = q; c0 < nchunk; c0 += 64) {
            uint4 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int c = c0 + 8 * j; w[j] = c < nchunk ? wrow[c] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + 8 * j;
                if (c < nchunk) {
                    const float wf[8] = {__uint_as_float(w[j].x << 16), __uint_as_float(w[j].x & 0xFFFF0000u), __uint_as_float(w[j].y << 16), __uint_as_float(w[j].y & 0xFFFF0000u),
                                         __uint_as_float(w[j].z << 16), __uint_as_float(w[j].z & 0xFFFF0000u), __uint_as_float(w[j].w << 16), __uint_as_float(w[j].w & 0xFFFF0000u)};
#pragma unroll
                    for (int b = 0; b < E2T_HEAD_MAXB; ++b) {
                        if (b < B) {
                            const uint4 x = head_lds[b * nchunk + c];
                            float a = acc[b];
                            a = fmaf(wf[0], __uint_as_float(x.x << 16), a); a = fmaf(wf[1], __uint_as_float(x.x & 0xFFFF0000u), a);
                            a = fmaf(wf[2], __uint_as_float(x.y << 16), a); a = fmaf(wf[3], __uint_as_float(x.y & 0xFFFF0000u), a);
                            a = fmaf(wf[4], __uint_as_float(x.z << 16), a); a = fmaf(wf[5], __uint_as_float(x.z & 0xFFFF0000u), a);
                            a = fmaf(wf[6], __uint_as_float(x.w << 16), a); a = fmaf(wf[7], __uint_as_float(x.w & 0xFFFF0000u), a);
                            acc[b] = a;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < E2T_HEAD_MAXB; ++b) {
            if (b < B) {
                float a = acc[b];
                a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);      // the row's eight partial sums (fixed order)
                if (v < V) {
                    a += bias ? bias[v] : 0.f;
                    if (head_better(a, v, bestv[b], besti[b])) { bestv[b] = a; besti[b] = v; }
                }
            }
        }
    }
    // wave -> workgroup
    float* red = (float*)(head_lds + B * nchunk);        // [4][B][2]
#pragma unroll
    for (int b = 0; b < E2T_HEAD_MAXB; ++b) {
        if (b < B) {
            float mv = bestv[b]; int mi = besti[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(mv, o, 64); const int oi = __shfl_xor(mi, o, 64);
                if (head_better(ov, oi, mv, mi)) { mv = ov; mi = oi; }
            }
            if (lane == 0) { red[(wave * B + b) * 2] = mv; red[(wave * B + b) * 2 + 1] = __int_as_float(mi); }
        }
    }
    __syncthreads();
    unsigned long long* part = (unsigned long long*)(scratch + 2);        // [gridDim.x][B] {value, index}; scratch[0] = ticket
    if (threadIdx.x < B) {
        const int b = threadIdx.x;
        float mv = red[b * 2]; int mi = __float_as_int(red[b * 2 + 1]);
        for (int w = 1; w < 4; ++w) {
            const float ov = red[(w * B + b) * 2]; const int oi = __float_as_int(red[(w * B + b) * 2 + 1]);
            if (head_better(ov, oi, mv, mi)) { mv = ov; mi = oi; }
        }
        __hip_atomic_store(part + (size_t)blockIdx.x * B + b, head_key(mv, mi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through stores have reached L2 before the ticket is taken
    __syncthreads();
    __shared__ unsigned ticket;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(scratch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    // ---- the last workgroup: final arg-max, bookkeeping, next step's input-projection rows ----
    __shared__ int tok[E2T_HEAD_MAXB];
    __shared__ unsigned long long allp[64 * E2T_HEAD_MAXB];
    // (every partial by a thread of its own: read one after the other by the B reducing threads they were 57 round trips to L2 in a
    //  row, 30 us)
    for (unsigned i = threadIdx.x; i < gridDim.x * (unsigned)B; i += 256)
        allp[i] = __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x < B) {
        const int b = threadIdx.x;
        float mv = -INFINITY; int mi = 0x7fffffff;
        for (unsigned g = 0; g < gridDim.x; ++g) {
            const unsigned long long kx = allp[g * B + b];
            const float ov = __uint_as_float((unsigned)(kx >> 32)); const int oi = (int)(unsigned)kx;
            if (head_better(ov, oi, mv, mi)) { mv = ov; mi = oi; }
        }
        const int arg = (mi == 0x7fffffff) ? 0 : mi;
        const int dn = done[b];
        out[(size_t)b * Lmax + l] = dn ? pad : arg;
        done[b] = dn | (arg == eos);
        if (next_tok) next_tok[b] = arg;
        tok[b] = arg;
    }
    if (threadIdx.x == 0) __hip_atomic_store(scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the ticket word for the next launch
    __syncthreads();
    if (table && gx_next) {
        // the chosen tokens' input-projection rows, all loads of a thread in flight together (a load -> store loop was one round
        // trip to L2 per trip: 6 us per utterance)
        if ((row_words & 3) == 0 && ((((uintptr_t)table) | ((uintptr_t)gx_next)) & 15) == 0) {
            const size_t rq = row_words >> 2, total = (size_t)B * rq;
            for (size_t i0 = threadIdx.x; i0 < total; i0 += 256 * 4) {
                uint4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const size_t i = i0 + 256 * j;
                    if (i < total) { const int b = (int)(i / rq); v[j] = ((const uint4*)(table + (size_t)tok[b] * row_words))[i - b * rq]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const size_t i = i0 + 256 * j;
                    if (i < total) { const int b = (int)(i / rq); ((uint4*)(gx_next + (size_t)b * row_words))[i - b * rq] = v[j]; }
                }
            }
        } else {
            for (int b = 0; b < B; ++b) {
                const unsigned* src = table + (size_t)tok[b] * row_words;
                unsigned* dst = gx_next + (size_t)b * row_words;
                for (size_t i = threadIdx.x; i < row_words; i += 256) dst[i] = src[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// a9, beam_width > 1 (mocha-1_word_sequence.yaml:31; temperature :82): one step of beam search for utterance b = blockIdx.x.
// Rows b*W + w of `logits` are the W live hypotheses.  Candidates: for a live beam w every token v, scored
// score[w] + log softmax(logits[w] / temperature)[v]; for a finished beam only "stay finished" (score unchanged).  The W best
// survive -- ties: lower beam first, within a beam the stay candidate, then lower token ids (oracle/seq2seq.py: beam_decode).
// Outputs: the new scores / finished flags / token histories (gathered from the parents), rowmap[b*W + k] = row of the parent
// (for the state reorder), next_tok = the input tokens of the next step.  All arrays are double-buffered by the caller.
// ---------------------------------------------------------------------------
#define E2T_BEAM_MAX 16
__global__ __launch_bounds__(256) void k_beam_step(const float* logits, int ldl, int W, int V, float inv_temp, int l, int Lmax, int eos, int pad,
                                                   const float* score_in, const int* done_in, const int* hyp_in,
                                                   float* score_out, int* done_out, int* hyp_out, int* rowmap, int* next_tok) {
    __shared__ float s_lse[E2T_BEAM_MAX], s_score[E2T_BEAM_MAX];
    __shared__ int s_done[E2T_BEAM_MAX];
    __shared__ float s_rv[4]; __shared__ int s_ri[4];
    __shared__ int s_sel[E2T_BEAM_MAX]; __shared__ float s_selv[E2T_BEAM_MAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < W) { s_score[tid] = score_in[b * W + tid]; s_done[tid] = done_in[b * W + tid]; }
    // log-sum-exp of every beam's row (scaled by 1 / temperature): wave wv takes beams wv, wv + 4, ...
    for (int w = wv; w < W; w += 4) {
        const float* row = logits + (size_t)(b * W + w) * ldl;
        float mx = -INFINITY;
        for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v] * inv_temp);
        mx = wave_max(mx);
        float se = 0.f;
        for (int v = lane; v < V; v += 64) se += __expf(row[v] * inv_temp - mx);
        se = wave_sum(se);
        if (lane == 0) s_lse[w] = mx + __logf(se);
    }
    __syncthreads();
    const int NC = W * (V + 1);                  // candidate id = w * (V + 1) + (token + 1); token -1 = stay finished
    for (int k = 0; k < W; ++k) {
        float bv = -INFINITY; int bi = 0x7FFFFFFF;
        for (int c = tid; c < NC; c += 256) {
            const int w = c / (V + 1), tv = c - w * (V + 1) - 1;
            bool taken = false;
            for (int q = 0; q < k; ++q) taken |= (s_sel[q] == c);
            if (taken) continue;
            float val;
            if (tv < 0) val = s_done[w] ? s_score[w] : -INFINITY;
            else val = s_done[w] ? -INFINITY : s_score[w] + (logits[(size_t)(b * W + w) * ldl + tv] * inv_temp - s_lse[w]);
            if (val > bv || (val == bv && c < bi)) { bv = val; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_rv[wv] = bv; s_ri[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 4; ++q) if (s_rv[q] > bv || (s_rv[q] == bv && s_ri[q] < bi)) { bv = s_rv[q]; bi = s_ri[q]; }
            s_sel[k] = bi; s_selv[k] = bv;
        }
        __syncthreads();
    }
    if (tid < W) {
        const int c = s_sel[tid];
        const int w = (c == 0x7FFFFFFF) ? 0 : c / (V + 1);
        const int tv = (c == 0x7FFFFFFF) ? -1 : c - w * (V + 1) - 1;
        const bool stay = tv < 0, was = s_done[w] != 0;
        score_out[b * W + tid] = s_selv[tid];
        done_out[b * W + tid] = (was || stay || tv == eos) ? 1 : 0;
        rowmap[b * W + tid] = b * W + w;
        if (next_tok) next_tok[b * W + tid] = stay ? eos : tv;
        const int* hi = hyp_in + (size_t)(b * W + w) * Lmax;
        int* ho = hyp_out + (size_t)(b * W + tid) * Lmax;
        for (int j = 0; j < Lmax; ++j) ho[j] = (j == l) ? ((stay || was) ? pad : tv) : hi[j];
    }
}
// The decoder state of the surviving hypotheses: h (row-major block of the ext output array) and c (the lane-native save of the
// step, csrc/lstm.hip) of row r <- those of row rowmap[r].  mode 0: state -> scratch; mode 1: state[r] <- scratch[rowmap[r]].
__global__ void k_beam_reorder(bf16_t* Yblk, int ldy, float* Cs_step, int B, int H, int H8, const int* rowmap, bf16_t* tmp_h, float* tmp_c, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, u = i - b * H;
    const int UT = (H + 15) >> 4;
    auto cidx = [&](int bb) {
        const size_t tile = (size_t)(bb >> 4) * UT + (u >> 4);
        const int lane = (((u & 15) >> 2) << 4) + (bb & 15);
        return ((tile * 2 + ((u & 3) >> 1)) * 64 + lane) * 2 + (u & 1);
    };
    if (mode == 0) {
        tmp_h[(size_t)b * H8 + u] = Yblk[(size_t)b * ldy + u];
        tmp_c[(size_t)b * H + u] = Cs_step[cidx(b)];
    } else {
        const int src = rowmap[b];
        Yblk[(size_t)b * ldy + u] = tmp_h[(size_t)src * H8 + u];
        Cs_step[cidx(b)] = tmp_c[(size_t)src * H + u];
    }
}

// ---------------------------------------------------------------------------
// a10: fused Adam (TF1 AdamOptimizer form) + EMA shadow over a flat fp32 range
// ---------------------------------------------------------------------------
__global__ void k_inc_step(int* step, const int* skip) { if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip && *skip != 0)) step[0] += 1; }

// 16-B accesses: the five arrays are the same element range of five equally aligned flat buffers, so one scalar head (up to the
// first 16-B boundary) and one scalar tail frame a float4 body (cfg2: 523 MB per bulk launch; the scalar form moved 4.7 TB/s).
// Per element the arithmetic is the scalar form's, operation for operation.
__device__ __forceinline__ void adam_ema_1(float& p, float g, float& m, float& v, float& e, float lr_t, float b1, float b2, float eps,
                                           float decay, float gscale) {
    // (contraction off, the fused multiply-adds spelled out: this function is inlined into two kernels -- k_adam_ema and k_adam_pack --
    //  whose results must agree bit for bit, and the compiler forms a*b + c into an fma or not depending on the code around it)
#pragma clang fp contract(off)
    const float gi = g * gscale;
    const float mi = fmaf(b1, m, (1.f - b1) * gi);
    const float vi = fmaf(b2, v, ((1.f - b2) * gi) * gi);
    const float pi = p - (lr_t * mi) / (sqrtf(vi) + eps);
    m = mi; v = vi; p = pi;
    e = fmaf(decay, e, (1.f - decay) * pi);
}
__global__ __launch_bounds__(256) void k_adam_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n,
                                                   const int* step, float lr, float b1, float b2, float eps, float decay,
                                                   float gscale, int step_offset, const int* skip, int vec) {
    if (skip && *skip != 0) return;          // the step's gradients are invalid (in-kernel wait timed out): no update
    const float t = (float)(*step + step_offset);
    const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
    size_t head = vec ? (((16 - ((uintptr_t)p & 15)) & 15) >> 2) : n;
    if (head > n) head = n;
    const size_t n4 = (n - head) >> 2, tail0 = head + (n4 << 2);
    for (size_t i = tid; i < head; i += nthr) adam_ema_1(p[i], g[i], m[i], v[i], ema[i], lr_t, b1, b2, eps, decay, gscale);
    for (size_t i = tail0 + tid; i < n; i += nthr) adam_ema_1(p[i], g[i], m[i], v[i], ema[i], lr_t, b1, b2, eps, decay, gscale);
    float4* p4 = (float4*)(p + head); const float4* g4 = (const float4*)(g + head);
    float4* m4 = (float4*)(m + head); float4* v4 = (float4*)(v + head); float4* e4 = (float4*)(ema + head);
    // g, m, v and the EMA shadow are touched exactly once per step: non-temporal loads and stores (they do not displace what the
    // concurrent weight-gradient GEMMs live on: cfg2 1.601 -> 1.585 ms, cfg4 8.43 -> 8.36 ms same box; alone 633 -> 615 us for
    // 98 M parameters = 5.7 TB/s).  p is re-read by the re-pack right behind the update: default policy.
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (size_t i = tid; i < n4; i += nthr) {
        const f4 pv = ((f4*)p4)[i];
        const f4 mv = __builtin_nontemporal_load((f4*)m4 + i), vv_ = __builtin_nontemporal_load((f4*)v4 + i);
        const f4 ev = __builtin_nontemporal_load((f4*)e4 + i);
        const f4 gg = __builtin_nontemporal_load((const f4*)g4 + i);
        float P[4] = {pv.x, pv.y, pv.z, pv.w}, Mv[4] = {mv.x, mv.y, mv.z, mv.w}, V[4] = {vv_.x, vv_.y, vv_.z, vv_.w}, E[4] = {ev.x, ev.y, ev.z, ev.w};
        const float G[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_ema_1(P[k], G[k], Mv[k], V[k], E[k], lr_t, b1, b2, eps, decay, gscale);
        const f4 pp = {P[0], P[1], P[2], P[3]}, mm = {Mv[0], Mv[1], Mv[2], Mv[3]}, vv = {V[0], V[1], V[2], V[3]}, ee = {E[0], E[1], E[2], E[3]};
        __builtin_nontemporal_store(mm, (f4*)m4 + i); __builtin_nontemporal_store(vv, (f4*)v4 + i);
        ((f4*)p4)[i] = pp;
        __builtin_nontemporal_store(ee, (f4*)e4 + i);
    }
}

// ---------------------------------------------------------------------------
// a10 + operand images in one pass (ABI 8, include/ecog2txt_hip.h e2t_adam_pack_batch): the HBM-bound tail of the train step.
// k_adam_ema writes the masters (36 B per parameter moved) and k_pack_batch reads them again once per image (2 x 6 B); here a
// workgroup owns a 64 x 64 tile of a parameter matrix, updates it with adam_ema_1 -- the same expression, so the same bits --
// and emits the tile's share of every image from the values it holds: 40 B per parameter.  The tile goes through LDS as fp32
// [64][65] (both the row-wise and the column-wise reads below are conflict-free or 2-way); fragments are written as whole 1-KiB
// wave transactions.
// ---------------------------------------------------------------------------
struct AdamPackArgs { const int* step; const int* skip; float lr, b1, b2, eps, decay, gscale; int step_offset; int update; };
__global__ __launch_bounds__(256) void k_adam_pack(const e2t_tile_desc* descs, int ndesc, float* p, const float* g, float* m, float* v,
                                                    float* ema, AdamPackArgs a) {
    // the tile as bf16 (the images' own precision) [64][66]: 8.3 KB -- three workgroups fit beside two resident GEMM workgroups of the
    // other branch (2 x 64 KB of LDS); as fp32 (16.6 KB) one did, and the update crawled while a GEMM was on the chip
    __shared__ bf16_t tile[64][66];
    if (a.update && a.skip && *a.skip != 0) return;      // invalid gradients: no update, and the images of the old masters stay
    int lo = 0, hi = ndesc - 1;
    const int bid = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].first_block <= bid) lo = mid; else hi = mid - 1; }
    // (scalar fields only: a local copy of the descriptor, indexed by the image loop below, would live in scratch)
    struct { int R, C, nimg; long long src_off, s0; } d = {descs[lo].R, descs[lo].C, descs[lo].nimg, descs[lo].src_off, descs[lo].s0};
    const float* gslab = descs[lo].gslab;                  // the gradient as the slabs a split K-major product left (or null: g)
    const long long gstride = descs[lo].gstride;
    const int gsplits = descs[lo].gsplits;
    const int lb = bid - descs[lo].first_block;
    const int tcn = (d.C + 63) >> 6;
    const int tr = lb / tcn, tc = lb - tr * tcn;
    if (tr >= ((d.R + 63) >> 6)) return;
    const int b = threadIdx.x >> 4, a4 = (threadIdx.x & 15) * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    float lr_t = 0.f;
    if (a.update) {
        const float t = (float)(*a.step + a.step_offset);
        lr_t = a.lr * sqrtf(1.f - powf(a.b2, t)) / (1.f - powf(a.b1, t));
    }
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    // One 16-B group of each buffer per thread and pass, four passes, NOT unrolled: like k_adam_ema the kernel lives on the number of
    // workgroups in flight, not on loads in flight per thread (unrolled, 20 groups per thread cost 172 registers = two workgroups
    // per CU: 3.9 TB/s against the 6.1 of the separate kernels)
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int r = tr * 64 + b + 16 * i, c = tc * 64 + a4;
        const bool in = r < d.R && c < d.C;                    // (C % 4 == 0: a 16-B group is inside or outside as a whole)
        const size_t idx = in ? (size_t)d.src_off + (size_t)r * (size_t)d.s0 + c : 0;      // (outside: the buffers' first 16 B, zeroed below)
        f4 pv = *(const f4*)(p + idx);
        if (a.update) {
            f4 gv;
            if (gslab) {
                // sum of the product's split-K slabs in split order: the bits k_splitk_reduce would have written to g
                const float* q = gslab + (in ? (size_t)(tr * 64 + b + 16 * i) * (size_t)d.s0 + (tc * 64 + a4) : 0);
                gv = __builtin_nontemporal_load((const f4*)q);
                for (int sp = 1; sp < gsplits; ++sp) gv += __builtin_nontemporal_load((const f4*)(q + (size_t)sp * gstride));
            } else {
                gv = __builtin_nontemporal_load((const f4*)(g + idx));
            }
            const f4 mv = __builtin_nontemporal_load((const f4*)(m + idx));
            const f4 vv = __builtin_nontemporal_load((const f4*)(v + idx));
            const f4 ev = __builtin_nontemporal_load((const f4*)(ema + idx));
            if (in) {
                float P[4] = {pv.x, pv.y, pv.z, pv.w}, Mv[4] = {mv.x, mv.y, mv.z, mv.w}, V[4] = {vv.x, vv.y, vv.z, vv.w}, E[4] = {ev.x, ev.y, ev.z, ev.w};
                const float G[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) adam_ema_1(P[k], G[k], Mv[k], V[k], E[k], lr_t, a.b1, a.b2, a.eps, a.decay, a.gscale);
                pv = (f4){P[0], P[1], P[2], P[3]};
                const f4 mm = {Mv[0], Mv[1], Mv[2], Mv[3]}, v2 = {V[0], V[1], V[2], V[3]}, ee = {E[0], E[1], E[2], E[3]};
                __builtin_nontemporal_store(mm, (f4*)(m + idx)); __builtin_nontemporal_store(v2, (f4*)(v + idx));
                *(f4*)(p + idx) = pv;
                __builtin_nontemporal_store(ee, (f4*)(ema + idx));
            }
        }
        if (!in) pv = z4;
        unsigned* t = (unsigned*)&tile[b + 16 * i][a4];           // (row stride 132 B, a4 multiple of 4: 4-B aligned pairs)
        t[0] = f2bf(pv.x) | ((unsigned)f2bf(pv.y) << 16); t[1] = f2bf(pv.z) | ((unsigned)f2bf(pv.w) << 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, q = lane >> 4;
    for (int ii = 0; ii < d.nimg; ++ii) {
        const e2t_tile_img im = descs[lo].img[ii];
        bf16_t* dst = (bf16_t*)im.dst;
        if (im.kind == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = tr * 64 + b + 16 * i, c = tc * 64 + a4;
                if (r >= d.R || c >= d.C) continue;
                const unsigned* t = (const unsigned*)&tile[b + 16 * i][a4];
                *(uint2*)(dst + (size_t)r * im.ld + c) = make_uint2(t[0], t[1]);
            }
        } else if (im.kind == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tc * 64 + b + 16 * i, r = tr * 64 + a4;           // output row = matrix column c, 4 consecutive matrix rows
                if (c >= d.C || r >= d.R) continue;
                const bf16_t o0 = tile[a4][b + 16 * i], o1 = tile[a4 + 1][b + 16 * i], o2 = tile[a4 + 2][b + 16 * i], o3 = tile[a4 + 3][b + 16 * i];
                bf16_t* o = dst + (size_t)c * im.ld + r;
                if (r + 3 < d.R) *(ushort4*)o = make_ushort4(o0, o1, o2, o3);
                else { o[0] = o0; if (r + 1 < d.R) o[1] = o1; if (r + 2 < d.R) o[2] = o2; }
            }
        } else {
            // fragment images: wave w emits two fragments (k-blocks kbl = 0, 1 of the tile) of its n tile / gate
            const int KB = im.ld;
#pragma unroll
            for (int kbl = 0; kbl < 2; ++kbl) {
                bf16_t x[8];
                size_t f;
                if (im.kind == 3) {                            // n = row, k = column
                    const int nt = tr * 4 + w, kb = tc * 2 + kbl;
                    if (nt >= ((d.R + 15) >> 4) || kb >= KB) continue;
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = tile[w * 16 + l15][kbl * 32 + q * 8 + j];
                    f = (size_t)nt * KB + kb;
                } else if (im.kind == 4) {                     // n = column, k = row
                    const int nt = tc * 4 + w, kb = tr * 2 + kbl;
                    if (nt >= ((d.C + 15) >> 4) || kb >= KB) continue;
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = tile[kbl * 32 + q * 8 + j][w * 16 + l15];
                    f = (size_t)nt * KB + kb;
                } else {                                       // gate-interleaved: wave = gate, unit tile = tile column
                    const int UT = ((d.C >> 2) + 15) >> 4, kb = tr * 2 + kbl;
                    if (tc >= UT || kb >= KB) continue;
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = tile[kbl * 32 + q * 8 + j][l15 * 4 + w];
                    f = ((size_t)w * UT + tc) * KB + kb;
                }
                uint4 o;
                o.x = x[0] | ((unsigned)x[1] << 16); o.y = x[2] | ((unsigned)x[3] << 16);
                o.z = x[4] | ((unsigned)x[5] << 16); o.w = x[6] | ((unsigned)x[7] << 16);
                ((uint4*)dst)[f * 64 + lane] = o;
            }
        }
    }
}

// fill of 32-bit words (zeroing the scatter-add target of the embedding gradient, decode-state resets): inside the
// captured step this replaces the framework's own fill kernels
__global__ __launch_bounds__(256) void k_fill_u32(unsigned* dst, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

// ---------------------------------------------------------------------------
// a2/a3 batch assembly: rows of a partition kept resident in HBM -> the step's batch buffers.  One workgroup column
// per destination row; idx < 0 (or beyond the list) = padding utterance: the row is zero-filled, which is exactly what
// the length kernels read as "no samples".  16-B accesses when the row size allows, HBM-bound copy.
// ---------------------------------------------------------------------------
// blockIdx.z = block: the same row selection applied to `gridDim.z` equally shaped blocks of rows (bf16-staged inputs: block t'
// holds decimated step t' of every utterance of a partition, the batch's rows of that step are gathered into block t' of the
// time-major operand)
__global__ __launch_bounds__(256) void k_gather_rows(const unsigned* src, const int* idx, int n, size_t row_words, unsigned* dst,
                                                      size_t src_block_words, size_t dst_block_words) {
    const int r = blockIdx.y;
    const int i = (r < n) ? idx[r] : -1;
    src += (size_t)blockIdx.z * src_block_words;
    dst += (size_t)blockIdx.z * dst_block_words;
    unsigned* d = dst + (size_t)r * row_words;
    const unsigned* sp = (i >= 0) ? src + (size_t)i * row_words : nullptr;
    const bool v4 = (row_words & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    if (v4) {
        const size_t n4 = row_words >> 2;
        for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
            if (sp) v = __builtin_nontemporal_load((const u32x4_t*)sp + k);      // read once: keep it out of the caches
            ((u32x4_t*)d)[k] = v;
        }
    } else {
        for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < row_words; k += (size_t)gridDim.x * 256) d[k] = sp ? sp[k] : 0u;
    }
}

// ---------------------------------------------------------------------------
// C ABI wrappers
// ---------------------------------------------------------------------------
#define ST ((hipStream_t)stream)

extern "C" int e2t_fill_u32(void* dst, size_t n, uint32_t value, void* stream) {
    E2T_CHECK_ARG(dst || n == 0);
    if (n == 0) return E2T_OK;
    size_t blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)dst, n, value);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_gather_rows_u32(const void* src, const int32_t* idx, int n, int rows_out, size_t row_words, void* dst, void* stream) {
    E2T_CHECK_ARG(src && dst && (idx || n == 0) && n >= 0 && rows_out >= n);
    if (rows_out == 0 || row_words == 0) return E2T_OK;
    size_t per = ((row_words & 3) == 0 ? row_words >> 2 : row_words);
    unsigned bx = (unsigned)((per + 255) / 256); if (bx > 64) bx = 64; if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_gather_rows, dim3(bx, rows_out), dim3(256), 0, (hipStream_t)stream, (const unsigned*)src, idx, n, row_words, (unsigned*)dst,
                       (size_t)0, (size_t)0);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_gather_rows_blocks_u32(const void* src, const int32_t* idx, int n, int rows_out, size_t row_words, int blocks,
                                          size_t src_block_words, size_t dst_block_words, void* dst, void* stream) {
    E2T_CHECK_ARG(src && dst && (idx || n == 0) && n >= 0 && rows_out >= n && blocks >= 0 && blocks <= 65535);
    E2T_CHECK_ARG(dst_block_words >= (size_t)rows_out * row_words);
    E2T_CHECK_ARG((row_words & 3) != 0 || ((src_block_words | dst_block_words) & 3) == 0);      // 16-B rows stay 16-B aligned in every block
    if (rows_out == 0 || row_words == 0 || blocks == 0) return E2T_OK;
    size_t per = ((row_words & 3) == 0 ? row_words >> 2 : row_words);
    unsigned bx = (unsigned)((per + 255) / 256); if (bx > 64) bx = 64; if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_gather_rows, dim3(bx, rows_out, blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned*)src, idx, n, row_words,
                       (unsigned*)dst, src_block_words, dst_block_words);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_seq_lengths_f32(const float* x, int B, int T, int C, int div, int32_t* lens, int32_t* lens_div, void* stream) {
    E2T_CHECK_ARG(x && lens && B > 0 && T > 0 && C > 0 && div > 0);
    hipLaunchKernelGGL(k_seq_lengths_f32, dim3(B), dim3(1024), 0, ST, x, T, C, div, lens, lens_div);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_seq_lengths_tail_f32(const float* x, int B, int T, int C, int div, int32_t* lens, int32_t* lens_div, void* stream) {
    E2T_CHECK_ARG(x && lens && B > 0 && T > 0 && C > 0 && div > 0);
    if ((C & 3) == 0 && C >= 32 && (((uintptr_t)x) & 15) == 0)
        hipLaunchKernelGGL(k_seq_lengths_tail_f32, dim3(B), dim3(1024), 0, ST, x, T, C, div, lens, lens_div);
    else
        hipLaunchKernelGGL(k_seq_lengths_f32, dim3(B), dim3(1024), 0, ST, x, T, C, div, lens, lens_div);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_seq_lengths_i32(const int32_t* x, int B, int L, int pad, int div, int32_t* lens, int32_t* lens_div, void* stream) {
    E2T_CHECK_ARG(x && lens && B > 0 && L > 0 && div > 0);
    hipLaunchKernelGGL(k_seq_lengths_i32, dim3((B + 63) / 64), dim3(64), 0, ST, x, B, L, pad, div, lens, lens_div);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_sum_i32(const int32_t* x, int n, int32_t* out, void* stream) {
    E2T_CHECK_ARG(x && out && n > 0);
    hipLaunchKernelGGL(k_sum_i32, dim3(1), dim3(256), 0, ST, x, n, out);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_sum2_f32(const float* x0, const float* x1, int n, const int32_t* count, float scale0, float scale1, float* out0, float* out1,
                            void* stream) {
    E2T_CHECK_ARG(x0 && x1 && out0 && out1 && n > 0);
    hipLaunchKernelGGL(k_sum_f32, dim3(2), dim3(256), 0, ST, x0, n, count, scale0, out0, x1, scale1, out1);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_sum_f32(const float* x, int n, const int32_t* count, float scale, float* out, void* stream) {
    E2T_CHECK_ARG(x && out && n > 0);
    hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, ST, x, n, count, scale, out);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_conv_pack_grouped(const float* x, const int32_t* lens, int B, int T, int C, int N, int G, void* A, int lda, void* stream) {
    E2T_CHECK_ARG(x && lens && A && B > 0 && T > 0 && C > 0 && N > 0 && G > 0);
    E2T_CHECK_ARG(lda % 8 == 0 && lda >= N * C);
    const int S = ((T + N * G - 1) / (N * G)) * G;          // steps of this layer: a whole number of groups
    hipLaunchKernelGGL(k_conv_pack, dim3(S * B), dim3(256), 0, ST, x, lens, B, T, C, N, S, (bf16_t*)A, lda, G);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_conv_pack(const float* x, const int32_t* lens, int B, int T, int C, int N, void* A, int lda, void* stream) {
    return e2t_conv_pack_grouped(x, lens, B, T, C, N, 1, A, lda, stream);
}
extern "C" int e2t_conv_unpack_grad_grouped(const float* dA, int ldda, const int32_t* lens, int B, int T, int C, int N, int G, float* dx, void* stream) {
    E2T_CHECK_ARG(dA && lens && dx && B > 0 && T > 0 && C > 0 && N > 0 && G > 0 && ldda >= N * C);
    const int S = ((T + N * G - 1) / (N * G)) * G;
    hipLaunchKernelGGL(k_conv_unpack_grad, dim3(B * T), dim3(256), 0, ST, dA, ldda, lens, B, T, C, N, S, dx, G);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_conv_unpack_grad(const float* dA, int ldda, const int32_t* lens, int B, int T, int C, int N, float* dx, void* stream) {
    return e2t_conv_unpack_grad_grouped(dA, ldda, lens, B, T, C, N, 1, dx, stream);
}
extern "C" int e2t_gather_rev_decim_f32(const float* a, const int32_t* tlens, int B, int T, int K, int N, float* out, void* stream) {
    E2T_CHECK_ARG(a && tlens && out && B > 0 && T > 0 && K > 0 && N > 0);
    const int S = (T + N - 1) / N;
    const size_t n = (size_t)S * B * K;
    hipLaunchKernelGGL(k_gather_rev_decim_f32, dim3((n + 255) / 256), dim3(256), 0, ST, a, tlens, B, T, K, N, S, out);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_gather_rev_decim_i32(const int32_t* a, const int32_t* tlens, int B, int T, int N, int32_t* out, void* stream) {
    E2T_CHECK_ARG(a && tlens && out && B > 0 && T > 0 && N > 0);
    const int S = (T + N - 1) / N;
    const size_t n = (size_t)S * B;
    hipLaunchKernelGGL(k_gather_rev_decim_i32, dim3((n + 255) / 256), dim3(256), 0, ST, a, tlens, B, T, N, S, out);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_decoder_tokens(const int32_t* y, int B, int L, int eos, int32_t* U, int32_t* Tg, void* stream) {
    E2T_CHECK_ARG(y && U && Tg && B > 0 && L > 0);
    hipLaunchKernelGGL(k_decoder_tokens, dim3((B * L + 255) / 256), dim3(256), 0, ST, y, B, L, eos, U, Tg);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_transpose_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, void* stream) {
    E2T_CHECK_ARG(in && out && R > 0 && C > 0 && ld_in >= C && ld_out >= R);
    hipLaunchKernelGGL(k_transpose_bf16, dim3((ld_out + 63) / 64, (C + 63) / 64), dim3(256), 0, ST, (const bf16_t*)in, ld_in, R, C,
                       (bf16_t*)out, ld_out);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_cast_pack(const float* src, long row_stride, long col_stride, int R, int C, void* dst, int ld_dst, void* stream) {
    E2T_CHECK_ARG(src && dst && R > 0 && C > 0 && ld_dst >= C);
    hipLaunchKernelGGL(k_cast_pack, dim3((C + 255) / 256, R), dim3(256), 0, ST, src, row_stride, col_stride, R, C, (bf16_t*)dst, ld_dst);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_pack_frag(const float* src, long n_stride, long k_stride, int Nn, int Kk, void* dst, void* stream) {
    E2T_CHECK_ARG(src && dst && Nn > 0 && Kk > 0);
    const int NT = (Nn + 15) / 16, KB = (Kk + 31) / 32;
    hipLaunchKernelGGL(k_pack_frag, dim3(KB, NT), dim3(64), 0, ST, src, n_stride, k_stride, Nn, Kk, KB, (bf16_t*)dst);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_pack_batch(const e2t_pack_desc* descs_dev, int ndesc, int total_blocks, const float* base, void* stream) {
    E2T_CHECK_ARG(descs_dev && base && ndesc > 0 && total_blocks > 0);
    hipLaunchKernelGGL(k_pack_batch, dim3(total_blocks), dim3(256), 0, ST, descs_dev, ndesc, base);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
static DropCfg mk_drop(const e2t_dropout* d) {
    DropCfg c{}; if (d) { c.rate = d->rate; c.seed = d->seed; c.step = d->step; c.stream = d->stream; } return c;
}
extern "C" int e2t_embed_fwd(const void* emb, int ld_emb, const int32_t* tok, int row0, int M, int E, void* out, int ld_out,
                             const e2t_dropout* drop, void* stream) {
    E2T_CHECK_ARG(emb && tok && out && M > 0 && E > 0 && row0 >= 0);
    const size_t n = (size_t)M * E;
    hipLaunchKernelGGL(k_embed_fwd, dim3((n + 255) / 256), dim3(256), 0, ST, (const bf16_t*)emb, ld_emb, tok, M, E, (bf16_t*)out, ld_out,
                       mk_drop(drop), row0);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_embed_bwd(const float* de, int ld_de, const int32_t* tok, int M, int E, float* demb, int ld_demb,
                             const e2t_dropout* drop, void* stream) {
    E2T_CHECK_ARG(de && tok && demb && M > 0 && E > 0);
    const size_t n = (size_t)M * E;
    hipLaunchKernelGGL(k_embed_bwd, dim3((n + 255) / 256), dim3(256), 0, ST, de, ld_de, tok, M, E, demb, ld_demb, mk_drop(drop));
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_softmax_ce(const float* logits, int ldl, int M, int V, const int32_t* tgt, const int32_t* lens, int rows_per_step,
                              const int32_t* ntok, float weight, float* rowloss, int32_t* pred, float* correct, void* dlogits,
                              int lddl, void* stream) {
    E2T_CHECK_ARG(logits && M > 0 && V > 0 && ldl >= V);
    E2T_CHECK_ARG(!dlogits || (tgt && lddl >= V));
    hipLaunchKernelGGL(k_softmax_ce, dim3((M + 3) / 4), dim3(256), 0, ST, logits, ldl, M, V, tgt, lens,
                       rows_per_step > 0 ? rows_per_step : 1, ntok, weight, rowloss, pred, correct, (bf16_t*)dlogits, lddl);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_mse(const float* P, int ldp, const float* At, int M, int K, const int32_t* lens, int rows_per_step,
                       const int32_t* nval, float weight, float* rowloss, void* dP, int lddp, void* stream) {
    E2T_CHECK_ARG(P && At && lens && nval && rowloss && M > 0 && K > 0 && rows_per_step > 0);
    hipLaunchKernelGGL(k_mse, dim3((M + 255) / 256), dim3(256), 0, ST, P, ldp, At, M, K, lens, rows_per_step, nval, weight, rowloss,
                       (bf16_t*)dP, lddp);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_final_state(const void* Yext, int ldy, const float* Cs, const int32_t* lens, int B, int H, void* h0, int ldh0,
                               float* c0, void* stream) {
    E2T_CHECK_ARG(Yext && Cs && lens && h0 && c0 && B > 0 && H > 0 && ldh0 >= 2 * H);
    const int n = B * 2 * H;
    hipLaunchKernelGGL(k_final_state, dim3((n + 255) / 256), dim3(256), 0, ST, (const bf16_t*)Yext, ldy, Cs, lens, B, H, (H + 7) / 8 * 8,
                       (bf16_t*)h0, ldh0, c0);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_greedy_update(const int32_t* pred, int B, int l, int Lmax, int eos, int pad, int32_t* done, int32_t* out,
                                 int32_t* next_tok, void* stream) {
    E2T_CHECK_ARG(pred && done && out && B > 0 && l >= 0 && l < Lmax);
    hipLaunchKernelGGL(k_greedy_update, dim3((B + 255) / 256), dim3(256), 0, ST, pred, B, l, Lmax, eos, pad, done, out, next_tok);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_greedy_step(const float* logits, int ldl, int B, int V, int l, int Lmax, int eos, int pad, int32_t* done, int32_t* out,
                               int32_t* next_tok, void* stream) {
    E2T_CHECK_ARG(logits && done && out && B > 0 && V > 0 && ldl >= V && l >= 0 && l < Lmax);
    hipLaunchKernelGGL(k_greedy_step, dim3((B + 3) / 4), dim3(256), 0, ST, logits, ldl, B, V, l, Lmax, eos, pad, done, out, next_tok);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_decode_init(int32_t* done, int32_t* hyp, int32_t* tok0, int32_t* dlens, int B, int Lmax, int eos, int pad, void* stream) {
    E2T_CHECK_ARG(done && hyp && tok0 && dlens && B > 0 && Lmax > 0);
    hipLaunchKernelGGL(k_decode_init, dim3((B * Lmax + 255) / 256 > 64 ? 64 : (B * Lmax + 255) / 256), dim3(256), 0, ST, done, hyp, tok0, dlens, B, Lmax, eos, pad);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_greedy_head_small(const void* h, int ldh, const void* WT, int ldw, const float* bias, int B, int V, int K, int l, int Lmax,
                                     int eos, int pad, int32_t* done, int32_t* out, int32_t* next_tok, const void* table, size_t row_words,
                                     void* gx_next, uint32_t* scratch, void* stream) {
    E2T_CHECK_ARG(h && WT && done && out && scratch && B > 0 && B <= E2T_HEAD_MAXB && V > 0 && K > 0 && l >= 0 && l < Lmax);
    E2T_CHECK_ARG(ldh >= K && ldw >= ((K + 7) & ~7) && (ldw & 7) == 0 && (((uintptr_t)WT) & 15) == 0 && (!table || gx_next));
    const int K8 = (K + 7) & ~7;
    const size_t lds = (size_t)B * K8 * 2 + 4 * B * 2 * sizeof(float);
    E2T_CHECK_ARG(lds <= 64 * 1024);
    int grid = (V + 31) / 32; if (grid > 64) grid = 64;          // 8 rows per wave pass, 4 waves: one pass at V = 1806 on 57 workgroups
    hipLaunchKernelGGL(k_greedy_head_small, dim3(grid), dim3(256), lds, ST, (const bf16_t*)h, ldh, (const bf16_t*)WT, ldw, bias, B, V, K, l, Lmax,
                       eos, pad, done, out, next_tok, (const unsigned*)table, row_words, (unsigned*)gx_next, scratch);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_beam_step(const float* logits, int ldl, int B, int W, int V, float temperature, int l, int Lmax, int eos, int pad,
                             const float* score_in, const int32_t* done_in, const int32_t* hyp_in, float* score_out, int32_t* done_out,
                             int32_t* hyp_out, int32_t* rowmap, int32_t* next_tok, void* stream) {
    E2T_CHECK_ARG(logits && score_in && done_in && hyp_in && score_out && done_out && hyp_out && rowmap);
    E2T_CHECK_ARG(B > 0 && W >= 1 && W <= E2T_BEAM_MAX && V > 0 && ldl >= V && temperature > 0.f && l >= 0 && l < Lmax);
    hipLaunchKernelGGL(k_beam_step, dim3(B), dim3(256), 0, ST, logits, ldl, W, V, 1.0f / temperature, l, Lmax, eos, pad, score_in, done_in,
                       hyp_in, score_out, done_out, hyp_out, rowmap, next_tok);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_beam_reorder(void* Yblk, int ldy, float* Cs_step, int rows, int H, const int32_t* rowmap, void* tmp_h, float* tmp_c,
                                void* stream) {
    E2T_CHECK_ARG(Yblk && Cs_step && rowmap && tmp_h && tmp_c && rows > 0 && H > 0 && ldy >= H);
    const int n = rows * H, H8 = (H + 7) / 8 * 8;
    for (int mode = 0; mode < 2; ++mode)
        hipLaunchKernelGGL(k_beam_reorder, dim3((n + 255) / 256), dim3(256), 0, ST, (bf16_t*)Yblk, ldy, Cs_step, rows, H, H8, rowmap,
                           (bf16_t*)tmp_h, tmp_c, mode);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_inc_step(int32_t* step, const int32_t* skip_if_nonzero, void* stream) {
    E2T_CHECK_ARG(step);
    hipLaunchKernelGGL(k_inc_step, dim3(1), dim3(64), 0, ST, step, skip_if_nonzero);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_adam_pack_batch(const e2t_tile_desc* descs_dev, int ndesc, int total_blocks, float* p, const float* g, float* m, float* v,
                                   float* ema, const int32_t* step, const e2t_adam_hyper* h, void* stream) {
    E2T_CHECK_ARG(descs_dev && p && ndesc > 0 && total_blocks > 0);
    E2T_CHECK_ARG(!h || (g && m && v && ema && step));
    E2T_CHECK_ARG(((uintptr_t)p & 15) == 0 && (!h || ((((uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) == 0)));
    AdamPackArgs a{};
    if (h) {
        a.step = step; a.skip = h->skip_if_nonzero; a.lr = h->lr; a.b1 = h->beta1; a.b2 = h->beta2; a.eps = h->eps; a.decay = h->ema_decay;
        a.gscale = h->grad_scale; a.step_offset = h->step_offset; a.update = 1;
    }
    hipLaunchKernelGGL(k_adam_pack, dim3(total_blocks), dim3(256), 0, ST, descs_dev, ndesc, p, g, m, v, ema, a);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
extern "C" int e2t_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, size_t n, const int32_t* step,
                                 const e2t_adam_hyper* h, void* stream) {
    E2T_CHECK_ARG(p && g && m && v && ema && step && h);
    if (n == 0) return E2T_OK;
    const uintptr_t a = (uintptr_t)p & 15;
    const int vec = (((uintptr_t)g & 15) == a && ((uintptr_t)m & 15) == a && ((uintptr_t)v & 15) == a && ((uintptr_t)ema & 15) == a && (a & 3) == 0) ? 1 : 0;
    // Short workgroups (one 16-B group per thread, at most four): 2048 workgroups looping over the whole range held every slot of
    // the chip for the kernel's whole duration, and a small kernel of the other branch -- the bottom stage's grouped reduction,
    // 25 us of work -- then sat behind them until the update was over (cfg4: 689 us beside the 812-us early update).
    size_t blocks = ((vec ? n / 4 + 8 : n) + 255) / 256; if (blocks > (1u << 18)) blocks = (blocks + 3) / 4;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    hipLaunchKernelGGL(k_adam_ema, dim3((unsigned)blocks), dim3(256), 0, ST, p, g, m, v, ema, n, step, h->lr, h->beta1, h->beta2,
                       h->eps, h->ema_decay, h->grad_scale, h->step_offset, h->skip_if_nonzero, vec);
    E2T_LAUNCH_CHECK(); return E2T_OK;
}
