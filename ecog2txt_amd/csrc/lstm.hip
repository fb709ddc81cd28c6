// Fused LSTM recurrence kernels (reference rows a7, a9:
// SequenceNetwork._encode_sequences and the decoder RNN, called from
// ecog2txt/trainers.py:318 via net.fit; 4-gate packing trainers.py:527-529).
//
// One launch = one time step of ONE layer for BOTH directions; the host side
// (e2t_lstm_seq_fwd / _bwd) issues the S launches back to back on one stream so
// the whole sequence replays from a hipGraph.  Layers of a bidirectional stack
// are strictly serial (layer l+1 at t=0 needs layer l's backward direction at
// t=0, produced last), so the only concurrency is {fwd,bwd} x batch x units,
// which is exactly the grid: (unit tiles of 16, row tiles of 16, directions).
//
// Forward step, per wave (64 lanes): a 16-row x 16-unit patch, four MFMA
// 16x16x32 accumulators (one per gate i,j,f,o) so every lane ends up holding
// all four pre-activations of its (row, unit) cells: the gate nonlinearities,
// cell update, dropout and the bf16 store are lane-local.  A operand = h_{t-1}
// rows gathered straight from the layer's bf16 output array (per-row time
// index => tf.reverse_sequence and variable lengths cost nothing); B operand =
// W_h pre-packed in MFMA fragment order so each load is one coalesced 1-KiB
// wave transaction served from the XCD-local L2.
//
// Backward step: dh_rec = dG_{t+1} . W_h^T (K = 4H) for a 16x16 patch, then
// the full LSTM cell backward for those cells (dG_t in bf16 for the later
// weight-gradient GEMMs, dc carried in fp32).  dW_h / dW_x are NOT accumulated
// per step: they are two large GEMMs over all steps afterwards (better MFMA
// utilisation).
#include "common.h"
#include "ecog2txt_hip.h"

struct LstmFwdArgs {
    const float* Gx;        // [S*B][ndir*H*4]  fp32, (dir, unit, gate) interleaved, bias included
    const bf16_t* WhF;      // [ndir][4 gates][UT][KB][64 lanes][8]  fragment-packed W_h
    bf16_t* Yext;           // [(S+2)*B][ldy]   time block tau = t+1; blocks 0 / S+1 = initial h / zero
    bf16_t* Ydrop;          // [S*B][ldy] or null
    float* Cs;              // [S*B][ndir*H]
    float* Gs;              // [S*B][ndir*H][4] post-activation gates (i,j,f,o)
    const int* lens;        // [B]
    const float* c0;        // [B][ndir*H] or null
    int S, B, H, H8, ndir, ldy, UT, KB, step;
    float forget_bias;
    DropCfg drop;
};

__global__ __launch_bounds__(64) void k_lstm_step_fwd(LstmFwdArgs p) {
    const int lane = threadIdx.x;
    const int ut = blockIdx.x, rt = blockIdx.y, dir = blockIdx.z;
    const int s = p.step, B = p.B, H = p.H;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- A operand: h_{t-1} rows (16 rows x K), gathered per row -------------
    const int ab = rt * 16 + frow;                    // batch row this lane loads for
    const bf16_t* arow = nullptr;
    if (ab < B) {
        const int len = p.lens[ab];
        // The backward direction starts from a zero state at its first step; the ext
        // block it would read (position t = len) is only re-zeroed for THIS batch at a
        // later launch, so it must not be read here.  The forward direction's first
        // step reads block 0 (initial state: zeros for the encoder, h0 for the decoder).
        if (s < len && !(dir == 1 && s == 0)) {
            const int t = dir ? (len - 1 - s) : s;
            const int tau_prev = dir ? (t + 2) : t;   // ext index of h_{t-1} in processing order
            arow = p.Yext + ((size_t)tau_prev * B + ab) * p.ldy + dir * p.H8;
        }
    }
    const bf16x8* wf = (const bf16x8*)p.WhF + ((size_t)(dir * 4) * p.UT + ut) * p.KB * 64 + lane;
    const size_t gate_stride = (size_t)p.UT * p.KB * 64;

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 4
    for (int kb = 0; kb < p.KB; ++kb) {
        const int k = kb * 32 + fq * 8;
        bf16x8 a = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (arow && k < p.H8) a = *(const bf16x8*)(arow + k);
        bf16x8 b0 = wf[(size_t)kb * 64];
        bf16x8 b1 = wf[gate_stride + (size_t)kb * 64];
        bf16x8 b2 = wf[2 * gate_stride + (size_t)kb * 64];
        bf16x8 b3 = wf[3 * gate_stride + (size_t)kb * 64];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b3, acc[3], 0, 0, 0);
    }

    // ---- lane-local cell update: lane owns unit u for rows fq*4 + r ----------
    const int u = ut * 16 + frow;
    if (u >= H) return;
    const int NH = p.ndir * H;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = rt * 16 + fq * 4 + r;
        if (b >= B) continue;
        const int len = p.lens[b];
        if (s < len) {
            const int t = dir ? (len - 1 - s) : s;
            const size_t m = (size_t)t * B + b;
            const float4 gx = *(const float4*)(p.Gx + (m * NH + dir * H + u) * 4);
            const float gi = sigmoidf_(acc[0][r] + gx.x);
            const float gj = tanhf_(acc[1][r] + gx.y);
            const float gf = sigmoidf_(acc[2][r] + gx.z + p.forget_bias);
            const float go = sigmoidf_(acc[3][r] + gx.w);
            float cprev;
            if (s > 0) {
                const int tp = dir ? (t + 1) : (t - 1);
                cprev = p.Cs[((size_t)tp * B + b) * NH + dir * H + u];
            } else {
                cprev = p.c0 ? p.c0[(size_t)b * NH + dir * H + u] : 0.f;
            }
            const float c = gf * cprev + gi * gj;
            const float h = go * tanhf_(c);
            p.Cs[m * NH + dir * H + u] = c;
            *(float4*)(p.Gs + (m * NH + dir * H + u) * 4) = make_float4(gi, gj, gf, go);
            p.Yext[((size_t)(t + 1) * B + b) * p.ldy + dir * p.H8 + u] = f2bf(h);
            if (p.Ydrop) {
                const float sc = drop_scale(p.drop, (unsigned long long)(m * NH + dir * H + u));
                p.Ydrop[m * p.ldy + dir * p.H8 + u] = f2bf(h * sc);
            }
        } else if (s < p.S) {
            // padded position s of this utterance: emit zeros (dynamic_rnn semantics)
            const size_t m = (size_t)s * B + b;
            p.Yext[((size_t)(s + 1) * B + b) * p.ldy + dir * p.H8 + u] = 0;
            if (p.Ydrop) p.Ydrop[m * p.ldy + dir * p.H8 + u] = 0;
        }
    }
}

struct LstmBwdArgs {
    const bf16_t* WhB;      // [ndir][UT][KB4][64][8]  fragment-packed W_h^T operand (K = 4H gate columns)
    bf16_t* dG;             // [S*B][lddg]  (dir, unit, gate) interleaved, bf16
    const float* dY;        // [S*B][lddy] gradient wrt the (dropped) layer output, or null
    const float* Gs; const float* Cs;
    const int* lens;
    const float* c0;        // [B][ndir*H] or null
    const float* dh_final;  // [B][ndir*H] or null: gradient wrt final state h
    const float* dc_final;  // [B][ndir*H] or null
    float* dc_carry;        // [B][ndir*H] workspace (in/out)
    float* dh0;             // [B][ndir*H] out, written when step == -1 (else untouched)
    float* dc0;             // [B][ndir*H] out, written when step == -1
    int S, B, H, H8, ndir, lddg, lddy, UT, KB4, step;
    DropCfg drop;
};

__global__ __launch_bounds__(64) void k_lstm_step_bwd(LstmBwdArgs p) {
    const int lane = threadIdx.x;
    const int ut = blockIdx.x, rt = blockIdx.y, dir = blockIdx.z;
    const int s = p.step, B = p.B, H = p.H;
    const int frow = lane & 15, fq = lane >> 4;
    const int K4 = 4 * H;

    // ---- A operand: dG of the step processed just before in the sweep (s+1) --
    const int ab = rt * 16 + frow;
    const bf16_t* arow = nullptr;
    if (ab < B) {
        const int len = p.lens[ab];
        if (s + 1 < len) {
            const int tn = dir ? (len - 2 - s) : (s + 1);
            arow = p.dG + ((size_t)tn * B + ab) * p.lddg + (size_t)dir * K4;
        }
    }
    const bf16x8* wf = (const bf16x8*)p.WhB + ((size_t)dir * p.UT + ut) * p.KB4 * 64 + lane;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kb = 0; kb < p.KB4; ++kb) {
        const int k = kb * 32 + fq * 8;
        bf16x8 a = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (arow && k < K4) a = *(const bf16x8*)(arow + k);
        bf16x8 b = wf[(size_t)kb * 64];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }

    const int u = ut * 16 + frow;
    if (u >= H) return;
    const int NH = p.ndir * H;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = rt * 16 + fq * 4 + r;
        if (b >= B) continue;
        const int len = p.lens[b];
        const size_t su = (size_t)b * NH + dir * H + u;          // state index [B][ndir*H]
        if (s < 0) {
            // pseudo-step -1: gradient into the initial state (decoder <- encoder seam)
            if (len > 0) {
                p.dh0[su] = acc[r];
                p.dc0[su] = p.dc_carry[su];
            } else {
                p.dh0[su] = p.dh_final ? p.dh_final[su] : 0.f;
                p.dc0[su] = p.dc_final ? p.dc_final[su] : 0.f;
            }
            continue;
        }
        if (s < len) {
            const int t = dir ? (len - 1 - s) : s;
            const size_t m = (size_t)t * B + b;
            const size_t e = m * NH + dir * H + u;
            float dh = acc[r];
            if (p.dY) {
                float g = p.dY[m * p.lddy + dir * p.H8 + u];
                g *= drop_scale(p.drop, (unsigned long long)e);
                dh += g;
            }
            const bool last = (s == len - 1);
            if (last && p.dh_final) dh += p.dh_final[su];
            float dc_in = last ? (p.dc_final ? p.dc_final[su] : 0.f) : p.dc_carry[su];
            const float4 g4 = *(const float4*)(p.Gs + e * 4);
            const float c_t = p.Cs[e];
            float cprev;
            if (s > 0) {
                const int tp = dir ? (t + 1) : (t - 1);
                cprev = p.Cs[((size_t)tp * B + b) * NH + dir * H + u];
            } else {
                cprev = p.c0 ? p.c0[su] : 0.f;
            }
            const float tc = tanhf_(c_t);
            const float dct = dc_in + dh * g4.w * (1.f - tc * tc);
            const float d_o = dh * tc * g4.w * (1.f - g4.w);
            const float d_i = dct * g4.y * g4.x * (1.f - g4.x);
            const float d_j = dct * g4.x * (1.f - g4.y * g4.y);
            const float d_f = dct * cprev * g4.z * (1.f - g4.z);
            ushort4 o;
            o.x = f2bf(d_i); o.y = f2bf(d_j); o.z = f2bf(d_f); o.w = f2bf(d_o);
            *(ushort4*)(p.dG + m * p.lddg + (size_t)dir * K4 + u * 4) = o;
            p.dc_carry[su] = dct * g4.z;
        } else if (s < p.S) {
            const size_t m = (size_t)s * B + b;
            *(ushort4*)(p.dG + m * p.lddg + (size_t)dir * K4 + u * 4) = make_ushort4(0, 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int e2t_lstm_seq_fwd(const e2t_lstm_desc* d, const float* Gx, const void* WhF, void* Yext,
                                void* Ydrop, float* Cs, float* Gs, const int32_t* lens, const float* c0,
                                int step_begin, int step_end, void* stream) {
    E2T_CHECK_ARG(d && Gx && WhF && Yext && Cs && Gs && lens);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && d->H > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(d->H % 2 == 0 && d->ldy % 8 == 0 && d->ldy >= d->ndir * ((d->H + 7) / 8) * 8);
    E2T_CHECK_ARG(0 <= step_begin && step_begin <= step_end && step_end <= d->S);
    LstmFwdArgs p{};
    p.Gx = Gx; p.WhF = (const bf16_t*)WhF; p.Yext = (bf16_t*)Yext; p.Ydrop = (bf16_t*)Ydrop;
    p.Cs = Cs; p.Gs = Gs; p.lens = lens; p.c0 = c0;
    p.S = d->S; p.B = d->B; p.H = d->H; p.H8 = (d->H + 7) / 8 * 8; p.ndir = d->ndir; p.ldy = d->ldy;
    p.UT = (d->H + 15) / 16; p.KB = (p.H8 + 31) / 32;
    p.forget_bias = d->forget_bias;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    dim3 grid(p.UT, (d->B + 15) / 16, d->ndir);
    for (int s = step_begin; s < step_end; ++s) {
        p.step = s;
        hipLaunchKernelGGL(k_lstm_step_fwd, grid, dim3(64), 0, (hipStream_t)stream, p);
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}

extern "C" int e2t_lstm_seq_bwd(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY,
                                int lddy, const float* Gs, const float* Cs, const int32_t* lens, const float* c0,
                                const float* dh_final, const float* dc_final, float* dc_carry, float* dh0,
                                float* dc0, void* stream) {
    E2T_CHECK_ARG(d && WhB && dG && Gs && Cs && lens && dc_carry);
    E2T_CHECK_ARG(d->S > 0 && d->B > 0 && d->H > 0 && (d->ndir == 1 || d->ndir == 2));
    E2T_CHECK_ARG(d->H % 2 == 0 && lddg % 8 == 0 && lddg >= d->ndir * 4 * d->H);
    E2T_CHECK_ARG((dh0 == nullptr) == (dc0 == nullptr));
    LstmBwdArgs p{};
    p.WhB = (const bf16_t*)WhB; p.dG = (bf16_t*)dG; p.dY = dY; p.Gs = Gs; p.Cs = Cs; p.lens = lens; p.c0 = c0;
    p.dh_final = dh_final; p.dc_final = dc_final; p.dc_carry = dc_carry; p.dh0 = dh0; p.dc0 = dc0;
    p.S = d->S; p.B = d->B; p.H = d->H; p.H8 = (d->H + 7) / 8 * 8; p.ndir = d->ndir;
    p.lddg = lddg; p.lddy = lddy;
    p.UT = (d->H + 15) / 16; p.KB4 = (4 * d->H + 31) / 32;
    p.drop.rate = d->drop_rate; p.drop.seed = d->drop_seed; p.drop.step = d->drop_step; p.drop.stream = d->drop_stream;
    dim3 grid(p.UT, (d->B + 15) / 16, d->ndir);
    for (int s = d->S - 1; s >= (dh0 ? -1 : 0); --s) {
        p.step = s;
        hipLaunchKernelGGL(k_lstm_step_bwd, grid, dim3(64), 0, (hipStream_t)stream, p);
    }
    E2T_LAUNCH_CHECK();
    return E2T_OK;
}
