// CPU BASELINE / TEST INFRASTRUCTURE ONLY -- never linked, imported or called by the product path (ecog2txt_amd/).
//
// "the build's own C++17/OpenMP fp32 CPU implementation of the identical train step (same spec as the
// oracle), timed in the same run on all host cores of the GPU box".  The reference's own CPU path (TF1.x + the un-vendored
// `machine_learning` package) cannot run here; this file restates oracle/seq2seq.py (which cites the reference lines it follows:
// trainers.py:786-823 forward fragment, :444-554 variable grammar, :527-529 gate packing) in fp32 C++ with OpenMP:
// reverse + strided temporal convolution (one layer) -> stacked bidirectional LSTM -> auxiliary FF head (squared error) ->
// LSTM decoder (teacher forced) -> projection -> masked cross entropy; manual reverse mode; Adam + EMA.
// tests/test_cpu_step.py pins it against the NumPy oracle (losses and every gradient, dropout off); bench.py's `cpu_baseline`
// times it (dropout on: a stateless hash mask, NOT the Philox stream of the device -- timing only).
//
// Build (oracle/build_cpu_step.sh): g++ -O3 -march=native -fopenmp -shared -fPIC oracle/cpu_step.cpp -o oracle/_cpu/libe2t_cpu_step.so
#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

namespace {
typedef float v16 __attribute__((vector_size(64)));
constexpr int MR = 6, NR = 32;

struct Buf {                                   // 64-byte aligned float array
    float* p = nullptr; size_t n = 0;
    void resize(size_t m) { if (m > n) { free(p); p = (float*)aligned_alloc(64, (m * 4 + 63) / 64 * 64); n = m; } }
    ~Buf() { free(p); }
};

// ---- GEMM: C[M x N] = (acc ? C : 0) + op(A)[M x K] . op(B)[K x N] (+ bias[N] when !acc), row-major, GotoBLAS-style packing ----
struct PackedB { Buf b; int K = 0, N = 0; };   // [ceil(N/NR)][K][NR]
static void pack_b(PackedB& P, bool tb, int K, int N, const float* B, int ldb) {
    P.K = K; P.N = N;
    const int np = (N + NR - 1) / NR;
    P.b.resize((size_t)np * K * NR);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < np; ++j) {
        float* d = P.b.p + (size_t)j * K * NR;
        const int n0 = j * NR, nv = std::min(NR, N - n0);
        for (int k = 0; k < K; ++k) {
            if (!tb) { const float* s = B + (size_t)k * ldb + n0; for (int c = 0; c < nv; ++c) d[k * NR + c] = s[c]; }
            else for (int c = 0; c < nv; ++c) d[k * NR + c] = B[(size_t)(n0 + c) * ldb + k];
            for (int c = nv; c < NR; ++c) d[k * NR + c] = 0.f;
        }
    }
}
static inline void ukernel(int kc, const float* __restrict a, const float* __restrict b, float* __restrict c, int ldc, int mr, int nr,
                           bool acc, const float* bias) {
    v16 r[MR][2];
    for (int i = 0; i < MR; ++i) { r[i][0] = v16{}; r[i][1] = v16{}; }
    for (int k = 0; k < kc; ++k) {
        const v16 b0 = *(const v16*)(b + (size_t)k * NR), b1 = *(const v16*)(b + (size_t)k * NR + 16);
        const float* ak = a + (size_t)k * MR;
#pragma GCC unroll 6
        for (int i = 0; i < MR; ++i) { r[i][0] += ak[i] * b0; r[i][1] += ak[i] * b1; }
    }
    for (int i = 0; i < mr; ++i) {
        float* ci = c + (size_t)i * ldc;
        float t[NR];
        memcpy(t, &r[i][0], 64); memcpy(t + 16, &r[i][1], 64);
        if (acc) for (int j = 0; j < nr; ++j) ci[j] += t[j];
        else if (bias) for (int j = 0; j < nr; ++j) ci[j] = t[j] + bias[j];
        else for (int j = 0; j < nr; ++j) ci[j] = t[j];
    }
}
static void gemm_pb(bool ta, int M, const float* A, int lda, const PackedB& P, float* C, int ldc, bool acc, const float* bias) {
    const int N = P.N, K = P.K;
    const int MB = M >= 2048 ? 96 : 48, NB = (M >= 2048 && N >= 1024) ? 128 : 64, KC = 384;
    const int nmb = (M + MB - 1) / MB, nnb = (N + NB - 1) / NB;
#pragma omp parallel
    {
        static thread_local Buf ab;
        ab.resize((size_t)MB * KC);
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int im = 0; im < nmb; ++im)
            for (int in = 0; in < nnb; ++in) {
                const int m0 = im * MB, mv = std::min(MB, M - m0), n0 = in * NB, nv = std::min(NB, N - n0);
                for (int k0 = 0; k0 < K; k0 += KC) {
                    const int kc = std::min(KC, K - k0);
                    // pack the A block: [ceil(mv/MR)][kc][MR]
                    for (int i0 = 0; i0 < mv; i0 += MR) {
                        float* d = ab.p + (size_t)(i0 / MR) * kc * MR;
                        const int iv = std::min(MR, mv - i0);
                        if (!ta) {
                            for (int i = 0; i < iv; ++i) { const float* s = A + (size_t)(m0 + i0 + i) * lda + k0; for (int k = 0; k < kc; ++k) d[k * MR + i] = s[k]; }
                        } else {
                            for (int k = 0; k < kc; ++k) { const float* s = A + (size_t)(k0 + k) * lda + m0 + i0; for (int i = 0; i < iv; ++i) d[k * MR + i] = s[i]; }
                        }
                        for (int i = iv; i < MR; ++i) for (int k = 0; k < kc; ++k) d[k * MR + i] = 0.f;
                    }
                    for (int j0 = 0; j0 < nv; j0 += NR) {
                        const float* bp = P.b.p + ((size_t)((n0 + j0) / NR) * K + k0) * NR;
                        for (int i0 = 0; i0 < mv; i0 += MR)
                            ukernel(kc, ab.p + (size_t)(i0 / MR) * kc * MR, bp, C + (size_t)(m0 + i0) * ldc + n0 + j0, ldc,
                                    std::min(MR, mv - i0), std::min(NR, nv - j0), acc || k0 > 0, (k0 == 0 && bias) ? bias + n0 + j0 : nullptr);
                    }
                }
            }
    }
}
static void gemm(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, bool acc,
                 const float* bias = nullptr) {
    static PackedB P;                           // (calls are serial: one scratch image)
    pack_b(P, tb, K, N, B, ldb);
    gemm_pb(ta, M, A, lda, P, C, ldc, acc, bias);
}
static void colsum(int M, int N, const float* A, int lda, float* out) {
#pragma omp parallel for schedule(static)
    for (int j0 = 0; j0 < N; j0 += 16) {
        const int nv = std::min(16, N - j0);
        double s[16] = {0};
        for (int i = 0; i < M; ++i) for (int j = 0; j < nv; ++j) s[j] += A[(size_t)i * lda + j0 + j];
        for (int j = 0; j < nv; ++j) out[j0 + j] = (float)s[j];
    }
}
static inline float sigm(float x) { return 1.f / (1.f + expf(-x)); }
// stateless dropout scale (timing only; the device uses Philox): keep / (1 - rate) or 0
static inline float dscale(uint64_t seed, uint64_t stream, uint64_t idx, float rate) {
    if (rate <= 0.f) return 1.f;
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + stream * 0xBF58476D1CE4E5B9ull + idx;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return ((z >> 40) * (1.0f / 16777216.0f)) >= rate ? 1.f / (1.f - rate) : 0.f;
}

struct Spec { int C, N, F, nl, H[8], E, Hd, V, aux_layer, naux, auxh[4], aux_dim, B, T, L, S; float ff_drop, rnn_drop, forget_bias, aux_scale, dec_scale; };
struct Seg { size_t off; int r, c; };

struct Dir {                                   // saves of one LSTM direction
    Buf Gx, G, Cs, Hp, Cp, dG; std::vector<float> hT, cT; PackedB Kh, KhT;
};
struct Net {
    Spec s;
    size_t np = 0;
    std::vector<float> p, g, m, v, ema;
    Seg conv_w, conv_b, emb, dec_k, dec_b;
    std::vector<Seg> enc_k, enc_b, aux_w, aux_b, proj_w, proj_b;     // enc: index l*2+d
    std::vector<int> aux_sizes, proj_sizes;
    int step = 0;
    uint64_t seed = 1;
    // activations
    Buf A, E, dE;
    std::vector<Buf> Y, Yd, dY;                // per encoder layer [S*B][2H]
    std::vector<Dir> dirs; Dir dec;
    std::vector<Buf> aux_act, proj_act;
    Buf e, de, Ydec, Hdrop, dH, logits, dlog, zb, gb, h, c, dcc, tmp, dh0, dc0;
    std::vector<int> lens, lens_d, dlens, U, tgt; std::vector<float> At; std::vector<unsigned char> avalid;
    double loss_dec = 0, loss_aux = 0, acc = 0;
};

static Seg add(Net& n, int r, int c) { Seg s{n.np, r, c}; n.np += (size_t)r * c; return s; }

// one LSTM direction, forward.  inp rows m = t*B + b.  reverse: processing step s reads time len-1-s.
static void lstm_fwd(Net& n, Dir& d, int S, int B, int H, const float* Gx /*[S*B][4H]*/, const int* lens, bool reverse, const float* kernel_h /*[H][4H]*/,
                     float* Y, int ldy, int ycol, const float* h0, const float* c0, int ld0, int col0, float forget_bias) {
    const int H4 = 4 * H;
    pack_b(d.Kh, false, H, H4, kernel_h, H4);
    d.G.resize((size_t)S * B * H4); d.Cs.resize((size_t)S * B * H); d.Hp.resize((size_t)S * B * H); d.Cp.resize((size_t)S * B * H);
    memset(d.Hp.p, 0, sizeof(float) * (size_t)S * B * H);            // (rows of padding steps enter the dW_h product with dG = 0)
    n.zb.resize((size_t)B * H4); n.h.resize((size_t)B * H); n.c.resize((size_t)B * H);
    float *h = n.h.p, *c = n.c.p, *z = n.zb.p;
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int u = 0; u < H; ++u) { h[(size_t)b * H + u] = h0 ? h0[(size_t)b * ld0 + col0 + u] : 0.f; c[(size_t)b * H + u] = c0 ? c0[(size_t)b * ld0 + col0 + u] : 0.f; }
    for (int s = 0; s < S; ++s) {
#pragma omp parallel for
        for (int b = 0; b < B; ++b) {
            const int t = std::min(std::max(reverse ? lens[b] - 1 - s : s, 0), S - 1);
            memcpy(z + (size_t)b * H4, Gx + ((size_t)t * B + b) * H4, sizeof(float) * H4);
        }
        gemm_pb(false, B, h, H, d.Kh, z, H4, true, nullptr);
#pragma omp parallel for
        for (int b = 0; b < B; ++b) {
            if (s >= lens[b]) continue;
            const int t = reverse ? lens[b] - 1 - s : s;
            const size_t m = (size_t)t * B + b;
            const float* zr = z + (size_t)b * H4;
            float *hr = h + (size_t)b * H, *cr = c + (size_t)b * H, *g = d.G.p + m * H4;
            float* y = Y + m * ldy + ycol;
            for (int u = 0; u < H; ++u) {
                const float gi = sigm(zr[u]), gj = tanhf(zr[H + u]), gf = sigm(zr[2 * H + u] + forget_bias), go = sigm(zr[3 * H + u]);
                d.Hp.p[m * H + u] = hr[u]; d.Cp.p[m * H + u] = cr[u];
                const float cn = gf * cr[u] + gi * gj, hn = go * tanhf(cn);
                g[u] = gi; g[H + u] = gj; g[2 * H + u] = gf; g[3 * H + u] = go;
                d.Cs.p[m * H + u] = cn; cr[u] = cn; hr[u] = hn; y[u] = hn;
            }
        }
    }
    d.hT.assign(h, h + (size_t)B * H); d.cT.assign(c, c + (size_t)B * H);
}
// BPTT of one direction.  dY: gradient wrt h_t at its time index (rows m, columns ycol..ycol+H).  dG out [S*B][4H] (zero on padding).
static void lstm_bwd(Net& n, Dir& d, int S, int B, int H, const int* lens, bool reverse, const float* kernel_h, const float* dY, int lddy, int ycol,
                     const float* dhf, const float* dcf, int ldf, int colf, float* dh0, float* dc0 /* [B][H] or null */) {
    const int H4 = 4 * H;
    pack_b(d.KhT, true, H4, H, kernel_h, H4);                     // op(B)[k = gate column][n = unit] = kernel_h[n][k]
    d.dG.resize((size_t)S * B * H4);
    memset(d.dG.p, 0, sizeof(float) * (size_t)S * B * H4);
    n.gb.resize((size_t)B * H4); n.zb.resize((size_t)B * H4); n.dcc.resize((size_t)B * H);
    float *gn = n.gb.p, *rec = n.zb.p, *dcc = n.dcc.p;
    memset(dcc, 0, sizeof(float) * (size_t)B * H);
    for (int s = S - 1; s >= -1; --s) {
        if (s < 0 && !dh0) break;
#pragma omp parallel for
        for (int b = 0; b < B; ++b) {
            float* o = gn + (size_t)b * H4;
            const bool has_next = s + 1 < lens[b];
            if (!has_next) { memset(o, 0, sizeof(float) * H4); continue; }
            const int tn = reverse ? lens[b] - 1 - (s + 1) : s + 1;
            memcpy(o, d.dG.p + ((size_t)tn * B + b) * H4, sizeof(float) * H4);
        }
        gemm_pb(false, B, gn, H4, d.KhT, rec, H, false, nullptr);
        if (s < 0) {
#pragma omp parallel for
            for (int b = 0; b < B; ++b)
                for (int u = 0; u < H; ++u) {
                    const bool a0 = lens[b] > 0;
                    dh0[(size_t)b * H + u] = a0 ? rec[(size_t)b * H + u] : (dhf ? dhf[(size_t)b * ldf + colf + u] : 0.f);
                    dc0[(size_t)b * H + u] = a0 ? dcc[(size_t)b * H + u] : (dcf ? dcf[(size_t)b * ldf + colf + u] : 0.f);
                }
            break;
        }
#pragma omp parallel for
        for (int b = 0; b < B; ++b) {
            if (s >= lens[b]) continue;
            const int t = reverse ? lens[b] - 1 - s : s;
            const size_t m = (size_t)t * B + b;
            const bool has_next = s + 1 < lens[b], is_last = s == lens[b] - 1;
            const float* g = d.G.p + m * H4;
            float* dg = d.dG.p + m * H4;
            for (int u = 0; u < H; ++u) {
                const float gi = g[u], gj = g[H + u], gf = g[2 * H + u], go = g[3 * H + u];
                float dh = dY[m * lddy + ycol + u] + rec[(size_t)b * H + u];
                if (is_last && dhf) dh += dhf[(size_t)b * ldf + colf + u];
                const float dc_in = has_next ? dcc[(size_t)b * H + u] : (dcf ? dcf[(size_t)b * ldf + colf + u] : 0.f);
                const float tc = tanhf(d.Cs.p[m * H + u]);
                const float dct = dc_in + dh * go * (1.f - tc * tc);
                dg[3 * H + u] = dh * tc * go * (1.f - go);
                dg[u] = dct * gj * gi * (1.f - gi);
                dg[H + u] = dct * gi * (1.f - gj * gj);
                dg[2 * H + u] = dct * d.Cp.p[m * H + u] * gf * (1.f - gf);
                dcc[(size_t)b * H + u] = dct * gf;
            }
        }
    }
}

// feed-forward head (hidden layers ReLU + dropout, last layer linear, stored TRANSPOSED [out][in]: trainers.py:513-520)
static void ff_fwd(Net& n, const std::vector<int>& sz, const std::vector<Seg>& W, const std::vector<Seg>& Bv, std::vector<Buf>& act, const float* x, int ldx,
                   int M, bool train, uint64_t stream) {
    const int nl = (int)sz.size() - 1;
    act.resize(nl);
    const float* cur = x; int ld = ldx;
    for (int i = 0; i < nl; ++i) {
        const bool last = i == nl - 1;
        act[i].resize((size_t)M * sz[i + 1]);
        gemm(false, last, M, sz[i + 1], sz[i], cur, ld, n.p.data() + W[i].off, last ? sz[i] : sz[i + 1], act[i].p, sz[i + 1], false, n.p.data() + Bv[i].off);
        if (!last) {
            const float rate = train ? n.s.ff_drop : 0.f;
            float* a = act[i].p; const size_t tot = (size_t)M * sz[i + 1];
#pragma omp parallel for
            for (size_t k = 0; k < tot; ++k) a[k] = a[k] > 0.f ? a[k] * dscale(n.seed, stream + i, k, rate) : 0.f;
        }
        cur = act[i].p; ld = sz[i + 1];
    }
}
// dout [M][out] (consumed); returns d(input) in dx [M][ldx-wide block] (accumulated when acc)
static void ff_bwd(Net& n, const std::vector<int>& sz, const std::vector<Seg>& W, const std::vector<Seg>& Bv, std::vector<Buf>& act, const float* x, int ldx,
                   int M, float* dout, float* dx, int lddx, bool acc, bool train) {
    const int nl = (int)sz.size() - 1;
    float* d = dout;
    static Buf t0, t1;
    for (int i = nl - 1; i >= 0; --i) {
        const bool last = i == nl - 1;
        const float* xin = i == 0 ? x : act[i - 1].p; const int ldi = i == 0 ? ldx : sz[i];
        float* gw = n.g.data() + W[i].off;
        if (last) gemm(true, false, sz[i + 1], sz[i], M, d, sz[i + 1], xin, ldi, gw, sz[i], false);        // [out][in] = d^T . x
        else gemm(true, false, sz[i], sz[i + 1], M, xin, ldi, d, sz[i + 1], gw, sz[i + 1], false);           // [in][out] = x^T . d
        colsum(M, sz[i + 1], d, sz[i + 1], n.g.data() + Bv[i].off);
        if (i == 0) { gemm(false, !last, M, sz[0], sz[1], d, sz[1], n.p.data() + W[0].off, last ? sz[0] : sz[1], dx, lddx, acc); break; }
        Buf& nx = (i & 1) ? t1 : t0;
        nx.resize((size_t)M * sz[i]);
        gemm(false, !last, M, sz[i], sz[i + 1], d, sz[i + 1], n.p.data() + W[i].off, last ? sz[i] : sz[i + 1], nx.p, sz[i], false);
        const float keep = (train && n.s.ff_drop > 0.f) ? 1.f / (1.f - n.s.ff_drop) : 1.f;
        const float* a = act[i - 1].p; float* q = nx.p; const size_t tot = (size_t)M * sz[i];
#pragma omp parallel for
        for (size_t k = 0; k < tot; ++k) q[k] = a[k] > 0.f ? q[k] * keep : 0.f;
        d = nx.p;
    }
}
}  // namespace

static bool g_prof = getenv("E2T_CPU_PROF") != nullptr;
static double g_t0 = 0;
static void tick(const char* what) { if (!g_prof) return; const double t = omp_get_wtime(); if (what) fprintf(stderr, "  %-28s %8.1f ms\n", what, (t - g_t0) * 1e3); g_t0 = t; }
extern "C" {
// cfg: C, N, F, nl, H[0..nl), E, Hd, V, aux_layer (-1: none), naux hidden layers, their sizes, aux_dim, B, T, L; fcfg: ff_drop, rnn_drop,
// forget_bias, aux_scale, dec_scale.  Parameter order of the flat array (e2t_cpu_layout reports offsets): conv W [N*C][F], conv b, per
// layer and direction kernel [(D+H)][4H] + bias [4H], aux FF (weights, biases per layer; last layer transposed), embedding [V][E],
// decoder kernel [(E+Hd)][4Hd] + bias, projection (one transposed layer [V][Hd] + bias).
void* e2t_cpu_create(const int* cfg, const float* fcfg) {
    Net* n = new Net();
    Spec& s = n->s; int k = 0;
    s.C = cfg[k++]; s.N = cfg[k++]; s.F = cfg[k++]; s.nl = cfg[k++];
    for (int l = 0; l < s.nl; ++l) s.H[l] = cfg[k++];
    s.E = cfg[k++]; s.Hd = cfg[k++]; s.V = cfg[k++]; s.aux_layer = cfg[k++]; s.naux = cfg[k++];
    for (int i = 0; i < s.naux; ++i) s.auxh[i] = cfg[k++];
    s.aux_dim = cfg[k++]; s.B = cfg[k++]; s.T = cfg[k++]; s.L = cfg[k++];
    s.S = (s.T + s.N - 1) / s.N;
    s.ff_drop = fcfg[0]; s.rnn_drop = fcfg[1]; s.forget_bias = fcfg[2]; s.aux_scale = fcfg[3]; s.dec_scale = fcfg[4];
    n->conv_w = add(*n, s.N * s.C, s.F); n->conv_b = add(*n, 1, s.F);
    for (int l = 0; l < s.nl; ++l) {
        const int D = l == 0 ? s.F : 2 * s.H[l - 1];
        for (int d = 0; d < 2; ++d) { n->enc_k.push_back(add(*n, D + s.H[l], 4 * s.H[l])); n->enc_b.push_back(add(*n, 1, 4 * s.H[l])); }
    }
    if (s.aux_layer >= 0) {
        n->aux_sizes.push_back(2 * s.H[s.aux_layer]);
        for (int i = 0; i < s.naux; ++i) n->aux_sizes.push_back(s.auxh[i]);
        n->aux_sizes.push_back(s.aux_dim);
        for (size_t i = 0; i + 1 < n->aux_sizes.size(); ++i) {
            const bool last = i + 2 == n->aux_sizes.size();
            n->aux_w.push_back(last ? add(*n, n->aux_sizes[i + 1], n->aux_sizes[i]) : add(*n, n->aux_sizes[i], n->aux_sizes[i + 1]));
            n->aux_b.push_back(add(*n, 1, n->aux_sizes[i + 1]));
        }
    }
    n->emb = add(*n, s.V, s.E);
    n->dec_k = add(*n, s.E + s.Hd, 4 * s.Hd); n->dec_b = add(*n, 1, 4 * s.Hd);
    n->proj_sizes = {s.Hd, s.V};
    n->proj_w.push_back(add(*n, s.V, s.Hd)); n->proj_b.push_back(add(*n, 1, s.V));
    n->p.assign(n->np, 0.f); n->g.assign(n->np, 0.f); n->m.assign(n->np, 0.f); n->v.assign(n->np, 0.f); n->ema.assign(n->np, 0.f);
    n->dirs.resize(2 * s.nl); n->Y.resize(s.nl); n->Yd.resize(s.nl); n->dY.resize(s.nl);
    return n;
}
void e2t_cpu_destroy(void* h) { delete (Net*)h; }
long e2t_cpu_num_params(void* h) { return (long)((Net*)h)->np; }
float* e2t_cpu_params(void* h) { return ((Net*)h)->p.data(); }
float* e2t_cpu_grads(void* h) { return ((Net*)h)->g.data(); }
float* e2t_cpu_ema(void* h) { return ((Net*)h)->ema.data(); }
void e2t_cpu_init_ema(void* h) { Net* n = (Net*)h; n->ema = n->p; }
void e2t_cpu_losses(void* h, double* out) { Net* n = (Net*)h; out[0] = n->loss_dec; out[1] = n->loss_aux; out[2] = n->acc; }

// forward + backward (gradients in e2t_cpu_grads); x [B][T][C] zero padded beyond lens; tokens [B][L] (0 = padding); aux [B][T][aux_dim].
void e2t_cpu_fwd_bwd(void* hnd, const float* x, const int* lens_in, const int* tokens, const float* aux, const int* aux_lens, int train) {
    Net& n = *(Net*)hnd; const Spec& s = n.s;
    const int B = s.B, T = s.T, C = s.C, N = s.N, S = s.S, F = s.F, L = s.L, M = S * B, Kc = N * C;
    float* P = n.p.data(); float* G = n.g.data();
    const double t_all = omp_get_wtime();
    tick(nullptr);
    std::fill(n.g.begin(), n.g.end(), 0.f);
    n.lens.assign(lens_in, lens_in + B); n.lens_d.resize(B);
    for (int b = 0; b < B; ++b) n.lens_d[b] = (n.lens[b] + N - 1) / N;
    const float ffr = train ? s.ff_drop : 0.f, rnr = train ? s.rnn_drop : 0.f;
    // a5 + a6: reverse, im2row (kernel width == stride), conv + ReLU + dropout, zero beyond the decimated length
    n.A.resize((size_t)M * Kc); n.E.resize((size_t)M * F);
#pragma omp parallel for collapse(2)
    for (int tp = 0; tp < S; ++tp)
        for (int b = 0; b < B; ++b) {
            float* a = n.A.p + ((size_t)tp * B + b) * Kc;
            for (int w = 0; w < N; ++w) {
                const int tt = tp * N + w;
                if (tt < n.lens[b]) memcpy(a + (size_t)w * C, x + ((size_t)b * T + (n.lens[b] - 1 - tt)) * C, sizeof(float) * C);
                else memset(a + (size_t)w * C, 0, sizeof(float) * C);
            }
        }
    gemm(false, false, M, F, Kc, n.A.p, Kc, P + n.conv_w.off, F, n.E.p, F, false, P + n.conv_b.off);
#pragma omp parallel for
    for (size_t k = 0; k < (size_t)M * F; ++k) {
        const int m = (int)(k / F), tp = m / B, b = m % B;
        const float e = n.E.p[k];
        n.E.p[k] = (tp < n.lens_d[b] && e > 0.f) ? e * dscale(n.seed, 1, k, ffr) : 0.f;
    }
    tick("conv front-end");
    // a7: stacked bidirectional LSTM
    const float* inp = n.E.p; int D = F;
    for (int l = 0; l < s.nl; ++l) {
        const int H = s.H[l], H4 = 4 * H;
        n.Y[l].resize((size_t)M * 2 * H); n.Yd[l].resize((size_t)M * 2 * H);
        memset(n.Y[l].p, 0, sizeof(float) * (size_t)M * 2 * H);
        for (int d = 0; d < 2; ++d) {
            Dir& dr = n.dirs[l * 2 + d];
            const Seg kk = n.enc_k[l * 2 + d];
            dr.Gx.resize((size_t)M * H4);
            gemm(false, false, M, H4, D, inp, D, P + kk.off, H4, dr.Gx.p, H4, false, P + n.enc_b[l * 2 + d].off);
            lstm_fwd(n, dr, S, B, H, dr.Gx.p, n.lens_d.data(), d == 1, P + kk.off + (size_t)D * H4, n.Y[l].p, 2 * H, d * H, nullptr, nullptr, 0, 0, s.forget_bias);
        }
        const float* y = n.Y[l].p; float* yd = n.Yd[l].p;
#pragma omp parallel for
        for (size_t k = 0; k < (size_t)M * 2 * H; ++k) yd[k] = y[k] * dscale(n.seed, 10 + l, k, rnr);
        inp = n.Yd[l].p; D = 2 * H;
    }
    tick("encoder forward");
    const int Ht = s.H[s.nl - 1], Hd = s.Hd;
    std::vector<float> h0((size_t)B * Hd), c0((size_t)B * Hd);
    for (int d = 0; d < 2; ++d)
        for (int b = 0; b < B; ++b)
            for (int u = 0; u < Ht; ++u) { h0[(size_t)b * Hd + d * Ht + u] = n.dirs[(s.nl - 1) * 2 + d].hT[(size_t)b * Ht + u]; c0[(size_t)b * Hd + d * Ht + u] = n.dirs[(s.nl - 1) * 2 + d].cT[(size_t)b * Ht + u]; }
    // a8: auxiliary head (squared error against the reversed, every-N-th-sample targets)
    n.loss_aux = 0;
    Buf dP;
    const bool use_aux = s.aux_layer >= 0 && aux != nullptr && s.aux_scale != 0.f;
    if (use_aux) {
        const int K = s.aux_dim, la = s.aux_layer;
        ff_fwd(n, n.aux_sizes, n.aux_w, n.aux_b, n.aux_act, n.Yd[la].p, 2 * s.H[la], M, train, 100);
        const float* Po = n.aux_act.back().p;
        dP.resize((size_t)M * K);
        long nval = 0;
        for (int b = 0; b < B; ++b) nval += (aux_lens[b] + N - 1) / N;
        nval = std::max(nval, 1L);
        double tot = 0;
#pragma omp parallel for reduction(+ : tot)
        for (int m = 0; m < M; ++m) {
            const int tp = m / B, b = m % B;
            const bool ok = tp * N < aux_lens[b];
            for (int k = 0; k < K; ++k) {
                float d = 0.f;
                if (ok) { const float at = aux[((size_t)b * T + (aux_lens[b] - 1 - tp * N)) * K + k]; d = Po[(size_t)m * K + k] - at; }
                tot += (double)d * d;
                dP.p[(size_t)m * K + k] = 2.f * d / (float)(nval * K) * s.aux_scale;
            }
        }
        n.loss_aux = tot / (double)(nval * K);
    }
    tick("aux head forward");
    // a9: decoder, teacher forced (<EOS> = 1 doubles as the start symbol)
    const int Md = L * B, E = s.E, Hd4 = 4 * Hd, V = s.V;
    n.dlens.assign(B, 0); n.U.resize(Md); n.tgt.resize(Md);
    for (int b = 0; b < B; ++b) {
        int len = 0; for (int l = 0; l < L; ++l) len += tokens[(size_t)b * L + l] != 0;
        n.dlens[b] = len;
        for (int l = 0; l < L; ++l) { n.U[(size_t)l * B + b] = l == 0 ? 1 : tokens[(size_t)b * L + l - 1]; n.tgt[(size_t)l * B + b] = tokens[(size_t)b * L + l]; }
    }
    n.e.resize((size_t)Md * E);
#pragma omp parallel for
    for (int m = 0; m < Md; ++m)
        for (int k = 0; k < E; ++k) n.e.p[(size_t)m * E + k] = P[n.emb.off + (size_t)n.U[m] * E + k] * dscale(n.seed, 20, (size_t)m * E + k, ffr);
    n.dec.Gx.resize((size_t)Md * Hd4);
    gemm(false, false, Md, Hd4, E, n.e.p, E, P + n.dec_k.off, Hd4, n.dec.Gx.p, Hd4, false, P + n.dec_b.off);
    n.Ydec.resize((size_t)Md * Hd); memset(n.Ydec.p, 0, sizeof(float) * (size_t)Md * Hd);
    lstm_fwd(n, n.dec, L, B, Hd, n.dec.Gx.p, n.dlens.data(), false, P + n.dec_k.off + (size_t)E * Hd4, n.Ydec.p, Hd, 0, h0.data(), c0.data(), Hd, 0, s.forget_bias);
    n.Hdrop.resize((size_t)Md * Hd);
#pragma omp parallel for
    for (size_t k = 0; k < (size_t)Md * Hd; ++k) n.Hdrop.p[k] = n.Ydec.p[k] * dscale(n.seed, 30, k, rnr);
    ff_fwd(n, n.proj_sizes, n.proj_w, n.proj_b, n.proj_act, n.Hdrop.p, Hd, Md, train, 200);
    const float* lg = n.proj_act.back().p;
    n.dlog.resize((size_t)Md * V);
    long ntok = 0; for (int b = 0; b < B; ++b) ntok += n.dlens[b];
    ntok = std::max(ntok, 1L);
    double ce = 0, correct = 0;
#pragma omp parallel for reduction(+ : ce, correct)
    for (int m = 0; m < Md; ++m) {
        const int l = m / B, b = m % B;
        const float* r = lg + (size_t)m * V; float* dl = n.dlog.p + (size_t)m * V;
        if (l >= n.dlens[b]) { memset(dl, 0, sizeof(float) * V); continue; }
        float mx = r[0]; int am = 0;
        for (int j = 1; j < V; ++j) if (r[j] > mx) { mx = r[j]; am = j; }
        double se = 0; for (int j = 0; j < V; ++j) se += exp((double)r[j] - mx);
        const double lse = mx + log(se);
        const int t = n.tgt[m];
        ce += lse - r[t]; correct += am == t;
        const float sc = s.dec_scale / (float)ntok;
        for (int j = 0; j < V; ++j) dl[j] = (float)exp((double)r[j] - lse) * sc;
        dl[t] -= sc;
    }
    n.loss_dec = ce / ntok; n.acc = correct / ntok;

    tick("decoder forward + CE");
    // ---------------- backward ----------------
    n.dH.resize((size_t)Md * Hd);
    ff_bwd(n, n.proj_sizes, n.proj_w, n.proj_b, n.proj_act, n.Hdrop.p, Hd, Md, n.dlog.p, n.dH.p, Hd, false, train);
#pragma omp parallel for
    for (size_t k = 0; k < (size_t)Md * Hd; ++k) n.dH.p[k] *= dscale(n.seed, 30, k, rnr);
    n.dh0.resize((size_t)B * Hd); n.dc0.resize((size_t)B * Hd);
    lstm_bwd(n, n.dec, L, B, Hd, n.dlens.data(), false, P + n.dec_k.off + (size_t)E * Hd4, n.dH.p, Hd, 0, nullptr, nullptr, 0, 0, n.dh0.p, n.dc0.p);
    gemm(true, false, E, Hd4, Md, n.e.p, E, n.dec.dG.p, Hd4, G + n.dec_k.off, Hd4, false);
    gemm(true, false, Hd, Hd4, Md, n.dec.Hp.p, Hd, n.dec.dG.p, Hd4, G + n.dec_k.off + (size_t)E * Hd4, Hd4, false);
    colsum(Md, Hd4, n.dec.dG.p, Hd4, G + n.dec_b.off);
    n.de.resize((size_t)Md * E);
    gemm(false, true, Md, E, Hd4, n.dec.dG.p, Hd4, P + n.dec_k.off, Hd4, n.de.p, E, false);
    for (int m = 0; m < Md; ++m)                                       // scatter-add (serial: rows of one token collide)
        for (int k = 0; k < E; ++k) G[n.emb.off + (size_t)n.U[m] * E + k] += n.de.p[(size_t)m * E + k] * dscale(n.seed, 20, (size_t)m * E + k, ffr);
    tick("decoder backward");
    // encoder, top layer down
    for (int l = s.nl - 1; l >= 0; --l) {
        const int H = s.H[l], H4 = 4 * H, Dl = l == 0 ? F : 2 * s.H[l - 1];
        if (l == s.nl - 1) { n.dY[l].resize((size_t)M * 2 * H); memset(n.dY[l].p, 0, sizeof(float) * (size_t)M * 2 * H); }
        if (use_aux && l == s.aux_layer)
            ff_bwd(n, n.aux_sizes, n.aux_w, n.aux_b, n.aux_act, n.Yd[l].p, 2 * H, M, dP.p, n.dY[l].p, 2 * H, true, train);
        float* dy = n.dY[l].p;
#pragma omp parallel for
        for (size_t k = 0; k < (size_t)M * 2 * H; ++k) dy[k] *= dscale(n.seed, 10 + l, k, rnr);
        const float* in = l == 0 ? n.E.p : n.Yd[l - 1].p;
        float* din;
        if (l == 0) { n.dE.resize((size_t)M * F); din = n.dE.p; } else { n.dY[l - 1].resize((size_t)M * Dl); din = n.dY[l - 1].p; }
        for (int d = 0; d < 2; ++d) {
            Dir& dr = n.dirs[l * 2 + d];
            const Seg kk = n.enc_k[l * 2 + d];
            const bool top = l == s.nl - 1;
            lstm_bwd(n, dr, S, B, H, n.lens_d.data(), d == 1, P + kk.off + (size_t)Dl * H4, dy, 2 * H, d * H, top ? n.dh0.p : nullptr, top ? n.dc0.p : nullptr, Hd, d * H,
                     nullptr, nullptr);
            gemm(true, false, Dl, H4, M, in, Dl, dr.dG.p, H4, G + kk.off, H4, false);
            gemm(true, false, H, H4, M, dr.Hp.p, H, dr.dG.p, H4, G + kk.off + (size_t)Dl * H4, H4, false);
            colsum(M, H4, dr.dG.p, H4, G + n.enc_b[l * 2 + d].off);
            gemm(false, true, M, Dl, H4, dr.dG.p, H4, P + kk.off, H4, din, Dl, d == 1);
        }
    }
    tick("encoder backward");
    // conv front-end: E > 0 iff ReLU active, kept by the dropout and inside the utterance
    const float keep = ffr > 0.f ? 1.f / (1.f - ffr) : 1.f;
#pragma omp parallel for
    for (size_t k = 0; k < (size_t)M * F; ++k) n.dE.p[k] = n.E.p[k] > 0.f ? n.dE.p[k] * keep : 0.f;
    gemm(true, false, Kc, F, M, n.A.p, Kc, n.dE.p, F, G + n.conv_w.off, F, false);
    colsum(M, F, n.dE.p, F, G + n.conv_b.off);
    tick("conv backward");
    if (g_prof) fprintf(stderr, "  total %.1f ms\n", (omp_get_wtime() - t_all) * 1e3);
}

// a10: Adam (TF1 formulation) + EMA on every parameter
void e2t_cpu_adam(void* hnd, float lr, float b1, float b2, float eps, float decay) {
    Net& n = *(Net*)hnd;
    const int t = ++n.step;
    n.seed += 1;
    const float lr_t = lr * sqrtf(1.f - powf(b2, (float)t)) / (1.f - powf(b1, (float)t));
    float *p = n.p.data(), *g = n.g.data(), *m = n.m.data(), *v = n.v.data(), *e = n.ema.data();
#pragma omp parallel for
    for (size_t i = 0; i < n.np; ++i) {
        m[i] = b1 * m[i] + (1.f - b1) * g[i];
        v[i] = b2 * v[i] + (1.f - b2) * g[i] * g[i];
        p[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
        e[i] = decay * e[i] + (1.f - decay) * p[i];
    }
}
int e2t_cpu_threads(void) { return omp_get_max_threads(); }
void e2t_cpu_set_threads(int t) { omp_set_num_threads(t); }
}
// diagnostics: GFLOP/s of the GEMM on one shape (tests / tuning only)
extern "C" double e2t_cpu_gemm_gflops(int ta, int tb, int M, int N, int K, int reps) {
    std::vector<float> A((size_t)M * K, 0.5f), B((size_t)K * N, 0.25f), C((size_t)M * N);
    gemm(ta, tb, M, N, K, A.data(), ta ? M : K, B.data(), tb ? K : N, C.data(), N, false);
    const double t0 = omp_get_wtime();
    for (int r = 0; r < reps; ++r) gemm(ta, tb, M, N, K, A.data(), ta ? M : K, B.data(), tb ? K : N, C.data(), N, false);
    return 2.0 * M * N * K * reps / (omp_get_wtime() - t0) * 1e-9;
}
