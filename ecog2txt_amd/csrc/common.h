// Shared device helpers for the gfx950 (CDNA4) kernels of the ECoG->text hot path.
// wave = 64 lanes; bf16 operands, fp32 accumulate (MFMA 16x16x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;        // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;         // MFMA 16x16 accumulator

#define E2T_WAVE 64

// fp32 -> bf16, round-to-nearest-even (bit-identical to oracle/bf16.py:round_bf16)
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two at once on the hardware converter (v_cvt_pk_bf16_f32): low half = bf16(a), high half = bf16(b).  Same bits as f2bf for
// every fp32 pattern that is not a NaN (a one-off probe walked all 2^32 patterns on the GPU in round 2: 0 mismatches); a NaN
// stays a (quiet) NaN here, f2bf carries its upper bits along
typedef float e2t_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 e2t_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f2bf_pk(float a, float b) {
    const e2t_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, e2t_bf16x2));
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly for |x| large
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (bit-identical to oracle/philox.py).  Element e of a dropped
// tensor uses counter (e>>2, stream) and output lane e&3; key = 64-bit seed.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct DropCfg {
    float rate;                  // 0 => disabled
    unsigned long long seed;     // base seed; effective seed = seed + *step (if step != null)
    const int* step;             // device step counter (graph-replay safe) or null
    unsigned stream;             // tensor id
};

// returns the multiplicative scale for logical element `idx`: 0 or 1/(1-rate)
__device__ __forceinline__ float drop_scale(const DropCfg& d, unsigned long long idx) {
    if (d.rate <= 0.0f) return 1.0f;
    unsigned long long seed = d.seed + (d.step ? (unsigned long long)(*d.step) : 0ull);
    unsigned long long ctr = idx >> 2;
    unsigned r[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), d.stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
    unsigned v = r[idx & 3ull] >> 8;
    unsigned thresh = (unsigned)(d.rate * 16777216.0f);
    return v >= thresh ? 1.0f / (1.0f - d.rate) : 0.0f;
}

// ---------------------------------------------------------------------------
// Direct-to-LDS DMA (global_load_lds_dwordx4): every lane supplies its own global source
// address; the 64 x 16 B land LINEARLY at LDS byte address `lds_addr` (wave-uniform) + lane*16.
// Issued from inline asm on purpose: hipcc tracks the builtin form as a pending LDS write and
// then drains it (s_waitcnt vmcnt(0)) in front of the next ds_read of ANY address, which
// serialises a double-buffered pipeline (seen in the ISA; cdna_hip_programming.md 5.7).  An asm
// DMA is invisible to that bookkeeping: completion is awaited explicitly with dma_wait_all()
// followed by a barrier before any wave reads the tile.  M0 is saved/restored inside the
// statement (compiler-reserved register).
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void e2t_lds_void;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(e2t_lds_void*)p;
}
__device__ __forceinline__ void dma16_to_lds(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(m) : "memory");
}
// same, address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit VALU address arithmetic per piece
__device__ __forceinline__ void dma16_to_lds_sbase(const void* sbase, unsigned voff_bytes, unsigned lds_addr) {
    unsigned keep;
    const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(m) : "memory");
}
// same, with the sc1 cache policy: bypasses the CU's L1, so data another workgroup published with
// write-through (sc1 / agent-scope atomic) stores during THIS launch is seen (cdna_hip_programming.md G16, R1)
__device__ __forceinline__ void dma16_to_lds_sc1(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(m) : "memory");
}
// same, non-temporal: data that is read exactly once (the input projections a recurrence streams) should not displace the lines
// the hand-off lives on (the exchange buffers in the XCD's L2)
__device__ __forceinline__ void dma16_to_lds_nt(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(m) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dma_wait_all_but2() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
// at most N vector-memory operations of this wave still outstanding (they retire in issue order)
template <int N> __device__ __forceinline__ void dma_wait_but() { static_assert(N >= 0 && N < 64, "vmcnt"); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// wave-level reductions (64 lanes) via shuffles
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Diagnostic switches.  The PRODUCT library reads no environment variables on its compute paths: every E2T_* knob that
// selects a kernel variant or hands a debug buffer to a kernel exists only in the diagnostics build (build.sh with
// E2T_DEBUG=1 -> libecog2txt_hip_dbg.so, used by scripts/: -DE2T_DEBUG); here it folds to its default at compile time.
#ifdef E2T_DEBUG
#include <stdlib.h>
static inline int e2t_dbg_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline const char* e2t_dbg_str(const char* name) { return getenv(name); }
static inline void* e2t_dbg_ptr(const char* name) { const char* e = getenv(name); return e ? (void*)strtoull(e, nullptr, 0) : nullptr; }
#else
static inline constexpr int e2t_dbg_int(const char*, int dflt) { return dflt; }
static inline constexpr const char* e2t_dbg_str(const char*) { return nullptr; }
static inline constexpr void* e2t_dbg_ptr(const char*) { return nullptr; }
#endif

// host-side status plumbing -------------------------------------------------
#define E2T_OK 0
#define E2T_ERR_ARG 1
#define E2T_ERR_HIP 2

int e2t_set_error(const char* fmt, ...);
#define E2T_CHECK_ARG(cond)                                                             \
    do { if (!(cond)) return e2t_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond) , E2T_ERR_ARG; } while (0)
#define E2T_LAUNCH_CHECK()                                                              \
    do { hipError_t e_ = hipGetLastError();                                             \
         if (e_ != hipSuccess) { e2t_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return E2T_ERR_HIP; } \
    } while (0)
