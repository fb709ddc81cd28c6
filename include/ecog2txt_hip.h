/* ecog2txt_hip.h -- C ABI of libecog2txt_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the ONE hot path of jgmakin/ecog2txt: everything that
 * `machine_learning.neural_networks.sequence_networks.SequenceNetwork` executes
 * on the device when the reference calls
 *     net.fit(subjects, ...)                   ecog2txt/trainers.py:318, 355, 367
 *     net.restore_and_assess(subjects, epoch)  ecog2txt/trainers.py:379-380
 *     net.restore_and_get_saliencies(...)      ecog2txt/trainers.py:722-725
 * The reference has no FFI of its own (pure Python over TF1.x),
 * so these entry points are [BUILD-DEFINES]: one per device stage of that path,
 * each citing the reference stage it replaces.  The Python host
 * (ecog2txt_amd/sequence_network.py) binds them with ctypes; see INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    stated otherwise; the caller owns all buffers; `stream` is a hipStream_t.
 *  - all launches are asynchronous on `stream`; no compute entry point synchronises,
 *    allocates or frees, so every call is hipGraph-capturable.  (Only e2t_comm_init /
 *    e2t_comm_destroy create and release resources: a communicator, its stream, its events.)
 *  - return 0 on success, non-zero on failure; e2t_last_error() gives the text.
 *    Nothing throws across the boundary.
 *  - "bf16" buffers are raw uint16 bfloat16 bits.  Every bf16 matrix has a leading
 *    dimension that is a multiple of 8 elements and ZERO padding columns that no
 *    kernel ever writes (the GEMM reads K rounded up to 8).
 *  - sequence tensors are TIME-MAJOR on the device: row m = t * B + b.
 */
#ifndef ECOG2TXT_HIP_H
#define ECOG2TXT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2T_ABI_VERSION 9

int e2t_abi_version(void);
/* sizeof() of the structs that cross the boundary, for bindings to check their layouts against:
 * which = 0 e2t_gemm_epilogue, 1 e2t_lstm_desc, 2 e2t_pack_desc, 3 e2t_adam_hyper, 4 e2t_dropout, 5 e2t_gemm_call, 6 e2t_tile_desc [ABI 8];
 * -1 for any other value */
int e2t_sizeof(int which);
const char* e2t_last_error(void);          /* thread-local, host pointer */
/* number of compute units / XCDs of `device`, 0 if no device is usable */
int e2t_device_cus(int device);

/* ---- dropout descriptor (FF_dropout / RNN_dropout, mocha-1_word_sequence.yaml:6,13) ---- */
typedef struct e2t_dropout {
    float rate;                 /* 0 disables */
    unsigned long long seed;    /* Philox key; effective key = seed + *step */
    const int32_t* step;        /* device step counter or NULL */
    unsigned stream;            /* tensor id (oracle/seq2seq.py STREAM_*) */
} e2t_dropout;

/* ---- a2/a3: batch assembly from a partition kept resident in HBM (role of the tf.data pipeline that feeds net.fit,
 *      trainers.py:896-900): dst row r = src row idx[r] for r < n with idx[r] >= 0, else zeros (a padding utterance:
 *      zero samples = zero length, subjects.py:386-390).  Rows are row_words 32-bit words; idx is a DEVICE array. ---- */
/* dst[0..n) = value (32-bit words; fp32 0.0 is word 0) */
int e2t_fill_u32(void* dst, size_t n, uint32_t value, void* stream);
int e2t_gather_rows_u32(const void* src, const int32_t* idx, int n, int rows_out, size_t row_words, void* dst, void* stream);
/* ABI 6: the same row selection applied to `blocks` equally shaped blocks of rows -- block t of dst (dst_block_words apart) row r =
 * block t of src (src_block_words apart) row idx[r].  Batch assembly from a partition whose inputs are staged as the bf16 im2row
 * rows of e2t_conv_pack (block = decimated step t', one row per utterance: SURVEY.md 8 d4 "bf16 in"; reference input contract
 * trainers.py:808-818, subjects.py:386-390): the time-major conv operand of a batch is gathered in one launch. */
int e2t_gather_rows_blocks_u32(const void* src, const int32_t* idx, int n, int rows_out, size_t row_words, int blocks,
                               size_t src_block_words, size_t dst_block_words, void* dst, void* stream);

/* ---- a4: nn.sequences_tools (trainers.py:789-790, 806-807) ---- */
int e2t_seq_lengths_f32(const float* x, int B, int T, int C, int div, int32_t* lens, int32_t* lens_div, void* stream);
/* the same lengths for END-padded batches (subjects.py:386-390), searched from the tail: reads the padding plus <= 32 rows
 * per utterance instead of all T rows.  length = index of the last non-zero row + 1 (callers guarantee that no interior
 * row is all-zero; narrow or unaligned rows fall back to the full count). */
int e2t_seq_lengths_tail_f32(const float* x, int B, int T, int C, int div, int32_t* lens, int32_t* lens_div, void* stream);
int e2t_seq_lengths_i32(const int32_t* x, int B, int L, int pad, int div, int32_t* lens, int32_t* lens_div, void* stream);
int e2t_sum_i32(const int32_t* x, int n, int32_t* out, void* stream);
int e2t_sum_f32(const float* x, int n, const int32_t* count, float scale, float* out, void* stream);
/* ABI 9: two such sums over n elements each (same count) in ONE launch -- the decoder's loss and token accuracy, which sat as two
 * launches on the critical branch between the cross-entropy and the backward pass; same association, same bits */
int e2t_sum2_f32(const float* x0, const float* x1, int n, const int32_t* count, float scale0, float scale1, float* out0, float* out1,
                 void* stream);

/* ---- a5+a6: tf.reverse_sequence (trainers.py:808-810) fused with the im2row staging of
 *      SequenceNetwork._convolve_sequences (trainers.py:813-818): x [B][T][C] fp32 ->
 *      A [(T/N)*B][lda] bf16, row (t',b), column (w,c); columns N*C.. are zero except column N*C = 1.0 when lda > N*C
 *      (the ones column that makes A^T . dE = [dW; db] in e2t_gemm_tn_bf16)  ---- */
int e2t_conv_pack(const float* x, const int32_t* lens, int B, int T, int C, int N, void* A, int lda, void* stream);
/* same with the rows in GROUPED order for a stack of conv layers (total stride = product of the layers' strides,
 * trainers.py:406-407): row m = (tg*B + b)*G + g holds step t' = tg*G + g, G = product of the strides of the layers above;
 * S = ceil(T / (N*G)) * G rows per utterance.  The G steps a later layer folds into one are then adjacent rows, so that
 * layer's im2row operand is a VIEW of this layer's output ([S/G*B][G*ld]) and its input gradient a view of dE. */
int e2t_conv_pack_grouped(const float* x, const int32_t* lens, int B, int T, int C, int N, int G, void* A, int lda, void* stream);
/* a12: scatter d/dA [S*B][ldda] fp32 back to d/dx [B][T][C] (restore_and_get_saliencies, trainers.py:722-725) */
int e2t_conv_unpack_grad(const float* dA, int ldda, const int32_t* lens, int B, int T, int C, int N, float* dx, void* stream);
int e2t_conv_unpack_grad_grouped(const float* dA, int ldda, const int32_t* lens, int B, int T, int C, int N, int G, float* dx, void* stream);

/* ---- a8: _prepare_encoder_targets: reverse then [:, 0::N, :] (trainers.py:791-799) ---- */
int e2t_gather_rev_decim_f32(const float* a, const int32_t* tlens, int B, int T, int K, int N, float* out, void* stream);
int e2t_gather_rev_decim_i32(const int32_t* a, const int32_t* tlens, int B, int T, int N, int32_t* out, void* stream);
/* a9: teacher-forcing inputs/targets, time-major; <EOS> as start symbol */
int e2t_decoder_tokens(const int32_t* y, int B, int L, int eos, int32_t* U, int32_t* Tg, void* stream);

/* ---- the matmul: C[M][N] (+)= alpha * A[M][K] . B[N][K]^T with fused epilogue ---- */
#define E2T_GEMM_RELU 1
#define E2T_GEMM_OUT_BF16 2
#define E2T_GEMM_ACCUMULATE 4      /* fp32 output only */
#define E2T_GEMM_DROPOUT 8
#define E2T_GEMM_SPLITK 16         /* split K over workgroups: partials in splitk_ws, fixed-order reduce which applies the epilogue */
#define E2T_GEMM_KEEP_SLABS 64     /* ABI 8, K-major entry points, with slabs_out: a split product whose reduction has nothing between the sum and the
                                      store (a weight gradient: alpha 1, fp32, no bias / mask / diverted column) is NOT reduced -- its slabs stay
                                      in splitk_ws and *slabs_out says where: the optimiser kernel sums them as it reads the gradient
                                      (e2t_adam_pack_batch), so C is never written and never read.  The workspace must then be the caller's to
                                      keep until that kernel has run. */
#define E2T_GEMM_LAST_ROW_ONES 32  /* ABI 8, e2t_gemm_tn_bf16 only: column M-1 of the K-major operand A is all ones (the ones column that turns
                                      [x | 1]^T . dG into weights + bias), so row M-1 of the product is the column sums of B.  A hint: where
                                      that row would cost a launch of its own (the ragged edge of a 256 x 256-tiled product) it is computed
                                      by a column-sum pass over B instead (fp32 sums in a fixed order; alpha and E2T_GEMM_ACCUMULATE apply) */
/* ABI 8: where the partial sums of a split product were LEFT (E2T_GEMM_KEEP_SLABS): element (z, m, n) of the product is
 * sum over s < splits of slab[(z * splits + s) * stride + m * N + n]; splits == 1: nothing was left, C holds the product. */
typedef struct e2t_slab_info { const float* slab; int splits; int batch; long long stride; } e2t_slab_info;
typedef struct e2t_gemm_epilogue {
    const float* bias;             /* [N] or NULL */
    const void* relu_bwd_src;      /* bf16 [M][ld]: out = src != 0 ? out : 0 (ReLU/dropout backward) */
    int ld_relu_bwd_src;
    const int32_t* row_lens;       /* [rows_per_step] or NULL: row m kept iff m / rows_per_step < row_lens[m % rows_per_step] */
    int rows_per_step;
    float alpha;
    int flags;
    float drop_rate; unsigned long long drop_seed; const int32_t* drop_step; unsigned drop_stream; int drop_ld;
    float* last_col_out;           /* fp32 [M] or NULL: column N-1 of the product is written here instead of C
                                      (a ones row appended to B turns it into the bias gradient) */
    void* splitk_ws;               /* device workspace for split-K partial slabs (or NULL: never split).  When offered, the
                                      library also splits on its own if the product has few 128x128 tiles and K >= 1024 */
    size_t splitk_ws_bytes;
    int batch;                     /* > 1: that many independent products of the same shape in ONE launch; product z uses
                                      A + z*a_batch_stride, B + z*b_batch_stride, C + z*c_batch_stride (strides in ELEMENTS of
                                      the respective array; bias / masks / row_lens are shared).  0 or 1: a single product.
                                      (The two directions of a recurrent weight gradient: twice the tiles, half the splits.) */
    long long a_batch_stride, b_batch_stride, c_batch_stride;
    int row_group;                 /* 0 / 1: output rows are (step, utterance) time-major; G > 1: grouped order of e2t_conv_pack_grouped --
                                      only the row_lens mask looks at it */
    e2t_slab_info* slabs_out;      /* ABI 8, host pointer or NULL: filled by the K-major entry points when E2T_GEMM_KEEP_SLABS is set */
} e2t_gemm_epilogue;
int e2t_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const e2t_gemm_epilogue* ep /* host pointer or NULL */, void* stream);
/* Same product and epilogue with K-MAJOR operands: C[M][N] (+)= alpha * sum_k A[k][m] * B[k][n] (A bf16 [K][lda >= M],
 * B bf16 [K][ldb >= N]; lda, ldb multiples of 8; any K).  This is the shape of every weight gradient (K = S*B rows of
 * activations and of their gradients as the layers wrote them), so no operand is ever transposed in memory: fragments are
 * gathered from the K-major LDS tile with the transposing read ds_read_b64_tr_b16. */
int e2t_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const e2t_gemm_epilogue* ep, void* stream);
/* Several K-major products in ONE launch: the weight gradients of a backward stage (dW_x and dW_h of a layer; projection and
 * decoder kernels of the head), trainers.py:318 (fit) with the kernel shapes of :527-541.  Same results as n calls of
 * e2t_gemm_tn_bf16 with E2T_GEMM_SPLITK (the slabs are summed in fixed order); the K splits are chosen for the group as a whole,
 * so that the workgroups of all products fill whole rounds of the chip.  n <= 8; every call carries its own epilogue (ep != NULL;
 * the split-K workspace of the FIRST call's epilogue serves the whole group).  The epilogues must be plain (alpha, bias, accumulate,
 * last_col_out): a call carrying E2T_GEMM_RELU, E2T_GEMM_DROPOUT or relu_bwd_src is refused (E2T_ERR_ARG) -- a grouped product may
 * run unsplit, and only the split-K reduction applies those. */
typedef struct e2t_gemm_call {
    const void* A; int lda; const void* B; int ldb; void* C; int ldc; int M, N, K;
    const e2t_gemm_epilogue* ep;
} e2t_gemm_call;
int e2t_gemm_tn_group_bf16(int n, const e2t_gemm_call* calls /* host array */, void* stream);
/* which instance e2t_gemm_{nt,tn}_bf16 would run this product on: *tile = 128 or 256 (square tiles), *splits = K splits
 * (1: none).  For profiling tools that attribute time to kernel instances (bench.py). */
int e2t_gemm_plan(int tn, int M, int N, int K, const e2t_gemm_epilogue* ep, int* tile, int* splits);
/* ABI 6 -- measurement hook (bench.py's `roofline`): with a device buffer of 2 * slots unsigned 64-bit words set, the k-th GEMM launch
 * issued afterwards (slot k % slots; a grouped launch counts once, a product the library cuts into two launches twice) records
 * stamp[2*slot] = min over its workgroups of the start time, stamp[2*slot + 1] = max of the end time, on the chip-wide 100-MHz clock
 * -- the duration a kernel trace reports for the launch, readable inside a replayed hipGraph (the caller resets the words to
 * ~0 / 0 between replays).  e2t_gemm_stamp_kinds returns the instance of each slot (E2T_GEMM_KIND_*, -1 = unused).
 * buf = NULL switches the hook off (the default: the kernels then execute one scalar compare more). */
#define E2T_GEMM_STAMP_MAX 512
enum { E2T_GEMM_KIND_TN128G = 0, E2T_GEMM_KIND_TN128 = 1, E2T_GEMM_KIND_TN256 = 2, E2T_GEMM_KIND_NT128 = 3, E2T_GEMM_KIND_NT256 = 4 };
int e2t_gemm_stamps(void* buf, int slots);
int e2t_gemm_stamp_kinds(int* kinds, int n);
/* a5+a6 in ONE pass over x: E[(t',b)][n] = epilogue(sum_{w,c} bf16(x[b][len-1-(t'*N+w)][c]) * WT[n][w*C+c]),
 * WT bf16 [F][ldw] K-contiguous (the conv image of e2t_pack_batch), E bf16 [S*B][lde]; ep: bias, flags (OUT_BF16 required,
 * RELU, DROPOUT), row_lens / rows_per_step (decimated lengths: rows beyond them are zeroed), dropout fields, splitk_ws
 * (optional: lets the launcher cut K to fill the chip) -- as for e2t_gemm_nt_bf16.  A_out (may be NULL): the packed bf16
 * im2row copy [S*B][lda_out] that e2t_conv_pack would write, columns < N*C only, emitted on the way (training: the
 * weight-gradient product reads it).  Applicable when e2t_conv_fwd_fused_ok(C, F): C % 64 == 0, F <= 128. */
int e2t_conv_fwd_fused_ok(int C, int F);
int e2t_conv_fwd_fused(const float* x, const int32_t* lens, int B, int T, int C, int N, const void* WT, int ldw, void* E, int lde,
                       int F, void* A_out, int lda_out, const e2t_gemm_epilogue* ep, void* stream);
int e2t_transpose_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, void* stream);

/* ---- weight packing: fp32 masters -> bf16 operand images (after every optimiser step) ---- */
int e2t_cast_pack(const float* src, long row_stride, long col_stride, int R, int C, void* dst, int ld_dst, void* stream);
int e2t_pack_frag(const float* src, long n_stride, long k_stride, int Nn, int Kk, void* dst, void* stream);
/* every image in one launch: device table of descriptors (src = base + src_off).  A descriptor's work is cut into units
 * (the workgroup counts quoted per kind below); for kinds 1 and 3 a workgroup processes E2T_PACK_UNITS consecutive
 * units (all their loads in flight at once), so such a descriptor owns ceil(units / E2T_PACK_UNITS) workgroups.
 * ABI 7: a workgroup of kinds 5 and 6 takes TWO k-blocks (the workgroup counts below), and all kinds share one LDS staging buffer. */
#define E2T_PACK_UNITS 4
typedef struct e2t_pack_desc {
    int kind;            /* 0: dst[r][c] = bf16(src[r*s0 + c*s1]), r < d0, c < d1, leading dim ld
                            1: MFMA fragment image of Bn[n][k] = src[n*s0 + k*s1], n < d0, k < d1, ld = ceil(d1/32)
                            2: as 0 for sources with s0 == 1 (contiguous along r): 64x64 tiles through LDS,
                               ceil(d0/64)*ceil(d1/64) workgroups instead of d0*ceil(d1/256)
                            3: as 0, four columns per thread (s1 == 1; d1, s0, src_off, ld multiples of 4; dst 8-B aligned):
                               d0*ceil(d1/1024) workgroups
                            4: as 2 with 16-B loads / 8-B stores (d0, d1, s1, src_off, ld multiples of 4; dst 8-B aligned)
                            5: as 1 for s0 == 1 (contiguous along n; d0, s1, src_off multiples of 4): [32 k][64 n] blocks
                               through LDS, two k-blocks per workgroup: ceil(ceil(d0/16)/4) * ceil(ceil(d1/32)/2) workgroups
                            6: the four per-gate fragment images [4][ceil(d0/16)][ld] (contiguous at dst) of a gate-interleaved
                               source Bn_g[n][k] = src[(n*4 + g) + k*s1], n < d0 units, k < d1 (s1, src_off multiples of 4):
                               one pass over the source, two k-blocks per workgroup: ceil(d0/16) * ceil(ceil(d1/32)/2) workgroups */
    int first_block;     /* first 256-thread workgroup of this descriptor (exclusive prefix, ascending) */
    long long src_off;   /* element offset into the fp32 base */
    long long s0, s1;
    int d0, d1, ld, pad_;
    void* dst;
} e2t_pack_desc;
int e2t_pack_batch(const e2t_pack_desc* descs_dev, int ndesc, int total_blocks, const float* base, void* stream);

/* ---- a7/a9: SequenceNetwork._encode_sequences (trainers.py:821-823) and the decoder RNN ---- */
typedef struct e2t_lstm_desc {
    int S, B, H, ndir;             /* steps, utterances, hidden units per direction, 1 or 2 directions */
    int ldy;                       /* leading dim of Yext / Ydrop (>= ndir * roundup(H,8), multiple of 8) */
    float forget_bias;
    float drop_rate; unsigned long long drop_seed; const int32_t* drop_step; unsigned drop_stream;
    int rb_begin, rb_count;        /* restrict the launches to utterance blocks [rb_begin, rb_begin+rb_count) of 64 rows
                                      (rb_count 0 = all); disjoint ranges are independent and may run on different streams */
} e2t_lstm_desc;
/* Gx [S*B][ndir*H*4] bf16 (dir,unit,gate interleaved; bias folded in; ABI 5: bf16, was fp32 -- readable 32 B past the last row); WhF: e2t_pack_frag images
 * [ndir][4][UT][KB]; Yext bf16 [(S+3)*B][ldy] (time block t+1; block 0 = initial state, blocks S+1, S+2
 * all-zero slack that no kernel writes); Ydrop bf16 [S*B][ldy] or NULL; Cs (fp32) / Gs (bf16 since ABI 5: (i, j, f, o) per cell, 8 B): lane-native per-step saves,
 * S*ndir*ceil(B/16)*ceil(H/16)*64 float4 resp. x4 (layout in csrc/lstm.hip); c0 fp32 [B][ndir*H] or NULL.
 * Runs steps [step_begin, step_end). */
int e2t_lstm_seq_fwd(const e2t_lstm_desc* d, const void* Gx, const void* WhF, void* Yext, void* Ydrop, float* Cs,
                     void* Gs, const int32_t* lens, const float* c0, int step_begin, int step_end, void* stream);
/* Same result (bit for bit) as e2t_lstm_seq_fwd over steps [0,S) in ONE persistent launch: W_h stays in registers,
 * h is exchanged between CUs inside the launch (stamped values, bounded retries).  Applicable when H % 8 == 0 and the
 * layer's workgroups fit the CUs one-to-one: ceil(H/16) * ceil(B/64) * ndir of them for H <= 416, ceil(H/32) * ceil(B/32)
 * * ndir for H <= 832; returns non-zero otherwise (use e2t_lstm_seq_fwd).  hx: bf16 exchange scratch [2][ndir][4*ceil(B/64)][ceil(H/32)][64][8], zero-filled once by the
 * caller and afterwards only touched by this entry point with the same S, B, H (zero it again after an error);
 * err: int32 [1], set to 1 if a wait timed out (results then invalid). */
int e2t_lstm_seq_fwd_persistent(const e2t_lstm_desc* d, const void* Gx, const void* WhF, void* Yext, void* Ydrop, float* Cs,
                                void* Gs, const int32_t* lens, const float* c0, void* hx, int32_t* err, int num_cus,
                                void* stream);
/* BPTT over all S steps (+ pseudo-step -1 when dh0/dc0 are given).  dG bf16 [(S+1)*B][lddg] out
 * (block S is all-zero slack that no kernel writes). */
int e2t_lstm_seq_bwd(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY, int lddy,
                     const void* Gs, const float* Cs, const int32_t* lens, const float* c0, const float* dh_final,
                     const float* dc_final, float* dc_carry, float* dh0, float* dc0, void* stream);
/* Same gradients as e2t_lstm_seq_bwd (to fp32 round-off: the K = 4H sum is split in 4 quarters instead of 2 halves) in
 * ONE persistent launch, incl. the pseudo-step -1 when dh0/dc0 are given.  Applicable when H % 8 == 0 and the workgroups
 * fit the CUs one-to-one: ceil(B/16) * ndir * ceil(ceil(H/16)/4) of them for H <= 416, ceil(B/32) * ndir * ceil(H/32) for
 * H <= 800; returns non-zero otherwise.  KQ = e2t_bwd_persist_kq(H) (0: not applicable).
 * dG is handed from CU to CU inside the launch through dgx with a 1-bit stamp in bit 14 of every bf16, so the RECURRENT
 * term saturates gate gradients at |x| < 2 (1.992); the dG written for the weight / input gradients is not touched.
 * dgx: bf16 exchange scratch [2][ndir][RTD][4*KQ][64][8] with RTD = ceil(B/16) (H <= 416) or 2*ceil(B/32), zero-filled
 * once by the caller; flags: uint32 [clusters*stride] with clusters*stride = ceil(B/16)*ndir*32 (H <= 416) or
 * ceil(B/32)*ndir*128 (only the last word of each cluster's row is used: the stamps its buffers were left with), both
 * zero-filled once by the caller and afterwards only touched by this entry point with the same S, B, H (zero them again
 * after an error); err as for the forward; additionally err[8] counts the publishes in which a gate gradient reached the
 * saturating range of the exchange copy (|x| >= 2): 0 in healthy training, a warning sign for the loss scales otherwise. */
int e2t_bwd_persist_kq(int H);
int e2t_lstm_seq_bwd_persistent(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY, int lddy,
                                const void* Gs, const float* Cs, const int32_t* lens, const float* c0,
                                const float* dh_final, const float* dc_final, float* dh0, float* dc0, void* dgx,
                                uint32_t* flags, int32_t* err, int num_cus, void* stream);
/* Large hidden sizes (config 4: H = 1024): weight-stationary persistent recurrences in which the four waves of a workgroup
 * hold DIFFERENT weights and share the state through LDS (csrc/lstm_big.hip).  e2t_lstm_big_ok(H): 1 if H is supported
 * (H % 64 == 0, 448 <= H <= 1024).  Forward: same result, bit for bit, as e2t_lstm_seq_fwd over steps [0,S); needs
 * ceil(B/64) * ndir * (H/32) workgroups co-resident (<= num_cus), else returns non-zero.
 * WhG: e2t_pack_frag image of Bn[n][k] = W_h[k][n] over ALL gate columns n < 4H (gate-interleaved master), per direction:
 * [ndir][4H/16][H/32][64][8].  hx: bf16 [2][ndir][4*ceil(B/64)][H/32][64][8]; flags: uint32 [ceil(B/64)*ndir][128]; both
 * zero-filled once by the caller and afterwards only touched by these entry points (zero them again after an error). */
int e2t_lstm_big_ok(int H);
int e2t_lstm_seq_fwd_big(const e2t_lstm_desc* d, const void* Gx, const void* WhG, void* Yext, void* Ydrop, float* Cs, void* Gs,
                         const int32_t* lens, const float* c0, void* hx, uint32_t* flags, int32_t* err, int num_cus, void* stream);
/* BPTT counterpart (same gradients as e2t_lstm_seq_bwd to fp32 round-off: the K = 4H sum is split in 4 quarters); H in
 * {512, 768, 1024}; no gradient into an initial state (dh0/dc0: use the other entry points).
 * dgx: bf16 [2][ndir][4*ceil(B/64)][4H/32][64][8]; flags as for the forward, a separate array. */
int e2t_lstm_seq_bwd_big(const e2t_lstm_desc* d, const void* WhB, void* dG, int lddg, const float* dY, int lddy, const void* Gs,
                         const float* Cs, const int32_t* lens, const float* c0, const float* dh_final, const float* dc_final,
                         void* dgx, uint32_t* flags, int32_t* err, int num_cus, void* stream);
/* encoder final state -> decoder initial state (App. D2) */
int e2t_final_state(const void* Yext, int ldy, const float* Cs, const int32_t* lens, int B, int H, void* h0, int ldh0,
                    float* c0, void* stream);

/* ---- a9: embedding, softmax cross-entropy, greedy decoding ---- */
int e2t_embed_fwd(const void* emb, int ld_emb, const int32_t* tok, int row0, int M, int E, void* out, int ld_out,
                  const e2t_dropout* drop, void* stream);
int e2t_embed_bwd(const float* de, int ld_de, const int32_t* tok, int M, int E, float* demb, int ld_demb,
                  const e2t_dropout* drop, void* stream);
int e2t_softmax_ce(const float* logits, int ldl, int M, int V, const int32_t* tgt, const int32_t* lens, int rows_per_step,
                   const int32_t* ntok, float weight, float* rowloss, int32_t* pred, float* correct, void* dlogits,
                   int lddl, void* stream);
int e2t_greedy_update(const int32_t* pred, int B, int l, int Lmax, int eos, int pad, int32_t* done, int32_t* out,
                      int32_t* next_tok, void* stream);
/* ABI 7: arg-max of the step's logits (lowest index on ties) + the bookkeeping of e2t_greedy_update, one launch */
int e2t_greedy_step(const float* logits, int ldl, int B, int V, int l, int Lmax, int eos, int pad, int32_t* done, int32_t* out,
                    int32_t* next_tok, void* stream);
/* ABI 9 -- decoding FEW utterances at a time (the reference's online predictor: one utterance per call, ecog2txt/trainers.py:925-949).
 * e2t_decode_init: the start state of a greedy search in one launch -- done[b] = 0, hyp[b][:] = pad, tok0[b] = eos (the start symbol),
 * dlens[b] = Lmax -- instead of four fills.
 * e2t_greedy_head_small: one decoder step's HEAD for B <= 8 rows in one launch instead of three (vocabulary projection GEMM on a
 * 128-row tile, arg-max, row gather): logits[b][v] = bias[v] + sum_k WT[v][k] h[b][k] (bf16 operands, fp32 accumulation; WT is the
 * K-contiguous image the projection GEMM multiplies, h the decoder state of the step as the recurrence wrote it), arg-max over v
 * (lowest index on ties) and e2t_greedy_step's bookkeeping (done / out / next_tok), and -- table != NULL -- row next_tok[b] of
 * `table` ([V][row_words] 32-bit words: the decoder's input projection of every token) copied to gx_next[b][row_words] for the next
 * step.  scratch: device words, >= 2 * 64 * 8 + 1, ZERO before the first call (the kernel leaves its ticket word at zero).
 * The sums are taken in a different order than the MFMA product's: logits agree to fp32 round-off, tokens wherever the top-2
 * margin exceeds it. */
int e2t_decode_init(int32_t* done, int32_t* hyp, int32_t* tok0, int32_t* dlens, int B, int Lmax, int eos, int pad, void* stream);
int e2t_greedy_head_small(const void* h, int ldh, const void* WT, int ldw, const float* bias, int B, int V, int K, int l, int Lmax,
                          int eos, int pad, int32_t* done, int32_t* out, int32_t* next_tok, const void* table, size_t row_words,
                          void* gx_next, uint32_t* scratch, void* stream);
/* Beam search (beam_width > 1, mocha-1_word_sequence.yaml:31; temperature :82): one step for B utterances x W hypotheses
 * (rows b*W + w of logits [B*W][ldl]).  A live hypothesis continues with every token, scored
 * score + log softmax(logits / temperature); a finished one (it has emitted <EOS>) only as itself; the W best survive (ties:
 * lower beam, then lower token).  score / done / hyp [B*W][Lmax] are double-buffered by the caller (in -> out); rowmap[r] =
 * row of hypothesis r's parent; next_tok (may be NULL) = input tokens of the next step.  W <= 16. */
int e2t_beam_step(const float* logits, int ldl, int B, int W, int V, float temperature, int l, int Lmax, int eos, int pad,
                  const float* score_in, const int32_t* done_in, const int32_t* hyp_in, float* score_out, int32_t* done_out,
                  int32_t* hyp_out, int32_t* rowmap, int32_t* next_tok, void* stream);
/* the decoder state (one direction) of the surviving hypotheses: row r of Yblk (bf16 [rows][ldy], one time block of Yext) and of
 * the step's lane-native cell save Cs_step <- those of row rowmap[r]; tmp_h bf16 [rows][roundup(H,8)], tmp_c fp32 [rows][H] */
int e2t_beam_reorder(void* Yblk, int ldy, float* Cs_step, int rows, int H, const int32_t* rowmap, void* tmp_h, float* tmp_c, void* stream);
/* ---- a8: Gaussian encoder-target head ---- */
int e2t_mse(const float* P, int ldp, const float* At, int M, int K, const int32_t* lens, int rows_per_step,
            const int32_t* nval, float weight, float* rowloss, void* dP, int lddp, void* stream);

/* ---- a10: Adam + EMA (EMA_decay, mocha-1_word_sequence.yaml:5) ---- */
typedef struct e2t_adam_hyper {
    float lr, beta1, beta2, eps, ema_decay, grad_scale;
    int step_offset;       /* the update uses t = *step + step_offset (1: a range updated before e2t_inc_step has run, while
                              other kernels of the same train step still key their dropout masks on *step) */
    const int32_t* skip_if_nonzero;   /* device word or NULL: when *skip_if_nonzero != 0 the launch leaves p, m, v, ema untouched
                              (the err word of the persistent recurrences: a step whose in-kernel wait timed out has
                              invalid gradients and must not reach the weights) */
} e2t_adam_hyper;
/* *step += 1 unless skip_if_nonzero (device word or NULL) is non-zero */
int e2t_inc_step(int32_t* step, const int32_t* skip_if_nonzero, void* stream);
int e2t_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, size_t n, const int32_t* step,
                      const e2t_adam_hyper* h /* host pointer */, void* stream);

/* ABI 8 -- the optimiser update and the operand re-pack of a parameter matrix in ONE pass (the HBM-bound tail of the train step:
 * e2t_adam_ema_step writes the masters and e2t_pack_batch reads them again once per image).  A descriptor names a sub-matrix of the
 * flat fp32 buffers in MASTER orientation (R rows of C contiguous elements, row stride s0, origin src_off) and up to three bf16
 * images of it; a 256-thread workgroup owns one 64 x 64 tile: it loads p (and, with an update, g, m, v, ema), applies exactly
 * e2t_adam_ema_step's arithmetic, writes p, m, v, ema back, and emits the tile's share of every image from LDS -- the same bits
 * e2t_pack_batch would have produced from the updated masters.  src_off, s0, C multiples of 4; image kinds:
 *   1  dst[r*ld + c] = bf16(x[r][c])                      (ld multiple of 4, dst 8-B aligned)
 *   2  dst[c*ld + r] = bf16(x[r][c])                      (ld multiple of 4, dst 8-B aligned)
 *   3  MFMA fragment image of Bn[n = r][k = c]            (ld = ceil(C/32) k-blocks; e2t_pack_desc kind 1)
 *   4  MFMA fragment image of Bn[n = c][k = r]            (ld = ceil(R/32); e2t_pack_desc kind 5)
 *   5  the four per-gate fragment images of a gate-interleaved matrix, Bn_g[n][k = r] = x[r][n*4 + g], contiguous at dst
 *      [4][ceil(C/64)][ld]                                 (ld = ceil(R/32); C multiple of 4; e2t_pack_desc kind 6)
 * Descriptors of one launch must not overlap (every element is updated by exactly one workgroup). */
#define E2T_TILE_IMG_MAX 3
typedef struct e2t_tile_img { void* dst; int kind; int ld; } e2t_tile_img;
typedef struct e2t_tile_desc {
    int first_block;     /* first workgroup of this descriptor (exclusive prefix, ascending); it owns ceil(R/64)*ceil(C/64) */
    int R, C, nimg;
    long long src_off;   /* element offset of the sub-matrix's origin in the flat buffers */
    long long s0;        /* row stride in elements */
    const float* gslab;  /* NULL: the gradient is g[src_off + r*s0 + c]; else it is the sum over s < gsplits, in that order, of
                            gslab[s*gstride + r*s0 + c] -- the slabs a K-major product left (E2T_GEMM_KEEP_SLABS) */
    long long gstride;
    int gsplits, pad_;
    e2t_tile_img img[E2T_TILE_IMG_MAX];
} e2t_tile_desc;
/* h == NULL: images only, from `p` (any flat fp32 buffer: the masters or the EMA shadows); g, m, v, ema are then ignored.
 * h != NULL: update + images; *h->skip_if_nonzero != 0 leaves everything untouched (the images stay those of the old masters). */
int e2t_adam_pack_batch(const e2t_tile_desc* descs_dev, int ndesc, int total_blocks, float* p, const float* g, float* m, float* v,
                        float* ema, const int32_t* step, const e2t_adam_hyper* h /* host pointer or NULL */, void* stream);

/* ---- e1: the exchange step of the utterance-sharded data-parallel path (new: the reference trains on one device,
 *      trainers.py:131).  RCCL over xGMI, one communicator per process / GPU.  Collectives run on a stream the
 *      communicator owns, ordered AFTER everything enqueued so far on `after_stream`; *ticket (may be NULL) names the
 *      collective for e2t_comm_wait.  At most 32 collectives may be outstanding (un-waited) at a time. ---- */
#define E2T_COMM_ID_BYTES 128
typedef struct e2t_comm e2t_comm;
int e2t_comm_unique_id(void* id128 /* host, E2T_COMM_ID_BYTES out: made on one rank, handed to all by the caller */);
int e2t_comm_init(e2t_comm** out, int rank, int nranks, const void* id128 /* host */, int device);
int e2t_comm_destroy(e2t_comm* c);
int e2t_comm_rank(const e2t_comm* c);
int e2t_comm_size(const e2t_comm* c);
/* ABI 6: the communicator's stream waits for the work enqueued so far on `after_stream`; nothing is issued.  A captured step calls it
 * once on the capture's origin stream, so that the communicator's stream enters the capture from there (see csrc/comm.hip). */
int e2t_comm_order_after(e2t_comm* c, void* after_stream);
/* in-place sum over ranks of buf[0..n) (a contiguous range of the flat gradient buffer) */
int e2t_comm_allreduce_f32(e2t_comm* c, float* buf, size_t n, void* after_stream, int* ticket);
int e2t_comm_allreduce_i32(e2t_comm* c, int32_t* buf, size_t n, void* after_stream, int* ticket);   /* token ids / counts */
/* ABI 8: element-wise MAXIMUM over the ranks (the persistent recurrences' error word: idempotent from step to step, keeps the code) */
int e2t_comm_allreduce_max_i32(e2t_comm* c, int32_t* buf, size_t n, void* after_stream, int* ticket);
int e2t_comm_broadcast(e2t_comm* c, void* buf, size_t bytes, int root, void* after_stream, int* ticket);
/* `stream` waits for collective `ticket` (-1: for every collective issued so far) */
int e2t_comm_wait(e2t_comm* c, int ticket, void* stream);

/* ---- host helper: CRC-32C (Castagnoli) of a HOST buffer, for the TFRecord / checkpoint framing of rows a3, f1, f2
 *      (SSE4.2 crc32 instruction); crc = 0 starts a new checksum, or pass a previous result to continue it ---- */
uint32_t e2t_crc32c(const void* data, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif
