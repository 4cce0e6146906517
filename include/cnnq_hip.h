/*
 * cnnq_hip.h - C ABI of libcnnq_hip.so: MI355X (gfx950) kernels for the GEMMLOWP-style
 * per-channel quantize / clip / dequantize hot path of submission2019/cnn-quantization.
 *
 * This is the drop-in boundary.  The reference's only native module is
 *     kernels/int_quantization.cpp:10-12  ->  float2gemmlowp(in, range, offset, num_bits,
 *                                             int_exp, enforce_true_zero, noise)
 *     kernels/gemmlowp.cu:8-45            (kernel + host wrapper)
 * which cnnq_pt_qdq replaces one for one.  Everything else the reference does on this path
 * is a chain of aten ops inside pytorch_quantizer/quantization/qtypes/int_quantizer.py
 * (abbreviated iq.py) and pytorch_quantizer/quantization/inference/
 * statistic_manager_perchannel.py (smpc.py); each entry point below names the lines it fuses.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - activations are fp32 NCHW, addressed as x[N][C][HW] (HW = H*W contiguous floats);
 *     weights [OFM][IFM*K*K] are the same layout with N = 1, C = OFM;
 *     per-sample statistics of any tensor are the same layout with N = 1, C = samples;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *     enqueued, never synchronised; nothing is allocated, every function is re-entrant; the
 *     library keeps no state and reads ONE environment variable, once per process:
 *     CNNQ_NT_BYTES (default 402653184) - tensors above it are streamed with non-temporal loads
 *     by the read-only passes (a property of the part's Infinity Cache; 0: always).  The kernel
 *     sweep knobs of the development builds (-DCNNQ_DEV_KNOBS) do not exist in this library;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     CNNQ_E* for rejected arguments (nothing is enqueued then).
 */
#ifndef CNNQ_HIP_H
#define CNNQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNNQ_EINVAL (-1) /* null pointer, non-positive size, unsupported option */
#define CNNQ_ERANGE (-2) /* C*HW does not fit the 31-bit plane index */
#define CNNQ_ENOTSUP (-3) /* this entry point has no kernel for the shape / alignment: use the general one */

/* Rows of the per-channel fp32 statistics table `stats[CNNQ_NSTAT][C]`
 * (iq.py:530-555 and smpc.py:54-79 name the same quantities). */
enum {
    CNNQ_STAT_MIN = 0,
    CNNQ_STAT_MAX = 1,
    CNNQ_STAT_MEAN = 2,
    CNNQ_STAT_STD = 3,     /* unbiased (n-1) */
    CNNQ_STAT_B = 4,       /* mean |x - mean| */
    CNNQ_STAT_KURT = 5,    /* mean(((x-mean)/std)^4) - 3 */
    CNNQ_STAT_STD_POS = 6, /* unbiased std of relu(x) */
    CNNQ_NSTAT = 7
};

/* Rows of the fp64 moment records `mom[CNNQ_NMOM][C]` that ranks exchange (all_gather) and
 * cnnq_pc_combine merges: partial results are mergeable, statistics are not. */
enum {
    CNNQ_MOM_MIN = 0,
    CNNQ_MOM_MAX = 1,
    CNNQ_MOM_SUM = 2,
    CNNQ_MOM_SUMSQ = 3,
    CNNQ_MOM_COUNT = 4,
    CNNQ_MOM_SUM_RELU = 5,
    CNNQ_MOM_SUMSQ_RELU = 6,
    CNNQ_NMOM = 7
};
/* Rows of the second-pass records `dev[CNNQ_NDEV][C]` (need the global mean / std). */
enum { CNNQ_DEV_ABS = 0, CNNQ_DEV_Z4 = 1, CNNQ_NDEV = 2 };

/* Rows of the quantisation parameter table `qp[CNNQ_NQP][C]` (iq.py:559-572). */
enum { CNNQ_QP_SCALE = 0, CNNQ_QP_ZP = 1, CNNQ_QP_QMAX = 2, CNNQ_NQP = 3 };
/* Rows of the diagnostic table `diag[CNNQ_NDIAG][C]` written by cnnq_pc_params. */
enum { CNNQ_DIAG_BITS = 0, CNNQ_DIAG_ALPHA = 1, CNNQ_DIAG_DELTA = 2, CNNQ_DIAG_OFFSET = 3, CNNQ_NDIAG = 4 };

/* Rows of the mid-tread parameter table `mt[CNNQ_NMT][C]` (row WSTART: the first integer code of the histogram
 * window, the same value in every column) and the size of its code histogram in uint64 words,
 * CNNQ_MT_HIST_WORDS(C): CNNQ_MT_HIST_BINS integer-code bins centred on 0, 2 out-of-range bins, 2*C counters for
 * codes clamped to a non-integer c_min[c] / c_max[c], then CNNQ_MT_HIST_REPLICAS copies of the
 * CNNQ_MT_HIST_WINDOW bins of the codes WSTART .. WSTART + WINDOW - 1 (workgroups spread their updates over the
 * copies; cnnq_midtread_entropy adds them up), and one flag word: non-zero when a count went to the
 * CNNQ_MT_HIST_BINS + 2 global bins, i.e. a code fell outside the window. */
enum {
    CNNQ_MT_DELTA = 0, CNNQ_MT_CMIN = 1, CNNQ_MT_CMAX = 2, CNNQ_MT_OMEGA = 3, CNNQ_MT_ALPHA = 4, CNNQ_MT_WSTART = 5,
    CNNQ_NMT = 6
};
#define CNNQ_MT_HIST_BINS 131072
#define CNNQ_MT_HIST_WINDOW 128
#define CNNQ_MT_HIST_REPLICAS 256
#define CNNQ_MT_HIST_WORDS(C) (CNNQ_MT_HIST_BINS + 2 + 2 * (C) + CNNQ_MT_HIST_REPLICAS * CNNQ_MT_HIST_WINDOW + 1)

/* library / build identification ("cnnq-hip <version> gfx950") */
const char* cnnq_version(void);

/* Number of partial groups G that the streaming statistics kernels emit per channel for a
 * tensor of this geometry; the caller sizes `part` as [G][CNNQ_NMOM][C] doubles (and
 * [G][CNNQ_NDEV][C] for the second pass).  `aligned16` = (x % 16 == 0).  Returns G > 0 or
 * a negative CNNQ_E*. */
int cnnq_pc_groups(int64_t N, int64_t C, int64_t HW, int aligned16);

/* Diagnostics (host only, nothing is enqueued): the launch plan the streaming kernels use for a
 * tensor - out = {vec, A, J, mode, nb, w, k, ncb, S, threads per workgroup, G, 0}: loads of `vec`
 * floats, A parameter sets per load (4 = a float4 may straddle two channels), J loads per lane per
 * sample; mode 1: a workgroup owns `w` loads of ONE channel (`nb` workgroups per channel), mode 2: `k`
 * whole channels; ncb column blocks x S batch splits workgroups; fine != 0: the short-workgroup
 * geometry of the table-driven elementwise passes.  tests/test_plan_cpu.py checks on the CPU that
 * these plans cover every element exactly once. */
int cnnq_plan_describe(int64_t N, int64_t C, int64_t HW, int aligned16, int fine, int32_t out[12]);

/* Pass A over x[N][C][HW]: per channel min, max, sum, sum of squares, count (and, when
 * want_relu, sum and sum of squares of relu(x)) -> part[G][CNNQ_NMOM][C].  One coalesced
 * read of x, no transposed copy.  Fuses the reductions of iq.py:534-550 (max/min/mean/std),
 * smpc.py:54-55,64-66 (mean, std, std_pos) that the reference runs as one full pass each
 * after a transpose+copy. */
int cnnq_pc_moments(const float* x, int64_t N, int64_t C, int64_t HW, int want_relu, double* part,
                    void* stream);

/* Merge G moment records per channel (G = groups of one tensor, or G = world size after an
 * all_gather of per-rank records) into mom[CNNQ_NMOM][C] and, when `stats` is not NULL, write
 * the whole table stats[CNNQ_NSTAT][C]: rows MIN, MAX, MEAN, STD, STD_POS (zero unless the records carry relu
 * sums), and zeros in rows B and KURT (cnnq_pc_combine_dev fills them after pass B).
 * Fixed merge order (a function of G and C) -> deterministic.  `mom` may be NULL. */
int cnnq_pc_combine(const double* part, int G, int64_t C, int has_relu, double* mom, float* stats,
                    void* stream);

/* Pass B: per channel sum |x - mean[c]| and (want_kurt) sum ((x-mean[c])/std[c])^4, reading
 * mean/std from stats rows MEAN/STD -> part2[G][CNNQ_NDEV][C].  iq.py:548 ('b'),
 * smpc.py:58-61 (kurtosis, b). */
int cnnq_pc_absdev(const float* x, int64_t N, int64_t C, int64_t HW, const float* stats, int want_kurt,
                   double* part2, void* stream);

/* All statistics of cnnq_pc_stats from ONE launch that reads x once (round 5; csrc/cnnq_stats1.hip.h): a workgroup keeps its
 * tile of x in registers (and LDS) across pass A and pass B, and the workgroups of a channel exchange their partial sums twice
 * through the slot region of the exchange workspace (cnnq_group_ws_alloc) - 4 instead of 8 bytes per element, one launch
 * instead of three; the same per-element arithmetic and final formulas as the chain, a different (fixed) order of the fp64
 * additions (round 6: nobody waits for the second exchange - the workgroup that arrives LAST folds it, writes the channel's row
 * and re-arms the slots; the others leave after publishing their words).  Flat-tile geometries (H*W % 4 == 0, H*W / 4 >= 128 and not a multiple of 256, 2 .. 512 tiles per channel), and
 * the row-piece tiles of cnnq_pc_minmax_qdq_group for shorter rows (k_stats_group).  CNNQ_ENOTSUP where the chain is faster:
 * more than 256 tiles per channel, row-piece geometries whose channels straddle the 16-byte loads (7x7) above 8 MB (flags
 * bit 3 lifts these two rules: tests), and where there is no 16-byte tiling at all.  (Round 6: every one-channel-per-lane
 * row-piece shape with a plan is taken - at a 64-sample shard the one launch of 20-45 us beats three launches and two merges.)
 * flags: 0 (tests: 1 = skip the waits and recompute).  cnnq_pc_stats_auto: this when it applies,
 * else cnnq_pc_stats (ws as there). */
int cnnq_pc_stats_single(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* gws,
                         size_t gws_bytes, double* mom, float* stats, unsigned flags, void* stream);
int cnnq_pc_stats_auto(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* ws, void* gws,
                       size_t gws_bytes, double* mom, float* stats, void* stream);
/* The route cnnq_pc_stats_single (flags as there) / cnnq_pc_stats_auto (flags 0) take for a geometry, without launching anything:
 * 1 - the flat-tile single launch, 2 - the row-piece single launch, 0 - the three-launch chain (cnnq_pc_stats_single would answer
 * CNNQ_ENOTSUP).  aligned16: x is 16-byte aligned; gws_bytes: the exchange workspace the caller holds (cnnq_pc_group_workspace). */
int cnnq_pc_stats_route(int64_t N, int64_t C, int64_t HW, int aligned16, size_t gws_bytes, unsigned flags);

/* Merge pass-B records; divides by the COUNT row of `mom` and writes stats rows B (and KURT).
 * `dev_out` (merged [CNNQ_NDEV][C] sums, for the cross-rank exchange) may be NULL, and
 * `stats` may be NULL when only the merged sums are wanted. */
int cnnq_pc_combine_dev(const double* part2, int G, int64_t C, const double* mom, int want_kurt,
                        double* dev_out, float* stats, void* stream);

/* All per-channel statistics of one tensor behind one call and one caller workspace `ws` of
 * cnnq_pc_stats_workspace(...) bytes (8-byte aligned): pass A -> pass B with the merge of the pass-A records fused
 * into its prologue -> one final merge of both passes (three launches for the seven statistics of smpc.py:45-79,
 * two when neither b nor kurtosis is wanted).  stats[CNNQ_NSTAT][C] is written completely (rows not requested are
 * zero); mom[CNNQ_NMOM][C] (merged moment record) may be NULL.  Single process: the cross-rank exchange needs the
 * separate calls above. */
size_t cnnq_pc_stats_workspace(int64_t N, int64_t C, int64_t HW, int aligned16);
int cnnq_pc_stats(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* ws,
                  double* mom, float* stats, void* stream);

/* Per-channel statistics -> quantisation parameters, entirely on the device (the reference
 * takes >= 6 host round trips here: iq.py:248,285-288,355,405).  One workgroup.
 *   clip: 0 = min/max range (iq.py:409-424), 1 = ACIQ laplace (iq.py:227-253),
 *         2 = ACIQ gaus (iq.py:255-264), 3 = p*std (iq.py:266-275, p = pstd);
 *   positive = force_positive || half_range (iq.py:289,411);
 *   bit_alloc != 0 (and num_bits <= 4): per-channel bit allocation from the prior statistic
 *         (prior_is_b ? B : STD) by the fixed-target iteration of iq.py:381-407 with
 *         `target` bits and round (1) or ceil (0);
 *   alpha2DeltaOffset (iq.py:284-300), delta = (offset + range) - offset (iq.py:351,443),
 *   scale / zero-point (iq.py:559-572).
 * Reads stats[CNNQ_NSTAT][C] rows as needed, writes qp[CNNQ_NQP][C] and (if not NULL)
 * diag[CNNQ_NDIAG][C]. */
typedef struct cnnq_params_cfg {
    int32_t num_bits;
    int32_t positive;
    int32_t clip;
    float pstd;
    int32_t bit_alloc;
    int32_t prior_is_b;
    double target;
    int32_t round_mode;
    int32_t direct_range; /* per-tensor clipping branch (iq.py:353-357): delta = range itself */
} cnnq_params_cfg;
int cnnq_pc_params(const float* stats, int64_t C, const cnnq_params_cfg* cfg /* host */, float* qp,
                   float* diag, void* stream);

/* Weights (iq.py:453-476): identical parameter derivation with the range always min/max and
 * the bit allocation prior always STD; provided as cnnq_pc_params with clip = 0. */

/* The core: y = dequant(quant(x)) on native NCHW with per-channel scale / zero point / qmax
 * (iq.py:573-592 - div, add, clamp, round, sub, mul as ONE pass; the four transpose copies of
 * iq.py:427,450,534 disappear).  IEEE division, separate roundings (no FMA), clamp before
 * round, round half to even: integer codes are bit-exact with the reference's.
 * Non-temporal loads and stores (x is read for the last time, y is never re-read by this path);
 * launched as many short workgroups in address order, which is what read+write streaming on
 * MI355X wants; reverse != 0 walks the tensor in descending address order (use it when the
 * previous pass over x went ascending: its tail is still in the 256 MB Infinity Cache).
 * `codes` (optional, may be NULL): the integer codes as one byte per element.
 * `hist` (optional, may be NULL): 256 uint64 bins, ZEROED BY THE CALLER, to which the code
 * histogram of the whole tensor is added (integer atomics -> deterministic); with cnnq_entropy
 * this replaces the full sort of torch.unique in utils/entropy.py:6-17 (iq.py:586-587). */
int cnnq_pc_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                uint8_t* codes, uint64_t* hist, int reverse, void* stream);

/* Packed int4 as the STORED activation format (beyond the reference, which only
 * simulates): cnnq_pc_quantize_pack4 writes the integer codes of a <= 4-bit quantization two per
 * byte (even element in the low nibble, N*C*HW/2 bytes) - 4 B read + 0.5 B written per element;
 * cnnq_pc_dequantize_pack4 turns them back into exactly the floats cnnq_pc_qdq would have produced.
 * Requires HW % 4 == 0, 16-byte aligned x / y, qmax <= 15 in every channel of qp. */
int cnnq_pc_quantize_pack4(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                           void* stream);
int cnnq_pc_dequantize_pack4(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                             void* stream);
/* The same with one byte per code (quantizations of up to 8 bits, e.g. the first layer the reference keeps at
 * 8 bit, iqm.py:551-555): 4 + 1 bytes per element; `codes` 4-byte aligned, H*W % 4 == 0. */
int cnnq_pc_quantize_u8(const float* x, uint8_t* codes, int64_t N, int64_t C, int64_t HW, const float* qp,
                        void* stream);
int cnnq_pc_dequantize_u8(const uint8_t* codes, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                          void* stream);

/* Bit allocation as the STORED format (the per-channel qmax = 2**bit_alloc - 1 of iq.py:563-564,
 * 582-584 made real): channel c stores bits[c] in 0..8 bits per code (bits = row CNNQ_DIAG_BITS of cnnq_pc_params'
 * diag), i.e. sum(bits)/8 bytes per spatial position.  Row (n, c) = H*W codes as a little-endian bit stream,
 * padded to 4 bytes, at byte n * rowoff[C] + rowoff[c]; cnnq_pc_packed_layout fills rowoff[C + 1] (uint32, device)
 * from bits; the packed buffer needs N * rowoff[C] bytes.  A 0-bit channel stores nothing.  The round trip
 * reproduces the fused Q/DQ (cnnq_pc_qdq with the same qp) bit for bit. */
int cnnq_pc_packed_layout(const float* bits, int64_t C, int64_t HW, uint32_t* rowoff, void* stream);
int cnnq_pc_quantize_packed(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                            const float* bits, const uint32_t* rowoff, void* stream);
/* The same pass with its kernel form spelled out (tests, measurements): form 0 = the library's choice, 1 = the general
 * kernel (any geometry), 2 = the lean kernel of round 3 (one channel per wave, scalar parameters; CNNQ_ENOTSUP for rows
 * of fewer than 8 elements, or whole-float4 rows (H*W % 4 == 0) of an x that is not 16-byte aligned).  Every form writes
 * the same bytes / the same floats; the load direction's lean form also needs a 4-byte aligned packed buffer.  Form 3: one-shot
 * workgroups in address order of the fp32 tensor, per-lane channel parameters - load direction (round 4; the library's choice
 * when it applies): CNNQ_ENOTSUP unless y is 16-byte aligned, the stream 4-byte aligned and the tensor below 2^32 elements;
 * store direction (round 5, k_pack_flat): CNNQ_ENOTSUP unless H*W % 4 == 0, x is 16-byte and the stream 4-byte aligned. */
int cnnq_pc_quantize_packed_form(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                                 const float* bits, const uint32_t* rowoff, int form, void* stream);
int cnnq_pc_dequantize_packed_form(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                                   const float* bits, const uint32_t* rowoff, int form, void* stream);
int cnnq_pc_dequantize_packed(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                              const float* bits, const uint32_t* rowoff, void* stream);

/* Config 2 (dynamic per-channel min/max, no clipping, uniform bit width: iq.py:409-451 with
 * bit allocation off) as three launches, two of them streaming:
 *   cnnq_pc_minmax         exact per-channel {min, max} partials pmm[G][2][C] (G = cnnq_pc_groups;
 *                          every entry written exactly once: no atomics, no initialisation);
 *   cnnq_pc_minmax_params  reduces the G pairs and derives qp[CNNQ_NQP][C] (iq.py:559-572);
 *   cnnq_pc_qdq            the fused Q/DQ (reverse = 1: descending addresses).
 *   cnnq_pc_minmax_qdq     all three in one call; pmm and qp are caller workspaces (qp is also the
 *                          parameter table used, for inspection).
 *   cnnq_pc_minmax_reduce  pmm[G][2][C] -> out[2][C]: a rank's local extrema.  Multi-GPU config 2:
 *                          minmax -> reduce -> all_gather of out over the ranks -> minmax_params with
 *                          the gathered [W][2][C] as pmm and G = W -> qdq (exact, so any world size
 *                          gives the bit-identical result of one GPU holding the whole batch). */
int cnnq_pc_minmax(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, void* stream);
/* Channel-slice views of a wider NCHW tensor (x[:, c0:c1]: pass x + c0*HW, C = c1 - c0 and the parent's
 * sample stride C_total*HW; 0 = contiguous): the same kernels, the plane size only ever served as that
 * stride.  For a (pointer, stride) pair `aligned16` of cnnq_pc_groups is (x % 16 == 0 && stride % 4 == 0).
 * Used to quantize concatenated outputs in place and to pipeline the multi-GPU exchange (half the
 * channels' statistics travel while the other half is being read). */
int cnnq_pc_minmax_strided(const float* x, int64_t N, int64_t C, int64_t HW, int64_t sample_stride, float* pmm,
                           void* stream);
int cnnq_pc_qdq_strided(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int64_t sample_stride,
                        const float* qp, uint8_t* codes, uint64_t* hist, int reverse, void* stream);
int cnnq_pc_minmax_reduce(const float* pmm, int G, int64_t C, float* out, void* stream);
int cnnq_pc_minmax_params(const float* pmm, int G, int64_t C, int num_bits, int positive, float* qp, void* stream);
int cnnq_pc_minmax_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                       float* pmm, float* qp, uint8_t* codes, uint64_t* hist, void* stream);

/* Config 2 in ONE launch that reads x ONCE (8 instead of 12 bytes per element, one launch boundary instead of
 * three): the same arithmetic as cnnq_pc_minmax_qdq (iq.py:409-451 + 557-603, bit allocation off), the same bits.
 * A workgroup owns whole channels for the whole batch and keeps them in registers between the statistics and the
 * Q/DQ; nothing is exchanged between workgroups and no workspace is needed.
 *   qp   out: qp[CNNQ_NQP][C].   mm: optional out [2][C] = the per-channel min and max.
 * Applies when one channel's batch population fits a workgroup's registers (N*H*W up to ~50 K elements per
 * channel: e.g. 64 x 28x28) with 16-byte aligned x and y and H*W % 4 == 0 (or a straddling layout such as 7x7
 * with C*H*W % 4 == 0); returns CNNQ_ENOTSUP otherwise (nothing enqueued: call cnnq_pc_minmax_qdq).
 * cnnq_pc_resident_describe fills out[8] = {A, threads per workgroup, K loads per lane, channels per workgroup,
 * float4 columns, row lanes, workgroups, 0} or returns CNNQ_ENOTSUP (tests, tools). */
int cnnq_pc_resident_describe(int64_t N, int64_t C, int64_t HW, int32_t out[8]);
int cnnq_pc_minmax_qdq_resident(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                                float* qp, float* mm, void* stream);

/* Config 2 in ONE launch and ONE read of x for tensors whose channels span several workgroups (csrc/cnnq_group.hip.h):
 * every workgroup keeps its tile of x in registers; the workgroups that hold pieces of the same channels exchange
 * their {min, max} pairs through `ws` (round 4, the slot meeting: a member's pair is stored once into its zero-at-rest slot
 * and IS its arrival, the members poll the group's slots, the last one to leave zeroes them; flags bit 5: write-through
 * stores and one arrival counter per channel group as in round 2; either way a bounded wait that falls back to recomputing
 * the extrema from x - never a deadlock, never different bits).
 *   x, y must not overlap (both are read / written through __restrict__ pointers, and a workgroup whose bounded wait
 *        expires re-reads x after other workgroups may have stored y); the same holds for every single-launch entry point.
 *   ws   from cnnq_group_ws_alloc(bytes >= cnnq_pc_group_workspace(N, C, HW)): fine-grained (uncached) device
 *        memory, zeroed once (the kernel re-arms it), so that what a workgroup reads never depends on the state of
 *        a per-XCD L2; one workspace must not be used by two launches that can run concurrently.
 *        Word 0 is a status word (cnnq_group_ws_status copies it to the host, synchronising): bit 0 is set when
 *        a wait timed out (results are unaffected; while it is set, later waits give up after 0.5 ms instead of 20 -
 *        cnnq_group_ws_status_clear lowers it).
 *   qp   out: qp[CNNQ_NQP][C].   mm: optional out [2][C] = the per-channel min and max.
 *   flags  bit 0: take the recompute path unconditionally (tests).
 *          bit 1: quantize every channel through the hardware divide (tests).  By default a channel whose extrema are finite, at most 2^70 in
 *          magnitude, with a scale of at most 2^30, takes the correctly rounded quotient from the channel's reciprocal
 *          (two fma corrections, csrc/cnnq_qdq.hip.h qdq1_fast): the same bits, half the arithmetic.
 *          bit 5 (32): the counter meeting of round 2 instead of the slot meeting (tests, A/B).
 *        A slot of the slot meeting must be ZERO AT REST - a stale non-zero word would be folded into a channel's extrema
 *        as if a member had stored it: hand the kernel only workspaces from cnnq_group_ws_alloc (zeroed at their full size),
 *        never one a launch was aborted on (device reset), and never replay a captured graph concurrently with eager
 *        launches on the same workspace; cnnq_group_ws_at_rest checks it.  The slot region sits between the counter lines
 *        and the pair blocks (2 MB, since round 4: re-query cnnq_pc_group_workspace, do not hard-code a size).
 * Same shape / alignment conditions and CNNQ_ENOTSUP convention as cnnq_pc_minmax_qdq_resident.
 * cnnq_pc_group_describe: out[8] = {A, K, mode, S, column blocks, workgroups per group, groups, workgroups}. */
size_t cnnq_pc_group_workspace(int64_t N, int64_t C, int64_t HW);
int cnnq_group_ws_alloc(size_t bytes, void** ws);   /* allocates + zeroes, synchronises the device */
int cnnq_group_ws_free(void* ws);
int cnnq_group_ws_status(const void* ws, uint32_t* status_host);
int cnnq_group_ws_status_clear(void* ws);
/* tests: counts the 32-bit words that are not zero in the regions every launch leaves zero (everything below the pair
 * blocks but the status word and the epoch-versioned corner of the header that cnnq_pt_minmax_qdq_fused writes before it
 * reads: counter lines, the slot meeting's slots); synchronises. */
int cnnq_group_ws_at_rest(const void* ws, uint64_t* nonzero_words_host);
int cnnq_pc_group_describe(int64_t N, int64_t C, int64_t HW, int32_t out[8]);
int cnnq_pc_minmax_qdq_group(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                             void* ws, float* qp, float* mm, unsigned flags, void* stream);

/* The two halves of config 2 around the cross-rank exchange (multi-GPU: every rank holds a batch shard), one call
 * each: cnnq_pc_minmax_local = cnnq_pc_minmax + cnnq_pc_minmax_reduce -> local[2][C] (two launches);
 * cnnq_pc_minmax_local_auto does the same in ONE launch when the geometry has a group plan and `gws` (the exchange
 * workspace of cnnq_pc_minmax_qdq_group, gws_bytes >= cnnq_pc_group_workspace(...); may be NULL) is given: the last
 * workgroup of a channel group to arrive folds the group's pairs - nobody waits.  After the all_gather of the W ranks'
 * records, cnnq_pc_gathered_qdq is ONE launch as well: every workgroup of the fused Q/DQ derives scale / zero point
 * of its channels from gathered[W][2][C] (the arithmetic of cnnq_pc_minmax_params: same bits on every rank) and the
 * first batch split writes them to qp[CNNQ_NQP][C] (may be NULL). */
int cnnq_pc_minmax_local(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, float* local, void* stream);
int cnnq_pc_minmax_local_auto(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, void* gws, size_t gws_bytes,
                              float* local, void* stream);
int cnnq_pc_gathered_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* gathered, int W,
                         int num_bits, int positive, float* qp, void* stream);

/* Config 2 behind one call.  `ws`: caller workspace of cnnq_pc_minmax_qdq_workspace(N, C, HW) bytes (4-byte aligned;
 * floats qp[CNNQ_NQP][C], mm[2][C], pmm[G][2][C]).  allow_single_launch != 0: the resident single launch when the
 * shape has one, else - when `gws` (a zeroed-once group workspace of gws_bytes >= cnnq_pc_group_workspace(...), see
 * cnnq_pc_minmax_qdq_group) is given - the group-exchange single launch; mm is valid after either.  Otherwise,
 * and for shapes neither supports, the three-launch chain. */
size_t cnnq_pc_minmax_qdq_workspace(int64_t N, int64_t C, int64_t HW);
int cnnq_pc_minmax_qdq_auto(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                            float* ws, void* gws, size_t gws_bytes, int allow_single_launch, void* stream);

/* Config 2 in ONE launch with the outputs that otherwise need the three-launch chain (round 3): y, plus optionally
 * `codes` (uint8, one per element, <= 8 bits; 4-byte aligned) and / or the code histogram `hist_rep`
 * (cnnq_hist_replica_bytes() bytes, zeroed ONCE by the caller: cnnq_entropy_replicas folds the replica tables into the
 * Shannon entropy of utils/entropy.py:6-17 - iq.py:586-587 - and leaves them zero) - or, with y == codes == hist_rep ==
 * NULL, `packed`: the 4-bit codes two per byte INSTEAD of y (<= 4 bits; the layout of cnnq_pc_quantize_pack4, decoded by
 * cnnq_pc_dequantize_pack4 with the qp this call writes): 4.5 bytes per element straight from x.
 * gws / gws_bytes: the exchange workspace of cnnq_pc_minmax_qdq_group (may be NULL: whole-channel shapes only).
 * qp[CNNQ_NQP][C] is written; mm[2][C] (may be NULL) receives the extrema.  CNNQ_ENOTSUP: the shape has no
 * single-launch kernel (use cnnq_pc_minmax_qdq and friends). */
size_t cnnq_hist_replica_bytes(void);
int cnnq_pc_minmax_qdq_single(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                              void* gws, size_t gws_bytes, float* qp, float* mm, uint8_t* codes, uint64_t* hist_rep,
                              uint8_t* packed, void* stream);
int cnnq_entropy_replicas(uint64_t* hist_rep, float* out, void* stream);
/* Round 6: the entropies of a whole forward's tensors in ONE launch at its end (the reference logs the value, nothing consumes
 * it mid-forward: int_quantizer.py:445,179,217): n sets of replica tables back to back (cnnq_hist_replica_bytes() each, every
 * tensor's launch counting into its own set) -> out[n]; all tables are zero afterwards. */
int cnnq_entropy_replicas_batch(uint64_t* hist_rep, int n, float* out, void* stream);

/* Config 2 in ONE launch and ONE read of x when the batch is sharded over `world` GPUs of one node
 * (csrc/cnnq_xrank.hip.h; opt-in in the Python host - CNNQ_XRANK=1 / auto - and only after it reproduced the collective
 * form's bits at first use; the collective form is cnnq_pc_minmax_local_auto -> all_gather -> cnnq_pc_gathered_qdq, which
 * reads x twice).  The reference has no counterpart (its DataParallel replicas use their own sub-batch's range,
 * inference_sim.py:196-200); this reproduces the single-GPU result of int_quantizer.py:409-451,557-603 on the global batch.
 *   windows  device array [world] of pointers: entry r is rank r's window (cnnq_xrank_alloc on rank r, opened here with
 *            cnnq_xrank_open from its hipIpc handle; the own window at [rank]); every window holds cmax channels: one 8-byte
 *            slot per (parity, source rank, channel), zero when empty - a rank stores the complement of its {min, max}
 *            pair into every window and polls its own (round 4: the pair is the signal; a small kernel behind the launch
 *            zeroes the launch's parity again).
 *   seq      1, 2, 3, ...: the same on every rank for the same launch; the ranks issue the same launches in the same
 *            order, each on ONE stream.  Not capturable into a graph (the number is a kernel argument; see _xrank_dev).
 *   status   device word: bit 2 is raised when a wait for a peer's record expired after timeout_ticks of the 100 MHz
 *            clock (the affected channels' outputs are NaN then) - check it at the next synchronisation point.
 *   ws / gws  as for cnnq_pc_minmax_qdq_auto (ws: cnnq_pc_minmax_qdq_workspace bytes; qp[CNNQ_NQP][C] and mm[2][C] - the
 *            GLOBAL extrema - are left at its start).
 * A rank whose shard has no single-launch kernel (shards may differ by a sample) speaks the same window protocol around
 * two passes over x, so every rank consumes the sequence number whatever its own plan. */
size_t cnnq_xrank_window_bytes(int world, int cmax);
int cnnq_xrank_alloc(int world, int cmax, void** window, unsigned char handle[64]);   /* allocates + zeroes the own window, exports its hipIpc handle; synchronises */
int cnnq_xrank_open(const unsigned char handle[64], void** window);                   /* maps a peer's window */
int cnnq_xrank_close(void* window);                                                   /* unmaps it */
int cnnq_xrank_free(void* window);                                                    /* releases the own window */
int cnnq_pc_minmax_qdq_xrank(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                             float* ws, void* gws, size_t gws_bytes, void* const* windows, int rank, int world, int cmax,
                             uint32_t seq, uint32_t* status, int64_t timeout_ticks, void* stream);
/* The same with DEVICE-side sequence numbers and the optional outputs of cnnq_pc_minmax_qdq_single (round 4):
 *   seq_dev   EIGHT device words (zero at start, one block per rank; round 6: words [4..7] hold the slots in use per parity,
 *             kept by the launches themselves): the launch's number is seq_dev[0] + 1 and the small kernel
 *             enqueued behind the launch advances the word - nothing about the call changes from launch to launch, so it
 *             can be captured into a HIP graph and replayed (every rank replays the same graph the same number of times).
 *   codes / hist_rep   as for cnnq_pc_minmax_qdq_single (num_bits <= 8): this rank's codes, and this rank's code counts in
 *             the replica tables - fold them with cnnq_hist_replicas_fold, sum the folded tables over the ranks, then
 *             cnnq_entropy gives the entropy of the global batch's codes (iq.py:586-587).
 * cnnq_hist_replicas_fold adds the replica tables to hist[256] and leaves them zero. */
int cnnq_pc_minmax_qdq_xrank_dev(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                                 float* ws, void* gws, size_t gws_bytes, void* const* windows, int rank, int world, int cmax,
                                 uint32_t* seq_dev, uint32_t* status, int64_t timeout_ticks, uint8_t* codes, uint64_t* hist_rep,
                                 void* stream);
/* ONE launch per tensor on the eager path (round 5): host numbering without the kernel behind the launch.
 *   seq       the host's launch number (1, 2, 3, ... the same on every rank, one per launch of the stream); 0: device numbering
 *             exactly as cnnq_pc_minmax_qdq_xrank_dev (seq_dev required), plus the clean-up of zero_c.
 *   zero_c    only without seq_dev: the channel count C of the launch TWO BACK on this stream (0 for the first two launches):
 *             workgroup 0 zeroes the slots that launch used - no reader of this rank is left and no peer can be pushing there yet
 *             (csrc/cnnq_xrank.hip.h).  With seq_dev (eight words) the launches keep these counts on the device and zero_c is
 *             ignored (round 6, ADVICE r5: captured, replayed and eager launches in any order leave the same trail).
 *   seq_dev   (may be NULL with seq != 0) device word that follows the host's count, so that a later captured launch with device
 *             numbering on the same word continues it.  A stream that switches to device numbering (its first captured launch)
 *             stays there, and passes the channel counts of its last two host-numbered launches as zero_c of its first two
 *             device-numbered ones.
 * The windows hold four parities (cnnq_xrank_window_bytes accounts for them). */
int cnnq_pc_minmax_qdq_xrank_seq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                                 float* ws, void* gws, size_t gws_bytes, void* const* windows, int rank, int world, int cmax,
                                 uint32_t seq, uint32_t* seq_dev, int zero_c, uint32_t* status, int64_t timeout_ticks, uint8_t* codes,
                                 uint64_t* hist_rep, void* stream);
int cnnq_hist_replicas_fold(uint64_t* hist_rep, uint64_t* hist, void* stream);

/* The dynamic ACIQ configurations (config 3: iq.py:327-352 + 409-451, statistics of this very tensor) behind
 * one call: pass A -> merge -> pass B when `b` is needed (laplace clipping, or bit allocation with the laplace
 * prior) -> merge -> cnnq_pc_params(cfg) -> fused Q/DQ.  `ws`: caller workspace of cnnq_pc_aciq_workspace(...)
 * bytes, 8-byte aligned (0 = invalid geometry); qp[CNNQ_NQP][C] and diag[CNNQ_NDIAG][C] (diag required with
 * bit allocation) are outputs.  Single process only: the cross-rank exchange sits between the launches. */
size_t cnnq_pc_aciq_workspace(int64_t N, int64_t C, int64_t HW, int aligned16);
int cnnq_pc_aciq_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws,
                     float* qp, float* diag, void* stream);

/* The same pipeline with pass B, the parameter derivation and the Q/DQ in ONE launch that reads x once (round 5; 12
 * instead of 16 bytes per element, four launches: pass A, merge, bit allocation, the single launch): the register-resident
 * tiles and the in-launch slot meeting of cnnq_pc_minmax_qdq_group, exchanging per-channel partial sums of |x - mean|
 * (added in member order by every member: the same b, bit for bit, in all of them and run after run).  For Laplace
 * clipping on the per-channel route (cfg->clip == 1, !direct_range), bit allocation on the 'gaus' prior only.  Returns
 * CNNQ_ENOTSUP - nothing enqueued - for other configurations, shapes without a single-launch plan, or a group workspace
 * `gws` (cnnq_group_ws_alloc, >= cnnq_pc_group_workspace bytes) that is NULL or too small.
 * ws: >= cnnq_pc_aciq_workspace bytes; stats[CNNQ_NSTAT][C] out (every row written; KURT / STD_POS zero); qp, diag as
 * cnnq_pc_params (diag required with bit allocation); codes / hist_rep (optional) as cnnq_pc_minmax_qdq_single;
 * flags: 0 (tests: 1 = skip the wait and recompute, 2 = IEEE divide for every channel). */
int cnnq_pc_aciq_qdq_single(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws,
                            void* gws, size_t gws_bytes, float* stats, float* qp, float* diag, uint8_t* codes,
                            uint64_t* hist_rep, unsigned flags, void* stream);
/* cnnq_pc_aciq_qdq_single when it applies, else cnnq_pc_aciq_qdq: one call, same ws. */
int cnnq_pc_aciq_qdq_auto(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws,
                          void* gws, size_t gws_bytes, float* qp, float* diag, void* stream);

/* Round 6 - configs 3 / 5 / 4 of a BATCH SHARD with the cross-rank exchange inside the launches (csrc/cnnq_xrank.hip.h,
 * csrc/cnnq_aciq.hip.h, csrc/cnnq_stats1.hip.h): the sharded forms of cnnq_pc_aciq_qdq_single, cnnq_pc_midtread_qdq_single and
 * cnnq_pc_stats_single - one host call per tensor, the launches of one GPU, NO collective.  The reference has no counterpart (its
 * DataParallel replicas quantize with their own sub-batch's statistics, inference_sim.py:196-200); these reproduce
 * int_quantizer.py:327-359 / 185-225 and statistic_manager_perchannel.py:45-79 on the GLOBAL batch from each rank's shard.
 *   cnnq_xrank_ctx   the exchange of config 2 (same windows, same launch numbering, same status word - the launches of all
 *                    four entry points share ONE sequence per stream):
 *       windows / rank / world / cmax   as cnnq_pc_minmax_qdq_xrank; these launches use eight slots per channel: 8 C <= cmax
 *       seq       the host's launch number (1, 2, 3, ...); 0: device numbering (captured launches), see _xrank_dev
 *       seq_dev   EIGHT device words, zeroed once (one block per rank and stream): [0] the launch number, [4..7] the slots in
 *                 use per parity - the launches record and clean them themselves (no host bookkeeping; also used by
 *                 cnnq_pc_minmax_qdq_xrank_seq / _dev since round 6, which therefore need the eight words too)
 *       status / timeout_ticks   as cnnq_pc_minmax_qdq_xrank: bit 2 when a wait for a peer expired (outputs NaN)
 * What travels (slot = word * C + channel): word 0 the {min, max} pair, 1 sum, 2 sum of squares, 3 / 4 the sums of relu(x), 5
 * the rank's element count, 6 sum |x - mean|, 7 sum ((x - mean) / std)^4 - sums as the complement of their fp64 bits, and every
 * reader adds the W ranks' words in RANK order: all ranks derive the same mean / std / b, hence the same parameters, bit for
 * bit.  A shard without a single-launch plan runs the chain's passes around the same slots, so the ranks need not agree on their
 * plans; every call consumes exactly one launch number. */
typedef struct cnnq_xrank_ctx {
    void* const* windows;
    int32_t rank, world, cmax;
    uint32_t seq;
    uint32_t* seq_dev;
    uint32_t* status;
    int64_t timeout_ticks;
} cnnq_xrank_ctx;
/* Config 3: k_moments -> k_xr_moments -> (k_bitalloc) -> k_fused_* <XR>: x read twice (12 bytes per element).  ws:
 * cnnq_pc_aciq_workspace bytes; stats [CNNQ_NSTAT][C], mom [CNNQ_NMOM][C] (fp64), qp, diag (as cnnq_pc_params): outputs, the
 * global batch's, the same on every rank.  Laplace clipping, bit allocation on the 'gaus' prior, num_bits <= 8 (else
 * CNNQ_ENOTSUP: nothing enqueued, no number consumed).  flags as cnnq_pc_aciq_qdq_single. */
int cnnq_pc_aciq_fused_xrank(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws, void* gws,
                             size_t gws_bytes, float* stats, double* mom, float* qp, float* diag, const cnnq_xrank_ctx* xc, unsigned flags,
                             void* stream);
/* Config 5 (mid-tread with clipping); mt [CNNQ_NMT][C] out; hist (optional, CNNQ_MT_HIST_WORDS(C), zeroed here) counts THIS
 * rank's codes: sum the ranks' tables, then cnnq_midtread_entropy_count with mom's COUNT row. */
int cnnq_pc_midtread_fused_xrank(const float* x, float* y, int64_t N, int64_t C, int64_t HW, double target, int sym, const double* tables,
                                 int ntab, void* ws, void* gws, size_t gws_bytes, float* stats, double* mom, float* mt, uint64_t* hist,
                                 const cnnq_xrank_ctx* xc, unsigned flags, void* stream);
/* Config 4: the seven statistics and the merged moment record of the GLOBAL batch on every rank, from ONE read of the shard
 * where it has a flat-tile plan (k_stats_flat<XR>), else k_moments -> k_xr_moments -> k_absdev -> k_xr_devsums.  ws:
 * cnnq_pc_stats_workspace bytes; flags as cnnq_pc_stats_single (bit 0: recompute path, bit 3: also channels of more than 128
 * tiles). */
int cnnq_pc_stats_xrank(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* ws, void* gws,
                        size_t gws_bytes, double* mom, float* stats, const cnnq_xrank_ctx* xc, unsigned flags, void* stream);

/* Weight bias / variance correction after quantization (iqm.py:374-391), in place on
 * wq[C][HW]: vcorr: wq = (wq - mean_q) * std_w/(std_q + 1e-8) + mean_q; bcorr: wq = wq - mean_q + mean_w
 * (mean_q is the pre-correction mean in both), with the reference's operation order.  stats_w /
 * stats_q are the stats tables (rows MEAN, STD) of the original and the quantized weights. */
int cnnq_pc_weight_correct(float* wq, int64_t C, int64_t HW, const float* stats_w, const float* stats_q, int vcorr,
                           int bcorr, void* stream);

/* Activation bias correction (iqm.py:180-196; -sm use with -bca):
 *   cnnq_pc_bcorr_sums   per channel sum(x'), sum(y), count(x' > 0), x' = relu(x) when relu_first,
 *                        -> part3[G][3][C] fp64 (G = cnnq_pc_groups);
 *   cnnq_pc_bcorr_bias   merges G records (or W ranks' sums) -> sums[3][C] (optional, for the
 *                        cross-rank exchange) and bias[c] = (sum x' - sum y)/(count + 1e-8);
 *   cnnq_pc_bcorr_apply  y += (y > 0) * bias[c], in place.
 * Fused form for per-channel quantizers with a parameter table qp (12 instead of 24 bytes per element,
 * the same floats): the quantized value is recomputed from x instead of stored and re-read -
 *   cnnq_pc_qdq_bcorr_sums  the sums of cnnq_pc_bcorr_sums with y = Q/DQ(x; qp) computed on the fly;
 *   cnnq_pc_qdq_bcorr       y = q + (q > 0) * bias[c], q = Q/DQ(x; qp) (iq.py:573-592 then iqm.py:196). */
int cnnq_pc_bcorr_sums(const float* x, const float* y, int64_t N, int64_t C, int64_t HW, int relu_first,
                       double* part3, void* stream);
int cnnq_pc_qdq_bcorr_sums(const float* x, int64_t N, int64_t C, int64_t HW, const float* qp, int relu_first,
                           double* part3, void* stream);
int cnnq_pc_qdq_bcorr(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* qp, const float* bias,
                      int reverse, void* stream);
int cnnq_pc_bcorr_bias(const double* part3, int G, int64_t C, double* sums, float* bias, void* stream);
int cnnq_pc_bcorr_apply(float* y, int64_t N, int64_t C, int64_t HW, const float* bias, void* stream);

/* Mid-tread quantization with per-channel BIN allocation (config 5, -mtq; iq.py:128-225):
 *   cnnq_pc_midtread_params  stats -> mt[CNNQ_NMT][C]: omega = round(C*2^target*std^(2/3)/sum) (eq. 10),
 *       clip != 0: alpha multiplier by linear interpolation in the 101-entry (omega, alpha) tables
 *       (`tables` = device fp64 [2][ntab], omega row then alpha row; iq.py:41-51,137-145),
 *       range = 2*alpha*b (sym) or max(mean,0)+alpha*b, Delta = range/omega, clamp bounds around the
 *       quantized mean; clip == 0 (weights): range = max-min (sym) or max.  One workgroup.
 *   cnnq_pc_midtread_qdq     y = clamp(round(x/Delta[c]), c_min[c], c_max[c]) * Delta[c] in one pass;
 *       `codes` (optional) fp32 codes; `hist` (optional, zeroed by the caller,
 *       CNNQ_MT_HIST_WORDS(C) uint64 words) the code histogram.
 *   cnnq_midtread_entropy    Shannon entropy of that histogram (equal clamp values merged). */
int cnnq_pc_midtread_params(const float* stats, int64_t C, double target, int clip, int sym, const double* tables,
                            int ntab, float* mt, void* stream);
int cnnq_pc_midtread_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* mt, int clip,
                         float* codes, uint64_t* hist, void* stream);
/* Config 5 (clip = 1) with pass B, the parameter derivation and the quantization in ONE launch that reads x once (round 5;
 * csrc/cnnq_aciq.hip.h MODE 1): pass A, merge, omega / clipping multiplier from the std alone, the single launch - 12 instead of
 * 16 bytes per element.  Channels too populous for the plain register tiles ([512,64,224,224]: 103 MB per channel) take eight
 * more tile rows in LDS and have the chip to themselves, one channel at a time.  mt: out, every row; row WSTART is the exact
 * first code of the histogram window for the non-negative range and an estimate (b = std / sqrt 2) for the symmetric one - the
 * histogram is correct either way (codes outside the window are counted in the global bins and the flag word says so).
 * stats [CNNQ_NSTAT][C] out; ws >= cnnq_pc_aciq_workspace bytes; gws / flags as cnnq_pc_aciq_qdq_single; hist: optional,
 * CNNQ_MT_HIST_WORDS(C) words, zeroed HERE (cnnq_midtread_entropy consumes it).  CNNQ_ENOTSUP: no single-launch plan. */
int cnnq_pc_midtread_qdq_single(const float* x, float* y, int64_t N, int64_t C, int64_t HW, double target, int sym,
                                const double* tables, int ntab, void* ws, void* gws, size_t gws_bytes, float* stats, float* mt,
                                uint64_t* hist, unsigned flags, void* stream);
int cnnq_midtread_entropy(const uint64_t* hist, const float* mt, int64_t C, int64_t total, float* out, void* stream);
/* ... with the element count in device memory: count[0] = elements per channel of the tensor the codes were counted over (row
 * CNNQ_MOM_COUNT of the merged moment record; total = count[0] * C).  For batch-sharded runs, whose shards may differ by a
 * sample: the global batch's size is known exactly on the device. */
int cnnq_midtread_entropy_count(const uint64_t* hist, const float* mt, int64_t C, const double* count, float* out, void* stream);
/* ... and of n <= 16 tensors in one launch (host arrays of n entries each: the histograms, their mt tables, channel counts and
 * element totals) -> out[n]. */
int cnnq_midtread_entropy_batch(int n, const uint64_t* const* hist, const float* const* mt, const int64_t* C, const int64_t* total,
                                float* out, void* stream);

/* Shannon entropy in bits, -sum p*log2(p) over the non-empty bins -> out[0] (utils/entropy.py:12-15). */
int cnnq_entropy(const uint64_t* hist, int nbins, float* out, void* stream);

/* Per-tensor path, replaces kernels/gemmlowp.cu:8-45 (`float2gemmlowp`).
 * cnnq_pt_setup derives {scale, shift, qmax, ...} on the device into ptp[8]:
 *   - from host scalars (range_offset_host != NULL: {range, offset}), or
 *   - from per-row statistics (rows MIN/MAX of a stats table with row stride `stats_stride`,
 *     `rows` entries each): rows_mode 0 = their mean over rows, the per-sample-then-batch-mean
 *     of iq.py:372,515-526; rows_mode 1 = min of mins / max of maxes, the whole-tensor
 *     min/max of iq.py:515-517; zero_min (iq.py:376-377) is applied afterwards;
 *   preserve_zero follows iq.py:613 when enforce_true_zero != 0.
 * cnnq_pt_qdq then streams x -> y with roundf / fminf / fmaxf semantics (gemmlowp.cu:11-24);
 * range <= 0 copies x to y (the reference returns its input there, gemmlowp.cu:31-32).
 * `noise` may be NULL (stochastic rounding is hard-wired off, iq.py:60). */
int cnnq_pt_setup(const float* range_offset_host, const float* stats, int64_t stats_stride, int rows,
                  int rows_mode, int zero_min, int num_bits, int int_exp, int enforce_true_zero, float* ptp,
                  void* stream);
int cnnq_pt_qdq(const float* x, float* y, int64_t n, const float* ptp, const float* noise, void* stream);
/* Config 1 in ONE launch (round 3): dynamic min / max of x viewed as [rows][n / rows] (rows_mode 0: batch mean of the
 * per-row extrema, iq.py:515-526; 1: the tensor's extrema) -> range / offset -> the GEMMLOWP kernel above, the same
 * bits as cnnq_pc_minmax + cnnq_pc_minmax_reduce + cnnq_pt_setup + cnnq_pt_qdq.  gws: an exchange workspace of
 * cnnq_group_ws_alloc of gws_bytes bytes.  ptp_out[8] (may be NULL) receives the parameters.  CNNQ_ENOTSUP: rows that
 * are not whole float4s, more than 1024 rows, unaligned pointers, a tensor of more 16 KB tiles than the workspace
 * holds 16-byte records for (use the four calls). */
int cnnq_pt_minmax_qdq_fused(const float* x, float* y, int64_t n, int rows, int rows_mode, int zero_min, int num_bits,
                             int int_exp, int enforce_true_zero, void* gws, size_t gws_bytes, float* ptp_out, void* stream);

/* KLD calibration (`-kld` with `-sm collect`): replaces the host loops of
 * inference/kld_threshold.py:6-84 (`get_kld_threshold_15bins`, one call per sample at
 * statistic_manager.py:80-82).  x is [rows][len] contiguous (rows = samples of the batch),
 * rowmm[2][rows] the per-row {min, max} (cnnq_pc_minmax + cnnq_pc_minmax_reduce with N = 1, C = rows).
 *   cnnq_kld_hist    hist[rows][CNNQ_KLD_BINS] (uint32; zeroed here): numpy.histogram(arr, 2001,
 *                    range = (-th, th)), th = max(|min|, |max|), float64 edges (kld_threshold.py:19-23);
 *                    NaN elements are not counted; len < 2^31.
 *   cnnq_kld_search  div[rows][CNNQ_KLD_NCAND] = KL(P || Q) for the 994 symmetric clipping candidates
 *                    (kld_threshold.py:31-77, 15 quantized bins), then per row
 *                    out[rows][3] = {optimal threshold (the upper histogram edge of the kept range),
 *                    its divergence, candidate index}: numpy.argmin semantics (kld_threshold.py:79-81).
 *                    The `kld_th` statistic is the maximum of out[:,0] over the batch. */
#define CNNQ_KLD_BINS 2001
#define CNNQ_KLD_QBINS 15
#define CNNQ_KLD_NCAND 994
int cnnq_kld_hist(const float* x, int64_t rows, int64_t len, const float* rowmm, uint32_t* hist, void* stream);
int cnnq_kld_search(const uint32_t* hist, int64_t rows, const float* rowmm, double* div, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNNQ_HIP_H */
