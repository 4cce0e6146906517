// cnnq_kernels.hip - gfx950 (MI355X / CDNA4) kernels behind include/cnnq_hip.h.
//
// Design: the whole path is HBM-bound elementwise + reduction work, so
// nothing here is shaped for MFMA.  All streaming kernels share ONE decomposition of the
// NCHW tensor x[N][C][HW]:
//
//   * the C*HW "plane" of one sample is cut into column blocks; a workgroup owns one column
//     block and walks it down the batch (n = n0 .. n1), so every lane keeps the SAME channel(s)
//     for its whole life: per-channel scale/zero-point are fetched once (staged through LDS)
//     and live in registers, the hot loop is load(16 B) -> ALU -> store(16 B), fully coalesced,
//     with no index division and no transposed copy;
//   * column blocks are aligned to channel boundaries: either a slice of ONE channel (mode 1,
//     large H*W) or k WHOLE channels (mode 2, small H*W), so reductions finish inside the
//     workgroup (wave64 shuffles + LDS) and each (group, channel) partial is written by
//     exactly one workgroup - no atomics, deterministic results;
//   * three load shapes: VEC4 (H*W % 4 == 0), VEC4-straddle (H*W % 4 != 0 but C*H*W % 4 == 0,
//     e.g. 7x7: a float4 may span two channels, per-element bookkeeping) and VEC1 (anything,
//     incl. unaligned base pointers);
//   * two launch geometries over that decomposition (make_geo): the REDUCTION passes (statistics)
//     use <= 64 batch splits, i.e. ~4096 long-lived workgroups that amortise their in-workgroup
//     reduction; the table-driven ELEMENTWISE passes (Q/DQ and friends) use ~14 KB of x per
//     workgroup, dispatched in address order - on MI355X read+write streaming reaches 6.1-6.7 TB/s
//     that way against 5.4 TB/s with long-lived workgroups; statistics passes over tensors too big
//     for the Infinity Cache use non-temporal loads (read-only 7.1 vs 6.3 TB/s).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (division must stay an IEEE
// divide followed by a separately rounded add: bit-exactness with the reference's aten ops).
//
// Layout: the kernels live in the cnnq_*.hip.h files next to this one (one file per stage of the path),
// all in one anonymous namespace of this single translation unit; below them is the C ABI.

#include <string.h>

#include <vector>
#include "cnnq_common.hip.h"
#include "cnnq_stats.hip.h"
#include "cnnq_params.hip.h"
#include "cnnq_qdq.hip.h"
#include "cnnq_pack4.hip.h"
#include "cnnq_midtread.hip.h"
#include "cnnq_corrections.hip.h"
#include "cnnq_pertensor.hip.h"
#include "cnnq_resident.hip.h"
#include "cnnq_group.hip.h"
#include "cnnq_plan.hip.h"
#include "cnnq_kld.hip.h"

extern "C" {

const char* cnnq_version(void) { return "cnnq-hip 0.4 gfx950"; }

int cnnq_pc_groups(int64_t N, int64_t C, int64_t HW, int aligned16) {
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, aligned16 != 0, 0, &v, &g);
    if (rc) return rc;
    return g.S * g.nb;
}

int cnnq_plan_describe(int64_t N, int64_t C, int64_t HW, int aligned16, int fine, int32_t out[12]) {
    if (!out) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, aligned16 != 0, 0, &v, &g, fine);
    if (rc) return rc;
    const int32_t vals[12] = {v.vec, v.A, v.J, g.mode, g.nb, g.w, g.k, g.ncb, g.S, TPB, g.S * g.nb, 0};
    for (int i = 0; i < 12; ++i) out[i] = vals[i];
    return 0;
}

int cnnq_pc_moments(const float* x, int64_t N, int64_t C, int64_t HW, int want_relu, double* part, void* stream) {
    if (!x || !part) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), 0, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_MOM(VEC, A, J)                                                                                        \
    do {                                                                                                             \
        if (want_relu && ntl) hipLaunchKernelGGL((k_moments<VEC, A, J, true, true>), grid, block, 0, st, x, g, part);   \
        else if (want_relu) hipLaunchKernelGGL((k_moments<VEC, A, J, true, false>), grid, block, 0, st, x, g, part);    \
        else if (ntl) hipLaunchKernelGGL((k_moments<VEC, A, J, false, true>), grid, block, 0, st, x, g, part);          \
        else hipLaunchKernelGGL((k_moments<VEC, A, J, false, false>), grid, block, 0, st, x, g, part);                  \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_MOM);
#undef LAUNCH_MOM
    return launch_status();
}

int cnnq_pc_combine(const double* part, int G, int64_t C, int has_relu, double* mom, float* stats, void* stream) {
    if (!part || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || (!mom && !stats)) return CNNQ_EINVAL;
    const dim3 grid((unsigned)((C + merge_cpw(G, (int)C) - 1) / merge_cpw(G, (int)C))), block(TPB);
    hipLaunchKernelGGL(k_combine, grid, block, 0, (hipStream_t)stream, part, G, (int)C, has_relu, mom, stats);
    return launch_status();
}

int cnnq_pc_absdev(const float* x, int64_t N, int64_t C, int64_t HW, const float* stats, int want_kurt,
                   double* part2, void* stream) {
    if (!x || !stats || !part2) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), /*rev=*/1, &v, &g);   // descending: follows the ascending pass A
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_DEV(VEC, A, J)                                                                                             \
    do {                                                                                                                  \
        if (want_kurt && ntl) hipLaunchKernelGGL((k_absdev<VEC, A, J, true, true>), grid, block, 0, st, x, g, stats, part2);  \
        else if (want_kurt) hipLaunchKernelGGL((k_absdev<VEC, A, J, true, false>), grid, block, 0, st, x, g, stats, part2);   \
        else if (ntl) hipLaunchKernelGGL((k_absdev<VEC, A, J, false, true>), grid, block, 0, st, x, g, stats, part2);         \
        else hipLaunchKernelGGL((k_absdev<VEC, A, J, false, false>), grid, block, 0, st, x, g, stats, part2);                 \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_DEV);
#undef LAUNCH_DEV
    return launch_status();
}

// pass B straight from the UNMERGED pass-A records (each workgroup merges its own channels' records in its prologue)
static int absdev_raw(const float* x, int64_t N, int64_t C, int64_t HW, const double* part, int want_kurt, double* part2,
                      void* stream) {
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), /*rev=*/1, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
    const int G = g.S * g.nb;
    const float* nostats = nullptr;
#define LAUNCH_DEVR(VEC, A, J)                                                                                              \
    do {                                                                                                                    \
        if (want_kurt && ntl) hipLaunchKernelGGL((k_absdev<VEC, A, J, true, true, true>), grid, block, 0, st, x, g, nostats, part2, part, G);   \
        else if (want_kurt) hipLaunchKernelGGL((k_absdev<VEC, A, J, true, false, true>), grid, block, 0, st, x, g, nostats, part2, part, G);    \
        else if (ntl) hipLaunchKernelGGL((k_absdev<VEC, A, J, false, true, true>), grid, block, 0, st, x, g, nostats, part2, part, G);          \
        else hipLaunchKernelGGL((k_absdev<VEC, A, J, false, false, true>), grid, block, 0, st, x, g, nostats, part2, part, G);                  \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_DEVR);
#undef LAUNCH_DEVR
    return launch_status();
}

// All per-channel statistics of one tensor behind ONE call and one caller workspace: pass A -> (pass B with the
// pass-A merge fused into its prologue -> one final merge of both passes) - three launches for the full set of
// smpc.py:45-79 instead of four, two for {min, max, mean, std}.  ws: doubles part[G][NMOM][C], part2[G][NDEV][C].
size_t cnnq_pc_stats_workspace(int64_t N, int64_t C, int64_t HW, int aligned16) {
    const int G = cnnq_pc_groups(N, C, HW, aligned16);
    if (G <= 0) return 0;
    return ((size_t)G * (CNNQ_NMOM + CNNQ_NDEV)) * (size_t)C * sizeof(double);
}

int cnnq_pc_stats(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* ws,
                  double* mom, float* stats, void* stream) {
    if (!x || !ws || !stats || ((uintptr_t)ws & 7)) return CNNQ_EINVAL;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    double* part2 = part + (size_t)G * CNNQ_NMOM * C;
    hipStream_t st = (hipStream_t)stream;
    // the merge kernels write every row of the table (zero where nothing was requested): no memset
    int rc = cnnq_pc_moments(x, N, C, HW, need_relu, part, stream);
    if (rc) return rc;
    if (!(need_b || need_kurt)) return cnnq_pc_combine(part, G, C, need_relu, mom, stats, stream);
    rc = absdev_raw(x, N, C, HW, part, need_kurt, part2, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_combine_all, dim3((unsigned)((C + merge_cpw(G, (int)C) - 1) / merge_cpw(G, (int)C))), dim3(TPB), 0, st, part, part2, G,
                       (int)C, need_relu, need_kurt, mom, stats);
    return launch_status();
}

// The same table from ONE launch that reads x once (cnnq_stats1.hip.h: the tile stays in registers across pass A and pass B,
// the partial sums meet through the slot region of the group workspace): 4 instead of 8 bytes per element, one launch instead
// of three.  Flat-tile plans only; CNNQ_ENOTSUP - nothing enqueued - otherwise (and for a gws that is NULL or too small): the
// caller takes cnnq_pc_stats.  flags: bit 0 - skip the waits and recompute (tests); bit 3 - also channels of more than 256 tiles and
// the row-piece routing that the default (0: what cnnq_pc_stats_auto passes) leaves to the chain because it loses there.
static int stats_single_impl(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* gws,
                             size_t gws_bytes, double* mom, float* stats, unsigned flags, void* stream, bool aligned, bool dry) {
    GPlan gp;
    if (plan_sums(N, C, HW, aligned, &gp, 0) != 0 || gp.ws_bytes > gws_bytes) return CNNQ_ENOTSUP;
    // two meetings with nothing to write behind them: with many members per channel they cost more than the second read.  Measured
    // up to 196 members ([512,64,112,112]: 431 us against the chain's 495-560 since the arithmetic of the tile went two elements
    // per instruction, round 6; 529 before, when the rule was 128); nothing measured beyond 256 - tests force those with flag 8
    if (gp.Gs > ST_MAX_MEMBERS && !(flags & 8u)) return CNNQ_ENOTSUP;
    St1Args sa;
    sa.stats = stats;
    sa.mom = mom;
    sa.count = (double)N * (double)HW;
    sa.need_relu = need_relu ? 1 : 0;
    sa.need_dev = (need_b || need_kurt) ? 1 : 0;
    sa.need_kurt = need_kurt ? 1 : 0;
    if (!gp.flat) {
        // short rows: row-piece tiles (k_stats_group), where they beat the chain (stats_group_pays: every one-channel-per-lane
        // shape, straddling rows only while the tensor is small)
        if (!(flags & 8u) && !stats_group_pays(gp, N, C, HW)) return CNNQ_ENOTSUP;
        const int rc = launch_stats_group(x, gp, sa, gws, gws_bytes, flags & 1u, (hipStream_t)stream, nullptr, dry);
        return (dry && rc == 0) ? 2 : rc;
    }
    const int rc = launch_stats_flat(x, gp, sa, gws, flags & 1u, N * C * HW * 4 > NT_BYTES, (hipStream_t)stream, nullptr, dry);
    return (dry && rc == 0) ? 1 : rc;
}

int cnnq_pc_stats_single(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* gws,
                         size_t gws_bytes, double* mom, float* stats, unsigned flags, void* stream) {
    if (!x || !stats || (gws && ((uintptr_t)gws & 127)) || ((uintptr_t)mom & 7)) return CNNQ_EINVAL;
    if (!gws) return CNNQ_ENOTSUP;
    return stats_single_impl(x, N, C, HW, need_b, need_kurt, need_relu, gws, gws_bytes, mom, stats, flags, stream, al16(x), false);
}

// Which route cnnq_pc_stats_single / _auto take for this geometry (nothing is launched): 1 - the flat-tile single launch, 2 - the
// row-piece single launch, 0 - CNNQ_ENOTSUP there, i.e. the three-launch chain.  For accounting (bench.py prices config 4 at
// the bytes its launches move) and for callers that want to size their expectations; negative: an error of the plan.
int cnnq_pc_stats_route(int64_t N, int64_t C, int64_t HW, int aligned16, size_t gws_bytes, unsigned flags) {
    const int rc = stats_single_impl(nullptr, N, C, HW, 1, 1, 1, nullptr, gws_bytes, nullptr, nullptr, flags, nullptr, aligned16 != 0, true);
    return rc == CNNQ_ENOTSUP ? 0 : rc;
}

// cnnq_pc_stats_single when it applies, else cnnq_pc_stats: one call, the same ws
int cnnq_pc_stats_auto(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* ws, void* gws,
                       size_t gws_bytes, double* mom, float* stats, void* stream) {
    const int rc = cnnq_pc_stats_single(x, N, C, HW, need_b, need_kurt, need_relu, gws, gws_bytes, mom, stats, 0u, stream);
    if (rc != CNNQ_ENOTSUP) return rc;
    return cnnq_pc_stats(x, N, C, HW, need_b, need_kurt, need_relu, ws, mom, stats, stream);
}

int cnnq_pc_combine_dev(const double* part2, int G, int64_t C, const double* mom, int want_kurt, double* dev_out,
                        float* stats, void* stream) {
    if (!part2 || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || (!dev_out && !stats) || (stats && !mom))
        return CNNQ_EINVAL;
    const dim3 grid((unsigned)((C + merge_cpw(G, (int)C) - 1) / merge_cpw(G, (int)C))), block(TPB);
    hipLaunchKernelGGL(k_combine_dev, grid, block, 0, (hipStream_t)stream, part2, G, (int)C, mom, want_kurt, dev_out,
                       stats);
    return launch_status();
}

int cnnq_pc_params(const float* stats, int64_t C, const cnnq_params_cfg* cfg, float* qp, float* diag,
                   void* stream) {
    if (!stats || !cfg || !qp || C <= 0 || C >= ((int64_t)1 << 31)) return CNNQ_EINVAL;
    if (cfg->num_bits < 1 || cfg->num_bits > 32 || cfg->clip < 0 || cfg->clip > 3) return CNNQ_EINVAL;
    // the ACIQ factor tables have entries for 0..8 bits only (iq.py:14-41: the reference's alpha_laplace / alpha_gaus
    // dictionaries raise KeyError beyond 8); wider codes are accepted for min/max and the '<p>std' clip alone
    if ((cfg->clip == 1 || cfg->clip == 2) && cfg->num_bits > 8) return CNNQ_EINVAL;
    if (cfg->bit_alloc && cfg->num_bits <= 4 && !diag) return CNNQ_EINVAL;  // bit table lives in diag
    float* bits_ws = diag ? diag + (size_t)CNNQ_DIAG_BITS * C : nullptr;
    const int threads = (int)(C >= PTPB ? PTPB : ((C + 63) / 64) * 64);
    hipLaunchKernelGGL(k_params, dim3(1), dim3(threads), 0, (hipStream_t)stream, stats, (int)C, *cfg, qp, diag,
                       bits_ws);
    return launch_status();
}

// channel-slice views: `sample_stride` floats between consecutive samples (>= C*HW; 0 = contiguous).  The
// kernels only ever use the plane size as that stride, so a slice x[:, c0:c1] of a wider NCHW tensor is just
// (x + c0*HW, C = c1 - c0, sample_stride = C_total*HW).
static int strided_ok(int64_t C, int64_t HW, int64_t sample_stride) {
    return sample_stride == 0 || (sample_stride >= C * HW && sample_stride < ((int64_t)1 << 31));
}

int cnnq_pc_qdq_strided(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int64_t sample_stride,
                        const float* qp, uint8_t* codes, uint64_t* hist, int reverse, void* stream) {
    if (!x || !y || !qp || !strided_ok(C, HW, sample_stride)) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const bool al = al16(x) && al16(y) && (!codes || ((uintptr_t)codes & 3) == 0) && sample_stride % 4 == 0;
    // the histogram variant zeroes and flushes an LDS table per workgroup: keep its workgroups long
    const int rc = plan(N, C, HW, al, reverse ? 1 : 0, &v, &g, /*fine=*/hist ? 0 : 1);
    if (rc) return rc;
    if (sample_stride) g.P = (int)sample_stride;
    return launch_qdq(x, y, g, v, qp, codes, reinterpret_cast<unsigned long long*>(hist), (hipStream_t)stream);
}

int cnnq_pc_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* qp, uint8_t* codes,
                uint64_t* hist, int reverse, void* stream) {
    return cnnq_pc_qdq_strided(x, y, N, C, HW, 0, qp, codes, hist, reverse, stream);
}

// stored-format codes: bits = 4 (two per byte) or 8 (one per byte)
static int quantize_codes(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp, int bits,
                          void* stream) {
    if (!x || !packed || !qp) return CNNQ_EINVAL;
    if (HW % 4 != 0 || !al16(x) || ((uintptr_t)packed & (bits == 4 ? 1 : 3))) return CNNQ_EINVAL;   // whole float4s
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, true, 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_QP(J)                                                                              \
    do {                                                                                          \
        if (bits == 4) hipLaunchKernelGGL((k_q_pack4<J, 4>), grid, block, 0, st, x, packed, g, qp); \
        else hipLaunchKernelGGL((k_q_pack4<J, 8>), grid, block, 0, st, x, packed, g, qp);           \
    } while (0)
    if (v.J == 4) LAUNCH_QP(4);
    else if (v.J == 2) LAUNCH_QP(2);
    else LAUNCH_QP(1);
#undef LAUNCH_QP
    return launch_status();
}

static int dequantize_codes(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp, int bits,
                            void* stream) {
    if (!packed || !y || !qp) return CNNQ_EINVAL;
    if (HW % 4 != 0 || !al16(y) || ((uintptr_t)packed & (bits == 4 ? 1 : 3))) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, true, 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_UP(J)                                                                                 \
    do {                                                                                             \
        if (bits == 4) hipLaunchKernelGGL((k_unpack4_dq<J, 4>), grid, block, 0, st, packed, y, g, qp); \
        else hipLaunchKernelGGL((k_unpack4_dq<J, 8>), grid, block, 0, st, packed, y, g, qp);           \
    } while (0)
    if (v.J == 4) LAUNCH_UP(4);
    else if (v.J == 2) LAUNCH_UP(2);
    else LAUNCH_UP(1);
#undef LAUNCH_UP
    return launch_status();
}

int cnnq_pc_quantize_pack4(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                           void* stream) {
    return quantize_codes(x, packed, N, C, HW, qp, 4, stream);
}
int cnnq_pc_dequantize_pack4(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                             void* stream) {
    return dequantize_codes(packed, y, N, C, HW, qp, 4, stream);
}
int cnnq_pc_quantize_u8(const float* x, uint8_t* codes, int64_t N, int64_t C, int64_t HW, const float* qp,
                        void* stream) {
    return quantize_codes(x, codes, N, C, HW, qp, 8, stream);
}
int cnnq_pc_dequantize_u8(const uint8_t* codes, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                          void* stream) {
    return dequantize_codes(codes, y, N, C, HW, qp, 8, stream);
}

// variable-width packed codes (bit allocation as the stored format)
constexpr bool PACK_FLAT_DEFAULT = false;      // form 0 of the store direction stays k_pack_lean: k_pack_flat measured equal (cnnq_pack4.hip.h)
static int packed_launch(bool quant, const float* x, float* y, uint8_t* packed, int64_t N, int64_t C, int64_t HW,
                         const float* qp, const float* bits, const uint32_t* rowoff, void* stream, int form = 0) {
    if (!packed || !qp || !bits || !rowoff || N <= 0 || C <= 0 || HW <= 0) return CNNQ_EINVAL;
    if (C * HW >= (int64_t)1 << 31 || N >= (int64_t)1 << 31 || C > 65535) return CNNQ_ERANGE;
    // k adjacent channels per workgroup: >= 512 eight-element groups per sample (16 KB contiguous) when the layer has
    // the channels for it
    const int64_t ngroups = (HW + 7) / 8;
    int64_t k = (512 + ngroups - 1) / ngroups;
    if (k > MAXCH) k = MAXCH;
    if (k > C) k = C;
    const int64_t ncb = (C + k - 1) / k;
    // short workgroups in address order, like the other elementwise passes: ~14-28 KB of x each
    int64_t rows = (14336 + k * HW * 2) / (k * HW * 4);
    if (rows < 1) rows = 1;
    int64_t S = (N + rows - 1) / rows;
    if (S * ncb >= (int64_t)1 << 31) return CNNQ_ERANGE;
    if (k * ngroups >= (int64_t)1 << 24) return CNNQ_ERANGE;   // the in-loop index arithmetic is exact in fp32 below that
    // round 3: the lean form (one channel per wave, scalar parameters) for whole-float4 rows; form 1 forces the general
    // kernel, 2 the lean one (CNNQ_ENOTSUP when the geometry does not allow it)
    if (quant && (form == 0 || form == 3)) {
        // round 5: one-shot workgroups in address order of x (k_pack_flat); form 3 forces it (CNNQ_ENOTSUP for rows that are
        // not whole float4s, an unaligned x or stream, or 2^32 groups of 32 codes and more)
        const int64_t fgroups = (HW / 4 + 7) / 8, ftotal = N * C * fgroups;
        constexpr int FU = 4;
        const bool flat_ok = HW % 4 == 0 && al16(x) && ((uintptr_t)packed & 3) == 0 && ftotal < ((int64_t)1 << 32) - 32 * FU &&
                             fgroups < ((int64_t)1 << 24) - 32 * FU && N * C < ((int64_t)1 << 31);
        if (flat_ok && (PACK_FLAT_DEFAULT || form == 3)) {
            hipLaunchKernelGGL((k_pack_flat<FU>), dim3((unsigned)((ftotal + 32 * FU - 1) / (32 * FU))), dim3(TPB), 0, (hipStream_t)stream, x, packed,
                               (unsigned)(N * C), (int)C, (int)HW, (unsigned)fgroups, (unsigned)ftotal, qp, bits, rowoff);
            return launch_status();
        }
        if (form == 3) return CNNQ_ENOTSUP;
    }
    if (quant && form != 1) {
        const int64_t nsl = 2 * ngroups;
        const int64_t rpc = nsl <= 128 ? 128 / nsl : 1;
        // whole-float4 rows of an aligned x, or (RAG) any row of at least 8 elements: its slots are 4-byte aligned
        const bool rag = HW % 4 != 0;
        const bool lean_ok = (rag ? HW >= 8 && ((uintptr_t)x & 3) == 0 : al16(x)) && rpc * C * HW * 4 < ((int64_t)1 << 32);
        if (lean_ok) {
            // ~8 KB of x per wave (32 KB per workgroup of four adjacent channels), whole chunks of rows
            static const int64_t wave_bytes = env_int("CNNQ_PACK_WAVE_BYTES", 8192);   // development knob
            int64_t rpw = (wave_bytes + HW * 2) / (HW * 4);
            if (rpw < 1) rpw = 1;
            rpw = ((rpw + rpc - 1) / rpc) * rpc;
            const int64_t Sl = (N + rpw - 1) / rpw, ncb4 = (C + 3) / 4;
            if (Sl * ncb4 >= (int64_t)1 << 31) return CNNQ_ERANGE;
            const dim3 lgrid((unsigned)(Sl * ncb4)), lblock(TPB);
            hipStream_t lst = (hipStream_t)stream;
#define LAUNCH_LEAN(S, R) hipLaunchKernelGGL((k_pack_lean<S, R>), lgrid, lblock, 0, lst, x, packed, (int)N, (int)C, (int)HW, (int)rpw, qp, bits, rowoff)
            if (nsl <= 128) { if (rag) LAUNCH_LEAN(true, true); else LAUNCH_LEAN(true, false); }
            else { if (rag) LAUNCH_LEAN(false, true); else LAUNCH_LEAN(false, false); }
#undef LAUNCH_LEAN
            return launch_status();
        }
        if (form == 2) return CNNQ_ENOTSUP;
    }
    if (!quant && (form == 0 || form == 3)) {
        // round 4: one-shot workgroups in address order of y (k_unpack_flat); form 3 forces it (CNNQ_ENOTSUP when y is not
        // 16-byte aligned, the stream not 4-byte aligned, or the tensor has 2^32 elements or more)
        static const int allow_flat = env_int("CNNQ_UNPACK_FLAT", 1);               // development knob
        const int64_t total = N * C * HW;
        const bool flat_ok = al16(y) && ((uintptr_t)packed & 3) == 0 && total < ((int64_t)1 << 32) - 1024 && N * C < ((int64_t)1 << 31);
        if (flat_ok && (allow_flat || form == 3)) {
            const unsigned total4 = (unsigned)((total + 3) / 4), tail = (unsigned)(total - (int64_t)(total4 - 1) * 4);
            hipStream_t fst = (hipStream_t)stream;
            static const int U = env_int("CNNQ_UNPACK_U", 4);                     // development knob: float4 per lane
            const dim3 fblock(TPB);
#define LAUNCH_UF(R, UU) hipLaunchKernelGGL((k_unpack_flat<R, UU>), dim3((total4 + TPB * UU - 1) / (TPB * UU)), fblock, 0, fst, packed, y, (int)N, (int)C, (int)HW, qp, bits, rowoff, total4, tail)
            if (HW % 4 == 0) { if (U == 1) LAUNCH_UF(1, 1); else if (U == 2) LAUNCH_UF(1, 2); else if (U == 8) LAUNCH_UF(1, 8); else LAUNCH_UF(1, 4); }
            else if (HW >= 4) { if (U == 1) LAUNCH_UF(2, 1); else if (U == 2) LAUNCH_UF(2, 2); else LAUNCH_UF(2, 4); }
            else LAUNCH_UF(4, 1);
#undef LAUNCH_UF
            return launch_status();
        }
        if (form == 3) return CNNQ_ENOTSUP;
    }
    if (!quant && form != 1 && form != 3) {
        // the load direction's lean form (k_unpack_lean): the same geometry; the packed stream is read as dwords
        const int64_t nsl = 2 * ngroups;
        const int64_t rpc = nsl <= 128 ? 128 / nsl : 1;
        const bool rag = HW % 4 != 0;
        const bool lean_ok = (rag ? HW >= 8 && ((uintptr_t)y & 3) == 0 : al16(y)) && ((uintptr_t)packed & 3) == 0 &&
                             rpc * C * HW * 4 < ((int64_t)1 << 32);
        if (lean_ok) {
            static const int64_t wave_bytes = env_int("CNNQ_UNPACK_WAVE_BYTES", 8192);   // development knob
            int64_t rpw = (wave_bytes + HW * 2) / (HW * 4);
            if (rpw < 1) rpw = 1;
            rpw = ((rpw + rpc - 1) / rpc) * rpc;
            const int64_t Sl = (N + rpw - 1) / rpw, ncb4 = (C + 3) / 4;
            if (Sl * ncb4 >= (int64_t)1 << 31) return CNNQ_ERANGE;
            const dim3 lgrid((unsigned)(Sl * ncb4)), lblock(TPB);
            hipStream_t lst = (hipStream_t)stream;
#define LAUNCH_ULEAN(S, R) hipLaunchKernelGGL((k_unpack_lean<S, R>), lgrid, lblock, 0, lst, packed, y, (int)N, (int)C, (int)HW, (int)rpw, qp, bits, rowoff)
            if (nsl <= 128) { if (rag) LAUNCH_ULEAN(true, true); else LAUNCH_ULEAN(true, false); }
            else { if (rag) LAUNCH_ULEAN(false, true); else LAUNCH_ULEAN(false, false); }
#undef LAUNCH_ULEAN
            return launch_status();
        }
        if (form == 2) return CNNQ_ENOTSUP;
    }
    const dim3 grid((unsigned)(ncb * S)), block(TPB);
    static const int64_t rows_min = env_int("CNNQ_PACK_ROWS_MIN", 256);   // development knob (slots per row)
    const bool rows_form = 2 * ngroups >= rows_min;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_PK(Q, R) \
    hipLaunchKernelGGL((k_packed<Q, R>), grid, block, 0, st, x, y, packed, (int)N, (int)C, (int)HW, (int)S, (int)k, qp, bits, rowoff)
    if (quant) { if (rows_form) LAUNCH_PK(true, true); else LAUNCH_PK(true, false); }
    else { if (rows_form) LAUNCH_PK(false, true); else LAUNCH_PK(false, false); }
#undef LAUNCH_PK
    return launch_status();
}

int cnnq_pc_packed_layout(const float* bits, int64_t C, int64_t HW, uint32_t* rowoff, void* stream) {
    if (!bits || !rowoff || C <= 0 || HW <= 0 || C * HW >= (int64_t)1 << 31) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_packed_layout, dim3(1), dim3(PTPB), 0, (hipStream_t)stream, bits, (int)C, (int)HW, rowoff);
    return launch_status();
}

int cnnq_pc_quantize_packed(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                            const float* bits, const uint32_t* rowoff, void* stream) {
    if (!x) return CNNQ_EINVAL;
    return packed_launch(true, x, nullptr, packed, N, C, HW, qp, bits, rowoff, stream);
}

// the same with the kernel form spelled out: 0 = the library's choice, 1 = the general kernel (any geometry), 2 = the
// lean kernel (one channel per wave; CNNQ_ENOTSUP for rows of fewer than 8 elements, or whole-float4 rows of an x that is not 16-byte aligned).  Same bytes.
int cnnq_pc_quantize_packed_form(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                                 const float* bits, const uint32_t* rowoff, int form, void* stream) {
    if (!x || form < 0 || form > 3) return CNNQ_EINVAL;
    return packed_launch(true, x, nullptr, packed, N, C, HW, qp, bits, rowoff, stream, form);
}

int cnnq_pc_dequantize_packed(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                              const float* bits, const uint32_t* rowoff, void* stream) {
    if (!y) return CNNQ_EINVAL;
    return packed_launch(false, nullptr, y, const_cast<uint8_t*>(packed), N, C, HW, qp, bits, rowoff, stream);
}

// the same with the kernel form spelled out (0 / 1 / 2 as in cnnq_pc_quantize_packed_form; the lean form needs a 4-byte
// aligned packed buffer on top of the conditions on y).  Same floats.
int cnnq_pc_dequantize_packed_form(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                                   const float* bits, const uint32_t* rowoff, int form, void* stream) {
    if (!y || form < 0 || form > 3) return CNNQ_EINVAL;
    return packed_launch(false, nullptr, y, const_cast<uint8_t*>(packed), N, C, HW, qp, bits, rowoff, stream, form);
}

int cnnq_pc_minmax_strided(const float* x, int64_t N, int64_t C, int64_t HW, int64_t sample_stride, float* pmm,
                           void* stream) {
    if (!x || !pmm || !strided_ok(C, HW, sample_stride)) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x) && sample_stride % 4 == 0, 0, &v, &g);
    if (rc) return rc;
    if (sample_stride) g.P = (int)sample_stride;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_MM(VEC, A, J)                                                                         \
    do {                                                                                             \
        if (ntl) hipLaunchKernelGGL((k_minmax<VEC, A, J, true>), grid, block, 0, st, x, g, pmm);       \
        else hipLaunchKernelGGL((k_minmax<VEC, A, J, false>), grid, block, 0, st, x, g, pmm);          \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_MM);
#undef LAUNCH_MM
    return launch_status();
}

int cnnq_pc_minmax(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, void* stream) {
    return cnnq_pc_minmax_strided(x, N, C, HW, 0, pmm, stream);
}

int cnnq_pc_minmax_reduce(const float* pmm, int G, int64_t C, float* out, void* stream) {
    if (!pmm || !out || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31)) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_minmax_reduce, dim3((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0,
                       (hipStream_t)stream, pmm, G, (int)C, out);
    return launch_status();
}

int cnnq_pc_minmax_params(const float* pmm, int G, int64_t C, int num_bits, int positive, float* qp, void* stream) {
    if (!pmm || !qp || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || num_bits < 1 || num_bits > 32)
        return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_minmax_params, dim3((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0,
                       (hipStream_t)stream, pmm, G, (int)C, num_bits, positive ? 1 : 0, qp);
    return launch_status();
}

int cnnq_pc_minmax_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                       float* pmm, float* qp, uint8_t* codes, uint64_t* hist, void* stream) {
    if (!x || !y || !pmm || !qp || num_bits < 1 || num_bits > 32) return CNNQ_EINVAL;
    if ((codes || hist) && num_bits > 8) return CNNQ_EINVAL;   // one byte per code, 256 histogram bins
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    int rc = cnnq_pc_minmax(x, N, C, HW, pmm, stream);
    if (rc) return rc;
    rc = cnnq_pc_minmax_params(pmm, G, C, num_bits, positive, qp, stream);
    if (rc) return rc;
    // descending address order: what the statistics pass read last is re-read first
    return cnnq_pc_qdq(x, y, N, C, HW, qp, codes, hist, /*reverse=*/1, stream);
}

// Config 2 in one launch and one read of x (cnnq_resident.hip.h)
int cnnq_pc_resident_describe(int64_t N, int64_t C, int64_t HW, int32_t out[8]) {
    if (!out) return CNNQ_EINVAL;
    WPlan p;
    const int rc = plan_whole(N, C, HW, true, &p);
    if (rc) return rc;
    const int32_t vals[8] = {p.A, p.T, p.K, p.g.k, p.g.CL, p.g.RL, p.wgs, 0};
    for (int i = 0; i < 8; ++i) out[i] = vals[i];
    return 0;
}

int cnnq_pc_minmax_qdq_resident(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                                float* qp, float* mm, void* stream) {
    if (!x || !y || !qp || num_bits < 1 || num_bits > 32) return CNNQ_EINVAL;
    WPlan p;
    const int rc = plan_whole(N, C, HW, al16(x) && al16(y), &p);
    if (rc) return rc;
    return launch_whole(x, y, p, num_bits, positive ? 1 : 0, qp, mm, (hipStream_t)stream);
}

// Config 2 in one launch and one read of x for tensors whose channels span several workgroups (cnnq_group.hip.h)
size_t cnnq_pc_group_workspace(int64_t N, int64_t C, int64_t HW) {
    // the larger of the two tilings a caller may end up with (flat tiles for the single launch, row pieces for
    // k_minmax_group's half of the multi-GPU path)
    GPlan p, q, r;
    const size_t a = plan_group(N, C, HW, true, &p, true, 1) ? 0 : p.ws_bytes;
    const size_t b = plan_group(N, C, HW, true, &q, /*allow_flat=*/false) ? 0 : q.ws_bytes;
    const size_t c = plan_sums(N, C, HW, true, &r, 1) ? 0 : r.ws_bytes;      // (straddling rows: shorter tiles, more members)
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

// The exchange workspace lives in fine-grained (uncached) device memory: the pairs one XCD writes must be what
// another XCD reads in the same launch AND in the next one, whatever a per-XCD L2 (not coherent with the others;
// agent-scope acquires drop L1, not L2) may still hold of the same slots.  These are the only entry points of the
// path that allocate; they synchronise the device.
int cnnq_group_ws_alloc(size_t bytes, void** ws) {
    if (!ws || bytes < GRP_WS_PAIRS) return CNNQ_EINVAL;
    hipError_t e = hipExtMallocWithFlags(ws, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*ws, 0, bytes);
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceSynchronize();
}

int cnnq_group_ws_free(void* ws) { return ws ? (int)hipFree(ws) : CNNQ_EINVAL; }

int cnnq_group_ws_status(const void* ws, uint32_t* status_host) {
    if (!ws || !status_host) return CNNQ_EINVAL;
    return (int)hipMemcpy(status_host, ws, sizeof(uint32_t), hipMemcpyDeviceToHost);
}

int cnnq_group_ws_status_clear(void* ws) {
    if (!ws) return CNNQ_EINVAL;
    const uint32_t zero = 0;
    return (int)hipMemcpy(ws, &zero, sizeof(uint32_t), hipMemcpyHostToDevice);   // synchronises, like the read
}

// tests: the regions every launch must leave zero - the header after the status word, the counter lines, the slots -
// copied to the host (synchronising) and counted: *nonzero_words_host = the 32-bit words that are not zero
int cnnq_group_ws_at_rest(const void* ws, uint64_t* nonzero_words_host) {
    if (!ws || !nonzero_words_host) return CNNQ_EINVAL;
    std::vector<uint32_t> h(GRP_WS_PAIRS / 4);
    const hipError_t e = hipMemcpy(h.data(), ws, GRP_WS_PAIRS, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    // the fused per-tensor kernel's corner of the header works by epoch (k_pt_fused: the epoch word and the row records
    // are written before they are read in every launch, nothing there needs to be zero): not part of the invariant
    constexpr size_t ptf0 = PTF_OFF / 4, ptf1 = (PTF_OFF + 256 + (size_t)PTF_MAX_ROWS * 16) / 4;
    uint64_t n = 0;
    for (size_t i = 1; i < h.size(); ++i) n += (i < ptf0 || i >= ptf1) && h[i] != 0u;
    *nonzero_words_host = n;
    return 0;
}

#ifdef GRP_TRACE
// development build only (tools/trace_group.py; not declared in include/cnnq_hip.h, absent from the product library)
int cnnq_debug_group_trace(void* buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_grp_trace), &p, sizeof(p));
}
#endif

int cnnq_pc_group_describe(int64_t N, int64_t C, int64_t HW, int32_t out[8]) {
    if (!out) return CNNQ_EINVAL;
    GPlan p;
    const int rc = plan_group(N, C, HW, true, &p, true, 1);
    if (rc) return rc;
    const int32_t vals[8] = {p.v.A, p.K + p.KL, p.g.mode, p.g.S, p.g.ncb, p.Gs, p.ngroups, p.g.S * p.g.ncb};
    for (int i = 0; i < 8; ++i) out[i] = vals[i];
    return 0;
}

int cnnq_pc_minmax_qdq_group(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                             void* ws, float* qp, float* mm, unsigned flags, void* stream) {
    if (!x || !y || !ws || !qp || num_bits < 1 || num_bits > 32 || ((uintptr_t)ws & 127)) return CNNQ_EINVAL;
    GPlan p;
    const int rc = plan_group(N, C, HW, al16(x) && al16(y), &p, true, flat_lds_rows(0, false));
    if (rc) return rc;
    return launch_group(x, y, p, num_bits, positive ? 1 : 0, ws, qp, mm, flags, (hipStream_t)stream);
}

// The two halves of config 2 around the cross-rank exchange, one call each: local extrema [2][C] of this rank's
// shard (k_minmax + k_minmax_reduce), and - after the all_gather - parameters from the W gathered records plus the
// fused Q/DQ.  pmm: workspace [G][2][C] floats; qp: workspace / output [CNNQ_NQP][C].
int cnnq_pc_minmax_local(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, float* local, void* stream) {
    if (!x || !pmm || !local) return CNNQ_EINVAL;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    const int rc = cnnq_pc_minmax(x, N, C, HW, pmm, stream);
    if (rc) return rc;
    return cnnq_pc_minmax_reduce(pmm, G, C, local, stream);
}

// ... in ONE launch when the geometry has a group plan and the caller brought the exchange workspace (k_minmax_group:
// the last workgroup of a channel group to arrive folds the group's pairs), else the two launches above
int cnnq_pc_minmax_local_auto(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, void* gws, size_t gws_bytes,
                              float* local, void* stream) {
    if (!x || !local) return CNNQ_EINVAL;
    // tensors beyond the Infinity Cache keep the streaming k_minmax (6.5-6.9 TB/s against ~5.5 for 128 KB register
    // tiles; one launch boundary is nothing next to their 60+ us)
    static const int64_t max_bytes = env_int("CNNQ_LOCAL_GROUP_MAX_MB", 384) * ((int64_t)1 << 20);   // development knob
    if (gws && !((uintptr_t)gws & 127) && N * C * HW * 4 <= max_bytes) {
        GPlan p;
        if (plan_group(N, C, HW, al16(x), &p, /*allow_flat=*/false) == 0 && p.ws_bytes <= gws_bytes)
            return launch_minmax_group(x, p, gws, local, (hipStream_t)stream);
    }
    return cnnq_pc_minmax_local(x, N, C, HW, pmm, local, stream);
}

// one launch: every workgroup of the fused Q/DQ derives the parameters of its channels from the W gathered records
int cnnq_pc_gathered_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* gathered, int W,
                         int num_bits, int positive, float* qp, void* stream) {
    if (!x || !y || !gathered || W <= 0 || num_bits < 1 || num_bits > 32) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    // the local statistics pass walked x ascending: descend, so that what it touched last is re-read first
    const int rc = plan(N, C, HW, al16(x) && al16(y), /*rev=*/1, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const GathArgs ga{W, num_bits, positive ? 1 : 0, qp};
    return launch_qdq_gathered(x, y, g, v, gathered, ga, (hipStream_t)stream);
}

// Config 2 behind ONE call: the resident single launch when the shape has one, else the group-exchange single
// launch (needs gws), else the three-launch chain.  ws layout (floats): qp[CNNQ_NQP][C], mm[2][C], pmm[G][2][C].
size_t cnnq_pc_minmax_qdq_workspace(int64_t N, int64_t C, int64_t HW) {
    const int g1 = cnnq_pc_groups(N, C, HW, 1), g0 = cnnq_pc_groups(N, C, HW, 0);
    const int G = g1 > g0 ? g1 : g0;
    if (G <= 0) return 0;
    return ((size_t)CNNQ_NQP + 2 + 2 * (size_t)G) * (size_t)C * sizeof(float);
}

int cnnq_pc_minmax_qdq_auto(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                            float* ws, void* gws, size_t gws_bytes, int allow_single_launch, void* stream) {
    if (!x || !y || !ws || num_bits < 1 || num_bits > 32 || C <= 0) return CNNQ_EINVAL;
    float* qp = ws;
    float* mm = ws + (size_t)CNNQ_NQP * C;
    float* pmm = mm + 2 * (size_t)C;
    if (allow_single_launch) {
        const size_t gneed = gws ? cnnq_pc_group_workspace(N, C, HW) : 0;
        const bool group_ok = gneed > 0 && gneed <= gws_bytes;
        // whole channels per workgroup needs no exchange, but with fewer channel blocks than ~3/4 of the CUs it leaves
        // the chip idle: [64,128,28,28] = 128 workgroups takes 17.8 us, the group form (1024 tiles) 13.4
        WPlan wp;
        const bool whole_ok = plan_whole(N, C, HW, al16(x) && al16(y), &wp) == 0;
        int rc = CNNQ_ENOTSUP;
        if (whole_ok && !(group_ok && wp.wgs < RES_MIN_WGS))
            rc = cnnq_pc_minmax_qdq_resident(x, y, N, C, HW, num_bits, positive, qp, mm, stream);
        if (rc != CNNQ_ENOTSUP) return rc;
        if (group_ok) {
            rc = cnnq_pc_minmax_qdq_group(x, y, N, C, HW, num_bits, positive, gws, qp, mm, 0u, stream);
            if (rc != CNNQ_ENOTSUP) return rc;
        }
        if (whole_ok) return cnnq_pc_minmax_qdq_resident(x, y, N, C, HW, num_bits, positive, qp, mm, stream);
    }
    return cnnq_pc_minmax_qdq(x, y, N, C, HW, num_bits, positive, pmm, qp, nullptr, nullptr, stream);
}

// Config 2 in ONE launch with the outputs the chain form used to be needed for: the uint8 codes, the code histogram
// (-me: iq.py:586-587) and / or the packed 4-bit codes INSTEAD of y (4.5 bytes per element straight from
// x).  Routing as cnnq_pc_minmax_qdq_auto, without the chain: CNNQ_ENOTSUP when the shape has no single-launch kernel.
size_t cnnq_hist_replica_bytes(void) { return (size_t)XHIST_REPLICAS * 256 * sizeof(unsigned long long); }

int cnnq_pc_minmax_qdq_single(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                              void* gws, size_t gws_bytes, float* qp, float* mm, uint8_t* codes, uint64_t* hist_rep,
                              uint8_t* packed, void* stream) {
    if (!x || !qp || num_bits < 1 || num_bits > 32 || C <= 0) return CNNQ_EINVAL;
    if (packed ? (y || codes || hist_rep || num_bits > 4 || ((uintptr_t)packed & 1)) : !y) return CNNQ_EINVAL;
    if ((codes || hist_rep) && num_bits > 8) return CNNQ_EINVAL;
    if (((uintptr_t)codes & 3) || ((uintptr_t)hist_rep & 7) || (gws && ((uintptr_t)gws & 127))) return CNNQ_EINVAL;
    const int out = packed ? 2 : (codes || hist_rep) ? 1 : 0;
    XOut xo;
    xo.codes = codes;
    xo.hist = reinterpret_cast<unsigned long long*>(hist_rep);
    xo.packed = packed;
    const bool al = al16(x) && (packed ? true : al16(y));
    GPlan gp;
    const bool group_ok = gws && plan_group(N, C, HW, al, &gp, true, flat_lds_rows(out, false)) == 0 && gp.ws_bytes <= gws_bytes;
    WPlan wp;
    const bool whole_ok = plan_whole(N, C, HW, al, &wp) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (whole_ok && !(group_ok && wp.wgs < RES_MIN_WGS))
        return launch_whole(x, y, wp, num_bits, positive ? 1 : 0, qp, mm, st, out, xo);
    if (group_ok) return launch_group(x, y, gp, num_bits, positive ? 1 : 0, gws, qp, mm, 0u, st, out, xo);
    if (whole_ok) return launch_whole(x, y, wp, num_bits, positive ? 1 : 0, qp, mm, st, out, xo);
    return CNNQ_ENOTSUP;
}

// The same single launch when the batch is sharded over `world` GPUs (opt-in; csrc/cnnq_xrank.hip.h): x is this rank's
// shard, the channel extrema are exchanged with the other ranks through `windows` INSIDE the launch, y / qp / mm are
// what a single GPU holding the whole batch would produce.  CNNQ_ENOTSUP when the shape has no single-launch kernel
// (nothing is enqueued and no sequence number is consumed: the caller takes the collective path on every rank - the
// plan depends only on the shape and the alignment of x / y, which must agree between the ranks).
size_t cnnq_xrank_window_bytes(int world, int cmax) {
    return (world > 0 && cmax > 0) ? xr_window_bytes(world, cmax) : 0;
}

int cnnq_xrank_alloc(int world, int cmax, void** window, unsigned char handle[64]) {
    if (!window || !handle || world <= 0 || cmax <= 0) return CNNQ_EINVAL;
    const size_t bytes = xr_window_bytes(world, cmax);
    hipError_t e = hipExtMallocWithFlags(window, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*window, 0, bytes);                  // an empty slot is zero
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, *window);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {                             // nothing is handed out on failure: the window is released here
        (void)hipFree(*window);
        *window = nullptr;
        return (int)e;
    }
    memcpy(handle, &h, 64);
    return 0;
}

// seq != 0: host numbering (seq_dev, if given, mirrors it; with `lean` no kernel is enqueued behind the launch: the caller passes
// zero_c, the channel count of the launch two back, and workgroup 0 cleans up); seq == 0: device numbering (k_xr_finish behind it)
static int xrank_launch(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive, float* ws, void* gws,
                        size_t gws_bytes, void* const* windows, int rank, int world, int cmax, uint32_t seq, uint32_t* seq_dev,
                        uint32_t* status, int64_t timeout_ticks, uint8_t* codes, uint64_t* hist_rep, void* stream, int zero_c = 0,
                        bool lean = false) {
    if (!x || !y || !ws || num_bits < 1 || num_bits > 32 || C <= 0) return CNNQ_EINVAL;
    if (!windows || !status || world <= 0 || rank < 0 || rank >= world || (!seq && !seq_dev) || C > cmax || timeout_ticks <= 0) return CNNQ_EINVAL;
    if (gws && ((uintptr_t)gws & 127)) return CNNQ_EINVAL;
    if ((codes || hist_rep) && num_bits > 8) return CNNQ_EINVAL;
    if (((uintptr_t)codes & 3) || ((uintptr_t)hist_rep & 7) || ((uintptr_t)seq_dev & 3)) return CNNQ_EINVAL;
    float* qp = ws;                                       // the layout of cnnq_pc_minmax_qdq_auto's workspace
    float* mm = ws + (size_t)CNNQ_NQP * C;
    float* pmm = mm + 2 * (size_t)C;
    XRank xr;
    xr.windows = windows;
    xr.rank = rank;
    xr.world = world;
    if (zero_c < 0 || zero_c > cmax || (lean && !seq)) return CNNQ_EINVAL;
    xr.seq = seq;
    xr.seq_dev = seq ? nullptr : seq_dev;
    xr.seq_mirror = (seq && lean) ? seq_dev : nullptr;
    xr.zero_c = zero_c;
    xr.cdev = seq_dev ? seq_dev + 4 : nullptr;            // round 6: the launches keep the slot counts themselves
    xr.nslots = (int)C;
    xr.slot0 = 0;
    xr.no_prologue = 0;
    xr.cmax = cmax;
    xr.status = status;
    xr.timeout = timeout_ticks;
    const int out = (codes || hist_rep) ? 1 : 0;
    XOut xo;
    xo.codes = codes;
    xo.hist = reinterpret_cast<unsigned long long*>(hist_rep);
    xo.packed = nullptr;
    const bool al = al16(x) && al16(y);
    GPlan gp;
    const bool group_ok = gws && plan_group(N, C, HW, al, &gp) == 0 && gp.ws_bytes <= gws_bytes;
    WPlan wp;
    const bool whole_ok = plan_whole(N, C, HW, al, &wp) == 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (whole_ok && !(group_ok && wp.wgs < RES_MIN_WGS)) rc = launch_whole(x, y, wp, num_bits, positive ? 1 : 0, qp, mm, st, out, xo, 0u, &xr);
    else if (group_ok) rc = launch_group(x, y, gp, num_bits, positive ? 1 : 0, gws, qp, mm, 0u, st, out, xo, &xr);
    else if (whole_ok) rc = launch_whole(x, y, wp, num_bits, positive ? 1 : 0, qp, mm, st, out, xo, 0u, &xr);
    else {
        // no single-launch kernel for this rank's shard (the ranks' shards may differ by a sample, and so may their plans):
        // the same window protocol around two passes - local extrema, one thread per channel pushes / waits / folds, Q/DQ
        // with the folded extrema as the only "gathered" record (codes / histogram: the parameter kernel + the fused Q/DQ,
        // which counts into the first replica table).  Every rank consumes the sequence number either way.
        rc = cnnq_pc_minmax_local_auto(x, N, C, HW, pmm, gws, gws_bytes, mm, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_xr_exchange, dim3((unsigned)((C + TPB - 1) / TPB)), dim3(TPB), 0, st, mm, (int)C, xr);
        rc = launch_status();
        if (rc) return rc;
        if (!out) rc = cnnq_pc_gathered_qdq(x, y, N, C, HW, mm, 1, num_bits, positive, qp, stream);
        else {
            rc = cnnq_pc_minmax_params(mm, 1, C, num_bits, positive, qp, stream);
            if (!rc) rc = cnnq_pc_qdq(x, y, N, C, HW, qp, codes, hist_rep, 1, stream);
        }
    }
    if (rc) return rc;
    if (lean) return 0;                                   // host numbering, cleaned up by workgroup 0 of the launch after next
    // behind the launch: the slots of its parity back to zero (every reader of this rank is done), the device-side launch
    // number advanced
    hipLaunchKernelGGL(k_xr_finish, dim3(1), dim3(1024), 0, st, windows, rank, world, cmax, (int)C, seq, seq ? nullptr : seq_dev,
                       seq_dev ? seq_dev + 4 : nullptr);
    return launch_status();
}

// Round 5: ONE launch per tensor on the eager path.  seq: the host's launch number (1, 2, 3, ... the same on every rank);
// zero_c: the channel count C of the launch two back on this stream (0 for the first two launches): workgroup 0 zeroes the slots
// that launch used; seq_dev (may be NULL): device word that follows the host's count, so that a later captured launch
// (cnnq_pc_minmax_qdq_xrank_dev on the same word) continues the numbering.  seq == 0: device numbering as
// cnnq_pc_minmax_qdq_xrank_dev, plus the clean-up of zero_c (the first two launches after the switch).
int cnnq_pc_minmax_qdq_xrank_seq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                                 float* ws, void* gws, size_t gws_bytes, void* const* windows, int rank, int world, int cmax,
                                 uint32_t seq, uint32_t* seq_dev, int zero_c, uint32_t* status, int64_t timeout_ticks, uint8_t* codes,
                                 uint64_t* hist_rep, void* stream) {
    if (!seq && !seq_dev) return CNNQ_EINVAL;
    return xrank_launch(x, y, N, C, HW, num_bits, positive, ws, gws, gws_bytes, windows, rank, world, cmax, seq, seq_dev, status,
                        timeout_ticks, codes, hist_rep, stream, zero_c, seq != 0);
}

int cnnq_pc_minmax_qdq_xrank(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                             float* ws, void* gws, size_t gws_bytes, void* const* windows, int rank, int world, int cmax,
                             uint32_t seq, uint32_t* status, int64_t timeout_ticks, void* stream) {
    if (!seq) return CNNQ_EINVAL;
    return xrank_launch(x, y, N, C, HW, num_bits, positive, ws, gws, gws_bytes, windows, rank, world, cmax, seq, nullptr, status,
                        timeout_ticks, nullptr, nullptr, stream);
}

int cnnq_pc_minmax_qdq_xrank_dev(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                                 float* ws, void* gws, size_t gws_bytes, void* const* windows, int rank, int world, int cmax,
                                 uint32_t* seq_dev, uint32_t* status, int64_t timeout_ticks, uint8_t* codes, uint64_t* hist_rep,
                                 void* stream) {
    if (!seq_dev) return CNNQ_EINVAL;
    return xrank_launch(x, y, N, C, HW, num_bits, positive, ws, gws, gws_bytes, windows, rank, world, cmax, 0u, seq_dev, status,
                        timeout_ticks, codes, hist_rep, stream);
}

// the replica tables of the single-launch kernels folded into one plain table hist[256] (+=: the caller zeroes it), the
// replicas left zero: a sharded run sums the ranks' tables before the entropy
int cnnq_hist_replicas_fold(uint64_t* hist_rep, uint64_t* hist, void* stream) {
    if (!hist_rep || !hist) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_hist_replicas_fold, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(hist_rep), reinterpret_cast<unsigned long long*>(hist));
    return launch_status();
}

// entropy (bits) of the replica histogram the call above filled; the tables are zero again afterwards
// n sets of replica tables, back to back (cnnq_hist_replica_bytes each), in ONE launch: out[i] = the entropy of set i; all left zero
int cnnq_entropy_replicas_batch(uint64_t* hist_rep, int n, float* out, void* stream) {
    if (!hist_rep || !out || n <= 0) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_entropy_replicas, dim3((unsigned)n), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(hist_rep), out);
    return launch_status();
}

int cnnq_entropy_replicas(uint64_t* hist_rep, float* out, void* stream) {
    if (!hist_rep || !out) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_entropy_replicas, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(hist_rep), out);
    return launch_status();
}

// The whole dynamic ACIQ pipeline (iq.py:327-352 + 409-451) behind ONE call: statistics pass A, merge, pass B
// when b is needed, merge, parameters (ACIQ clipping, bit allocation, scale / zero point), fused Q/DQ - six
// launches (five since round 2: the first merge runs inside pass B), one host call, one caller workspace.  ws layout (doubles first): part[G][NMOM][C], mom[NMOM][C],
// part2[G][NDEV][C], then floats stats[NSTAT][C].
size_t cnnq_pc_aciq_workspace(int64_t N, int64_t C, int64_t HW, int aligned16) {
    const int G = cnnq_pc_groups(N, C, HW, aligned16);
    if (G <= 0) return 0;
    return ((size_t)G * CNNQ_NMOM + CNNQ_NMOM + (size_t)G * CNNQ_NDEV) * (size_t)C * sizeof(double) +
           (size_t)CNNQ_NSTAT * (size_t)C * sizeof(float);
}

int cnnq_pc_aciq_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws,
                     float* qp, float* diag, void* stream) {
    if (!x || !y || !cfg || !ws || !qp || ((uintptr_t)ws & 7)) return CNNQ_EINVAL;
    if (cfg->num_bits < 1 || cfg->num_bits > 32 || cfg->clip < 0 || cfg->clip > 3) return CNNQ_EINVAL;
    if ((cfg->clip == 1 || cfg->clip == 2) && cfg->num_bits > 8) return CNNQ_EINVAL;   // as cnnq_pc_params, before any launch
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    double* mom = part + (size_t)G * CNNQ_NMOM * C;
    double* part2 = mom + (size_t)CNNQ_NMOM * C;
    float* stats = reinterpret_cast<float*>(part2 + (size_t)G * CNNQ_NDEV * C);
    const bool use_ba = cfg->bit_alloc && cfg->num_bits <= 4;
    const bool need_b = cfg->clip == 1 || (use_ba && cfg->prior_is_b);
    hipStream_t st = (hipStream_t)stream;
    // the merge kernels write every row of the table (zero for KURT, STD_POS; B without pass B)
    int rc = cnnq_pc_moments(x, N, C, HW, 0, part, stream);
    if (rc) return rc;
    if (need_b) {
        // pass B merges the pass-A records of its own channels in its prologue; one final merge writes all rows
        rc = absdev_raw(x, N, C, HW, part, 0, part2, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_combine_all, dim3((unsigned)((C + merge_cpw(G, (int)C) - 1) / merge_cpw(G, (int)C))), dim3(TPB), 0, st, part, part2, G,
                           (int)C, 0, 0, mom, stats);
        rc = launch_status();
    } else {
        rc = cnnq_pc_combine(part, G, C, 0, mom, stats, stream);
    }
    if (rc) return rc;
    rc = cnnq_pc_params(stats, C, cfg, qp, diag, stream);
    if (rc) return rc;
    // pass B walks the tensor descending, so the Q/DQ after it ascends; straight after pass A it descends
    return cnnq_pc_qdq(x, y, N, C, HW, qp, nullptr, nullptr, /*reverse=*/need_b ? 0 : 1, stream);
}

// Config 3 with pass B, the parameters and the Q/DQ in ONE launch (cnnq_aciq.hip.h): pass A -> merge -> (bit allocation)
// -> k_fused_flat / k_fused_group<MODE 0>: 12 instead of 16 bytes per element, four launches.  Laplace clipping on the per-channel
// route (clip == 1, no direct_range), bit allocation on the 'gaus' prior only (the 'laplace' prior IS b, which only exists
// inside the last launch).  CNNQ_ENOTSUP - nothing enqueued - for every other configuration, for shapes without a
// single-launch plan and for a `gws` too small: the caller takes cnnq_pc_aciq_qdq.  ws: part[G][CNNQ_NMOM][C] doubles
// (cnnq_pc_aciq_workspace covers it); stats [CNNQ_NSTAT][C] is written completely (KURT, STD_POS: zero).
int cnnq_pc_aciq_qdq_single(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws,
                            void* gws, size_t gws_bytes, float* stats, float* qp, float* diag, uint8_t* codes,
                            uint64_t* hist_rep, unsigned flags, void* stream) {
    if (!x || !y || !cfg || !ws || !stats || !qp || ((uintptr_t)ws & 7)) return CNNQ_EINVAL;
    if (cfg->num_bits < 1 || cfg->num_bits > 32 || cfg->clip < 0 || cfg->clip > 3) return CNNQ_EINVAL;
    if ((cfg->clip == 1 || cfg->clip == 2) && cfg->num_bits > 8) return CNNQ_EINVAL;
    if (((uintptr_t)codes & 3) || ((uintptr_t)hist_rep & 7) || (gws && ((uintptr_t)gws & 127))) return CNNQ_EINVAL;
    const bool use_ba = cfg->bit_alloc && cfg->num_bits <= 4;
    if (use_ba && !diag) return CNNQ_EINVAL;                     // the bit table lives in diag
    if (cfg->clip != 1 || cfg->direct_range || (use_ba && cfg->prior_is_b) || !gws) return CNNQ_ENOTSUP;
    GPlan gp;
    // (with the codes / the histogram wanted the 160 KB tiles of a big channel have no instance - 32 KB of LDS rows next to the 32 KB
    //  code table: planned without them, so that CNNQ_ENOTSUP comes BEFORE anything is enqueued: ADVICE r5)
    if (plan_sums(N, C, HW, al16(x) && al16(y), &gp, (codes || hist_rep) ? 0 : 1) != 0 || gp.ws_bytes > gws_bytes) return CNNQ_ENOTSUP;
    if ((size_t)gp.ngroups * gp.gstride * 8 > GRP_WS_SLOT_BYTES) return CNNQ_ENOTSUP;
    if (gp.KL && !(gp.flat && gp.K == 32 && gp.KL == 8)) return CNNQ_ENOTSUP;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    hipStream_t st = (hipStream_t)stream;
    int rc = cnnq_pc_moments(x, N, C, HW, 0, part, stream);
    if (rc) return rc;
    rc = cnnq_pc_combine(part, G, C, 0, nullptr, stats, stream);
    if (rc) return rc;
    float* bits = nullptr;
    if (use_ba) {
        bits = diag + (size_t)CNNQ_DIAG_BITS * C;
        const int threads = (int)(C >= PTPB ? PTPB : ((C + 63) / 64) * 64);
        hipLaunchKernelGGL(k_bitalloc, dim3(1), dim3(threads), 0, st, stats + (size_t)CNNQ_STAT_STD * C, (int)C, *cfg, bits);
        rc = launch_status();
    }
    if (rc) return rc;
    FusedArgs aa = {};
    aa.stats = stats;
    aa.bits = bits;
    aa.qp = qp;
    aa.diag = diag;
    aa.cfg = *cfg;
    aa.count = (double)N * (double)HW;
    XOut xo;
    xo.codes = codes;
    xo.hist = reinterpret_cast<unsigned long long*>(hist_rep);
    xo.packed = nullptr;
    return launch_fused(0, x, y, gp, aa, gws, flags & 3u, st, (codes || hist_rep) ? 1 : 0, xo);
}

// ... behind ONE call with the chain as the fallback: ws as cnnq_pc_aciq_qdq (cnnq_pc_aciq_workspace bytes); the
// statistics table is the one inside ws either way
int cnnq_pc_aciq_qdq_auto(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws,
                          void* gws, size_t gws_bytes, float* qp, float* diag, void* stream) {
    if (!x || !y || !cfg || !ws || !qp || ((uintptr_t)ws & 7)) return CNNQ_EINVAL;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    float* stats = reinterpret_cast<float*>(reinterpret_cast<double*>(ws) + ((size_t)G * CNNQ_NMOM + CNNQ_NMOM + (size_t)G * CNNQ_NDEV) * C);
    const int rc = cnnq_pc_aciq_qdq_single(x, y, N, C, HW, cfg, ws, gws, gws_bytes, stats, qp, diag, nullptr, nullptr, 0u, stream);
    if (rc != CNNQ_ENOTSUP) return rc;
    return cnnq_pc_aciq_qdq(x, y, N, C, HW, cfg, ws, qp, diag, stream);
}

// ---- round 6: configs 3 / 5 / 4 of a batch shard with the cross-rank exchange INSIDE the launches --------------------------------
// (cnnq_xrank.hip.h; the same windows, numbering and status word as config 2's cnnq_pc_minmax_qdq_xrank_seq / _dev).  One host
// call per tensor, ONE launch number, no collective: every statistic of the global batch travels through the windows.
static int xr_from_ctx(const cnnq_xrank_ctx* xc, int64_t C, XRank* xr) {
    if (!xc || !xc->windows || !xc->status || !xc->seq_dev || ((uintptr_t)xc->seq_dev & 3)) return CNNQ_EINVAL;
    if (xc->world <= 0 || xc->rank < 0 || xc->rank >= xc->world || xc->timeout_ticks <= 0 || C <= 0 || C * ST_XW > xc->cmax) return CNNQ_EINVAL;
    xr->windows = xc->windows;
    xr->rank = xc->rank;
    xr->world = xc->world;
    xr->seq = xc->seq;
    xr->seq_dev = xc->seq ? nullptr : xc->seq_dev;
    xr->seq_mirror = xc->seq ? xc->seq_dev : nullptr;
    xr->zero_c = 0;
    xr->cdev = xc->seq_dev + 4;
    xr->nslots = (int)(C * ST_XW);                       // the sums layout: eight words per channel
    xr->slot0 = (int)(C * (ST_XW_COUNT + 1));            // word 6: sum |x - mean|
    xr->no_prologue = 0;
    xr->cmax = xc->cmax;
    xr->status = xc->status;
    xr->timeout = xc->timeout_ticks;
    return 0;
}
// device numbering: the slots of the launch zeroed and the number advanced behind it (host numbering: the launch after next cleans up)
static int xr_finish_ctx(const cnnq_xrank_ctx* xc, int64_t C, hipStream_t st) {
    if (xc->seq) return 0;
    hipLaunchKernelGGL(k_xr_finish, dim3(1), dim3(1024), 0, st, xc->windows, xc->rank, xc->world, xc->cmax, (int)(C * ST_XW), 0u, xc->seq_dev,
                       xc->seq_dev + 4);
    return launch_status();
}
// pass A of a shard made global: k_moments -> k_xr_moments (the first kernel of the launch number: it runs the prologue)
static int xr_pass_a(const float* x, int64_t N, int64_t C, int64_t HW, int need_relu, double* part, int G, const XRank& xr, double* mom,
                     float* stats, void* stream) {
    int rc = cnnq_pc_moments(x, N, C, HW, need_relu, part, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_xr_moments, dim3((unsigned)((C + XS_CPB - 1) / XS_CPB)), dim3(TPB), 0, (hipStream_t)stream, part, G, (int)C, need_relu, xr, mom, stats);
    return launch_status();
}
// pass B of a shard through the chain's kernel, its sums made global: k_absdev -> k_xr_devsums (rows B / KURT of stats)
static int xr_pass_b(const float* x, int64_t N, int64_t C, int64_t HW, int nw, int want_kurt, double* part2, int G, const double* mom,
                     XRank xr, float* stats, void* stream) {
    int rc = cnnq_pc_absdev(x, N, C, HW, stats, want_kurt, part2, stream);
    if (rc) return rc;
    xr.no_prologue = 1;
    hipLaunchKernelGGL(k_xr_devsums, dim3((unsigned)((C + 31) / 32)), dim3(TPB), 0, (hipStream_t)stream, part2, G, (int)C, nw, want_kurt,
                       mom + (size_t)CNNQ_MOM_COUNT * C, xr, stats);
    return launch_status();
}

// Config 3 of a batch shard (Laplace clipping, optional bit allocation on the 'gaus' prior), the whole pipeline: k_moments ->
// k_xr_moments (the six words of the pass-A record through the windows: the table of the GLOBAL batch on every rank) ->
// (k_bitalloc) -> ONE launch for the tile's sum |x - mean|, the local slot meeting, the ranks' sums through the windows, added in
// rank order, the parameters and the Q/DQ out of the registers - the four launches of one GPU, x read twice (12 bytes per
// element), no collective.  stats [CNNQ_NSTAT][C], mom [CNNQ_NMOM][C] (fp64), qp, diag: outputs, the same on every rank.  A
// shard without a single-launch plan runs k_absdev -> k_xr_devsums -> k_params -> k_qdq around the same slots, so the ranks need
// not agree on their plans; either way the call consumes ONE launch number.  ws: cnnq_pc_aciq_workspace bytes.  CNNQ_ENOTSUP
// (nothing enqueued, no number consumed) only for configurations cnnq_pc_aciq_qdq_single refuses too.
int cnnq_pc_aciq_fused_xrank(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const cnnq_params_cfg* cfg, void* ws, void* gws,
                             size_t gws_bytes, float* stats, double* mom, float* qp, float* diag, const cnnq_xrank_ctx* xc, unsigned flags,
                             void* stream) {
    if (!x || !y || !cfg || !ws || !stats || !mom || !qp || ((uintptr_t)ws & 7) || ((uintptr_t)mom & 7)) return CNNQ_EINVAL;
    if (cfg->num_bits < 1 || cfg->num_bits > 8 || (gws && ((uintptr_t)gws & 127))) return CNNQ_EINVAL;
    const bool use_ba = cfg->bit_alloc && cfg->num_bits <= 4;
    if (use_ba && !diag) return CNNQ_EINVAL;
    if (cfg->clip != 1 || cfg->direct_range || (use_ba && cfg->prior_is_b)) return CNNQ_ENOTSUP;
    XRank xr;
    int rc = xr_from_ctx(xc, C, &xr);
    if (rc) return rc;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(ws);
    rc = xr_pass_a(x, N, C, HW, 0, part, G, xr, mom, stats, stream);
    if (rc) return rc;
    GPlan gp;
    const bool single = gws && plan_sums(N, C, HW, al16(x) && al16(y), &gp, 1) == 0 && gp.ws_bytes <= gws_bytes &&
                        (size_t)gp.ngroups * gp.gstride * 8 <= GRP_WS_SLOT_BYTES && !(gp.KL && !(gp.flat && gp.K == 32 && gp.KL == 8));
    if (single) {
        float* bits = nullptr;
        if (use_ba) {
            bits = diag + (size_t)CNNQ_DIAG_BITS * C;
            const int threads = (int)(C >= PTPB ? PTPB : ((C + 63) / 64) * 64);
            hipLaunchKernelGGL(k_bitalloc, dim3(1), dim3(threads), 0, st, stats + (size_t)CNNQ_STAT_STD * C, (int)C, *cfg, bits);
            rc = launch_status();
            if (rc) return rc;
        }
        FusedArgs aa = {};
        aa.stats = stats;
        aa.bits = bits;
        aa.qp = qp;
        aa.diag = diag;
        aa.cfg = *cfg;
        aa.count_dev = mom + (size_t)CNNQ_MOM_COUNT * C;
        XRank xb = xr;
        xb.no_prologue = 1;
        rc = launch_fused(0, x, y, gp, aa, gws, flags & 3u, st, 0, XOut{}, &xb);
    } else {
        rc = xr_pass_b(x, N, C, HW, 1, 0, part, G, mom, xr, stats, stream);
        if (!rc) rc = cnnq_pc_params(stats, C, cfg, qp, diag, stream);
        if (!rc) rc = cnnq_pc_qdq(x, y, N, C, HW, qp, nullptr, nullptr, 0, stream);
    }
    if (rc) return rc;
    return xr_finish_ctx(xc, C, st);
}

// Config 5 of a batch shard (mid-tread, clip = 1): as above with the bin allocation (k_mt_params<GUESS> on the global std) in
// place of the bit allocation and MODE 1 of the fused kernels; mt [CNNQ_NMT][C] out; hist (optional, CNNQ_MT_HIST_WORDS(C),
// zeroed here) counts THIS rank's codes - sum it over the ranks, then cnnq_midtread_entropy_count with mom's COUNT row.
int cnnq_pc_midtread_fused_xrank(const float* x, float* y, int64_t N, int64_t C, int64_t HW, double target, int sym, const double* tables,
                                 int ntab, void* ws, void* gws, size_t gws_bytes, float* stats, double* mom, float* mt, uint64_t* hist,
                                 const cnnq_xrank_ctx* xc, unsigned flags, void* stream) {
    if (!x || !y || !tables || ntab < 2 || !ws || !stats || !mom || !mt || ((uintptr_t)ws & 7) || ((uintptr_t)hist & 7) || ((uintptr_t)mom & 7)) return CNNQ_EINVAL;
    if (gws && ((uintptr_t)gws & 127)) return CNNQ_EINVAL;
    XRank xr;
    int rc = xr_from_ctx(xc, C, &xr);
    if (rc) return rc;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hist && hipMemsetAsync(hist, 0, (size_t)CNNQ_MT_HIST_WORDS(C) * sizeof(uint64_t), st) != hipSuccess) return launch_status();
    double* part = reinterpret_cast<double*>(ws);
    rc = xr_pass_a(x, N, C, HW, 0, part, G, xr, mom, stats, stream);
    if (rc) return rc;
    GPlan gp;
    const bool single = gws && plan_sums(N, C, HW, al16(x) && al16(y), &gp, 1) == 0 && gp.ws_bytes <= gws_bytes &&
                        (size_t)gp.ngroups * gp.gstride * 8 <= GRP_WS_SLOT_BYTES && (gp.flat || gp.v.A == 1) &&
                        !(gp.KL && !(gp.flat && gp.K == 32 && gp.KL == 8));
    const MtCfg mcfg{target, 1, sym ? 1 : 0};
    if (single) {
        hipLaunchKernelGGL(k_mt_params<true>, dim3(1), dim3(PTPB), 0, st, stats, (int)C, mcfg, tables, ntab, mt);
        rc = launch_status();
        if (rc) return rc;
        FusedArgs fa = {};
        fa.stats = stats;
        fa.count_dev = mom + (size_t)CNNQ_MOM_COUNT * C;
        fa.mt = mt;
        fa.mcfg = mcfg;
        fa.hist = reinterpret_cast<unsigned long long*>(hist);
        XRank xb = xr;
        xb.no_prologue = 1;
        rc = launch_fused(1, x, y, gp, fa, gws, flags & 3u, st, hist ? 1 : 0, XOut{}, &xb);
    } else {
        rc = xr_pass_b(x, N, C, HW, 1, 0, part, G, mom, xr, stats, stream);
        if (!rc) rc = cnnq_pc_midtread_params(stats, C, target, 1, sym, tables, ntab, mt, stream);
        if (!rc) rc = cnnq_pc_midtread_qdq(x, y, N, C, HW, mt, 1, nullptr, hist, stream);
    }
    if (rc) return rc;
    return xr_finish_ctx(xc, C, st);
}

// Config 4 of a batch shard: the seven statistics of the GLOBAL batch from ONE read of this rank's shard (k_stats_flat with the
// cross-rank stage: both phases' folds exchanged inside the launch; eight slots per channel, 8 C <= cmax).  stats [CNNQ_NSTAT][C]
// and mom [CNNQ_NMOM][C] are the global batch's on every rank.  A shard without a flat-tile plan (or with more than 256 tiles per
// channel) runs the chain's two passes with their records made global by k_xr_moments / k_xr_devsums around the same slots -
// four launches, no collective (ws: cnnq_pc_stats_workspace bytes).  ONE launch number per call.
int cnnq_pc_stats_xrank(const float* x, int64_t N, int64_t C, int64_t HW, int need_b, int need_kurt, int need_relu, void* ws, void* gws,
                        size_t gws_bytes, double* mom, float* stats, const cnnq_xrank_ctx* xc, unsigned flags, void* stream) {
    if (!x || !stats || !mom || !ws || ((uintptr_t)ws & 7) || (gws && ((uintptr_t)gws & 127)) || ((uintptr_t)mom & 7)) return CNNQ_EINVAL;
    XRank xr;
    int rc = xr_from_ctx(xc, C, &xr);
    if (rc) return rc;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int need_dev = (need_b || need_kurt) ? 1 : 0;
    GPlan gp;
    const bool planned = gws && plan_sums(N, C, HW, al16(x), &gp, 0) == 0 && gp.ws_bytes <= gws_bytes;
    const bool single = planned && gp.flat && !gp.KL && (gp.Gs <= ST_MAX_MEMBERS || (flags & 8u)) &&
                        (size_t)gp.ngroups * gp.gstride * ST_LINE * 8 <= GRP_WS_SLOT_BYTES;
    St1Args sa;
    sa.stats = stats;
    sa.mom = mom;
    sa.count = (double)N * (double)HW;                     // this rank's; the launch exchanges it with the sums
    sa.need_relu = need_relu ? 1 : 0;
    sa.need_dev = need_dev;
    sa.need_kurt = need_kurt ? 1 : 0;
    rc = CNNQ_ENOTSUP;
    if (single) {
        rc = launch_stats_flat(x, gp, sa, gws, flags & 1u, false, st, &xr);
    } else if (planned && !gp.flat && ((flags & 8u) || stats_group_pays(gp, N, C, HW))) {
        rc = launch_stats_group(x, gp, sa, gws, gws_bytes, flags & 1u, st, &xr);       // CNNQ_ENOTSUP (nothing enqueued): the slots do not fit
    }
    if (rc == CNNQ_ENOTSUP) {
        double* part = reinterpret_cast<double*>(ws);
        double* part2 = part + (size_t)G * CNNQ_NMOM * C;
        rc = xr_pass_a(x, N, C, HW, need_relu ? 1 : 0, part, G, xr, mom, stats, stream);
        if (!rc && need_dev) rc = xr_pass_b(x, N, C, HW, 2, need_kurt ? 1 : 0, part2, G, mom, xr, stats, stream);
    }
    if (rc) return rc;
    return xr_finish_ctx(xc, C, st);
}

int cnnq_pc_weight_correct(float* wq, int64_t C, int64_t HW, const float* stats_w, const float* stats_q, int vcorr,
                           int bcorr, void* stream) {
    if (!wq || !stats_w || !stats_q || C <= 0 || HW <= 0 || C > 65535 * 1024 || HW >= ((int64_t)1 << 31))
        return CNNQ_EINVAL;
    if (C > 65535) return CNNQ_ERANGE;
    int64_t bx = (HW + TPB - 1) / TPB;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_weight_correct, dim3((unsigned)bx, (unsigned)C), dim3(TPB), 0, (hipStream_t)stream, wq, (int)C,
                       (int)HW, stats_w, stats_q, vcorr, bcorr);
    return launch_status();
}

int cnnq_pc_bcorr_sums(const float* x, const float* y, int64_t N, int64_t C, int64_t HW, int relu_first,
                       double* part3, void* stream) {
    if (!x || !y || !part3) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x) && al16(y), 0, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BS(VEC, A, J) \
    hipLaunchKernelGGL((k_bcorr_sums<VEC, A, J, false, false>), grid, block, 0, st, x, y, g, relu_first, nullptr, part3)
    CNNQ_DISPATCH(v, LAUNCH_BS);
#undef LAUNCH_BS
    return launch_status();
}

int cnnq_pc_qdq_bcorr_sums(const float* x, int64_t N, int64_t C, int64_t HW, const float* qp, int relu_first,
                           double* part3, void* stream) {
    if (!x || !qp || !part3) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), 0, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_BS(VEC, A, J)                                                                                          \
    do {                                                                                                              \
        if (ntl)                                                                                                      \
            hipLaunchKernelGGL((k_bcorr_sums<VEC, A, J, true, true>), grid, block, 0, st, x, nullptr, g, relu_first,  \
                               qp, part3);                                                                            \
        else                                                                                                          \
            hipLaunchKernelGGL((k_bcorr_sums<VEC, A, J, true, false>), grid, block, 0, st, x, nullptr, g, relu_first, \
                               qp, part3);                                                                            \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_BS);
#undef LAUNCH_BS
    return launch_status();
}

int cnnq_pc_qdq_bcorr(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* qp, const float* bias,
                      int reverse, void* stream) {
    if (!x || !y || !qp || !bias) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x) && al16(y), reverse != 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_QB(VEC, A, J) hipLaunchKernelGGL((k_qdq_bias<VEC, A, J>), grid, block, 0, st, x, y, g, qp, bias)
    CNNQ_DISPATCH(v, LAUNCH_QB);
#undef LAUNCH_QB
    return launch_status();
}

int cnnq_pc_bcorr_bias(const double* part3, int G, int64_t C, double* sums, float* bias, void* stream) {
    if (!part3 || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || (!sums && !bias)) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_bcorr_bias, dim3((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0, (hipStream_t)stream, part3, G,
                       (int)C, sums, bias);
    return launch_status();
}

int cnnq_pc_bcorr_apply(float* y, int64_t N, int64_t C, int64_t HW, const float* bias, void* stream) {
    if (!y || !bias) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(y), 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BA(VEC, A, J) hipLaunchKernelGGL((k_bcorr_apply<VEC, A, J>), grid, block, 0, st, y, g, bias)
    CNNQ_DISPATCH(v, LAUNCH_BA);
#undef LAUNCH_BA
    return launch_status();
}

int cnnq_pc_midtread_params(const float* stats, int64_t C, double target, int clip, int sym, const double* tables,
                            int ntab, float* mt, void* stream) {
    if (!stats || !mt || !tables || ntab < 2 || C <= 0 || C >= ((int64_t)1 << 31)) return CNNQ_EINVAL;
    const MtCfg cfg{target, clip ? 1 : 0, sym ? 1 : 0};
    hipLaunchKernelGGL(k_mt_params<false>, dim3(1), dim3(PTPB), 0, (hipStream_t)stream, stats, (int)C, cfg, tables, ntab, mt);
    return launch_status();
}

int cnnq_pc_midtread_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* mt, int clip,
                         float* codes, uint64_t* hist, void* stream) {
    if (!x || !y || !mt) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    // short tiles in address order (14 KB); 56 KB with the histogram, whose per-workgroup flush - one global atomic per
    // live bin - wants fewer workgroups (swept 14 .. 448 KB on VGG-16 b512: 21.7 / 20.2 / 19.4 / 19.5 / 20.3 / 20.9 ms)
    const int rc = plan(N, C, HW, al16(x) && al16(y) && (!codes || al16(codes)), 0, &v, &g, /*fine=*/hist ? 57344 : 1);
    if (rc) return rc;
    const int total = g.S * g.ncb;
    const int wgs = total;
    const dim3 grid((unsigned)wgs), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(hist);
#define LAUNCH_MT3(VEC, A, J, CL, HI)                                                                                   \
    do {                                                                                                               \
        if (codes) hipLaunchKernelGGL((k_mt_qdq<VEC, A, J, CL, HI, true>), grid, block, 0, st, x, y, g, mt, codes, h, total); \
        else hipLaunchKernelGGL((k_mt_qdq<VEC, A, J, CL, HI, false>), grid, block, 0, st, x, y, g, mt, codes, h, total);      \
    } while (0)
#define LAUNCH_MT(VEC, A, J)                                      \
    do {                                                          \
        if (clip && h) LAUNCH_MT3(VEC, A, J, true, true);         \
        else if (clip) LAUNCH_MT3(VEC, A, J, true, false);        \
        else if (h) LAUNCH_MT3(VEC, A, J, false, true);           \
        else LAUNCH_MT3(VEC, A, J, false, false);                 \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_MT);
#undef LAUNCH_MT
#undef LAUNCH_MT3
    return launch_status();
}

// Config 5 with pass B, the step sizes / clamp bounds and the quantization in ONE launch (cnnq_aciq.hip.h, MODE 1): pass A ->
// merge -> k_mt_params<GUESS> (omega and the clipping multiplier need nothing but the std) -> k_fused_flat / k_fused_group:
// 12 instead of 16 bytes per element.  clip = 1 only (without clipping nothing needs pass B).  hist (optional,
// CNNQ_MT_HIST_WORDS(C) words) is zeroed here.  CNNQ_ENOTSUP - nothing enqueued - for shapes without a single-launch plan or a
// gws that is NULL / too small: the caller takes pc_stats -> cnnq_pc_midtread_params -> cnnq_pc_midtread_qdq.
int cnnq_pc_midtread_qdq_single(const float* x, float* y, int64_t N, int64_t C, int64_t HW, double target, int sym,
                                const double* tables, int ntab, void* ws, void* gws, size_t gws_bytes, float* stats, float* mt,
                                uint64_t* hist, unsigned flags, void* stream) {
    if (!x || !y || !tables || ntab < 2 || !ws || !stats || !mt || ((uintptr_t)ws & 7) || ((uintptr_t)hist & 7)) return CNNQ_EINVAL;
    if (gws && ((uintptr_t)gws & 127)) return CNNQ_EINVAL;
    if (!gws) return CNNQ_ENOTSUP;
    GPlan gp;
    if (plan_sums(N, C, HW, al16(x) && al16(y), &gp, 1) != 0 || gp.ws_bytes > gws_bytes) return CNNQ_ENOTSUP;
    if ((size_t)gp.ngroups * gp.gstride * 8 > GRP_WS_SLOT_BYTES || (!gp.flat && gp.v.A != 1)) return CNNQ_ENOTSUP;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    hipStream_t st = (hipStream_t)stream;
    if (hist && hipMemsetAsync(hist, 0, (size_t)CNNQ_MT_HIST_WORDS(C) * sizeof(uint64_t), st) != hipSuccess) return launch_status();
    int rc = cnnq_pc_moments(x, N, C, HW, 0, part, stream);
    if (rc) return rc;
    rc = cnnq_pc_combine(part, G, C, 0, nullptr, stats, stream);
    if (rc) return rc;
    const MtCfg mcfg{target, 1, sym ? 1 : 0};
    hipLaunchKernelGGL(k_mt_params<true>, dim3(1), dim3(PTPB), 0, st, stats, (int)C, mcfg, tables, ntab, mt);
    rc = launch_status();
    if (rc) return rc;
    FusedArgs fa = {};
    fa.stats = stats;
    fa.count = (double)N * (double)HW;
    fa.mt = mt;
    fa.mcfg = mcfg;
    fa.hist = reinterpret_cast<unsigned long long*>(hist);
    return launch_fused(1, x, y, gp, fa, gws, flags & 3u, st, hist ? 1 : 0, XOut{});
}

int cnnq_midtread_entropy(const uint64_t* hist, const float* mt, int64_t C, int64_t total, float* out, void* stream) {
    if (!hist || !mt || !out || C <= 0 || total <= 0) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_mt_entropy, dim3(1), dim3(PTPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(hist), mt, (int)C, (double)total, out);
    return launch_status();
}

// the same with the element count taken from device memory: count[0] elements per channel (row CNNQ_MOM_COUNT of the merged
// moment record of a batch-sharded run, whose global batch size only the device knows exactly - shards may differ by a sample)
int cnnq_midtread_entropy_count(const uint64_t* hist, const float* mt, int64_t C, const double* count, float* out, void* stream) {
    if (!hist || !mt || !out || !count || ((uintptr_t)count & 7) || C <= 0) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_mt_entropy, dim3(1), dim3(PTPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(hist), mt, (int)C, 0., out, count);
    return launch_status();
}

// the entropies of n histograms of the mid-tread path in ONE launch (n <= 16; the arrays are host arrays of n entries)
int cnnq_midtread_entropy_batch(int n, const uint64_t* const* hist, const float* const* mt, const int64_t* C, const int64_t* total,
                                float* out, void* stream) {
    if (n <= 0 || n > MT_ENT_BATCH || !hist || !mt || !C || !total || !out) return CNNQ_EINVAL;
    MtEntBatch b = {};
    for (int i = 0; i < n; ++i) {
        if (!hist[i] || !mt[i] || C[i] <= 0 || total[i] <= 0) return CNNQ_EINVAL;
        b.hist[i] = reinterpret_cast<const unsigned long long*>(hist[i]);
        b.mt[i] = mt[i];
        b.C[i] = (int)C[i];
        b.total[i] = (double)total[i];
    }
    hipLaunchKernelGGL(k_mt_entropy_batch, dim3((unsigned)n), dim3(PTPB), 0, (hipStream_t)stream, b, out);
    return launch_status();
}

int cnnq_entropy(const uint64_t* hist, int nbins, float* out, void* stream) {
    if (!hist || !out || nbins <= 0) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_entropy, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(hist), nbins, out);
    return launch_status();
}

int cnnq_pt_setup(const float* range_offset_host, const float* stats, int64_t stats_stride, int rows, int rows_mode,
                  int zero_min, int num_bits, int int_exp, int enforce_true_zero, float* ptp, void* stream) {
    if (!ptp || num_bits < 1 || num_bits > 31) return CNNQ_EINVAL;
    if (!range_offset_host && (!stats || rows <= 0 || stats_stride < rows)) return CNNQ_EINVAL;
    const float hr = range_offset_host ? range_offset_host[0] : 0.f;
    const float ho = range_offset_host ? range_offset_host[1] : 0.f;
    hipLaunchKernelGGL(k_pt_setup, dim3(1), dim3(64), 0, (hipStream_t)stream, range_offset_host ? 1 : 0, hr, ho,
                       stats, stats_stride, rows, rows_mode, zero_min, num_bits, int_exp, enforce_true_zero, ptp);
    return launch_status();
}

// config 1 behind one call AND one launch (k_pt_fused): x viewed as [rows][n / rows]; rows_mode 0: batch mean of the
// per-row extrema (conv activations, iq.py:515-526), 1: the tensor's extrema.  gws: the exchange workspace of
// cnnq_group_ws_alloc (its header region: an epoch word and row records that every launch writes
// before it reads - nothing there has to be zero - and the counter lines, zero between launches).  ptp_out (may be NULL): the eight parameters
// cnnq_pt_setup would have written.  CNNQ_ENOTSUP: shapes the kernel does not take (rows not whole float4s, more than
// 1024 rows, unaligned pointers, more 16 KB tiles than the workspace has records for) - use cnnq_pc_minmax + cnnq_pc_minmax_reduce + cnnq_pt_setup + cnnq_pt_qdq.
int cnnq_pt_minmax_qdq_fused(const float* x, float* y, int64_t n, int rows, int rows_mode, int zero_min, int num_bits,
                             int int_exp, int enforce_true_zero, void* gws, size_t gws_bytes, float* ptp_out, void* stream) {
    if (!x || !y || !gws || gws_bytes < GRP_WS_PAIRS || n <= 0 || rows <= 0 || num_bits < 1 || num_bits > 31 || ((uintptr_t)gws & 127)) return CNNQ_EINVAL;
    if (n % rows) return CNNQ_EINVAL;
    const int64_t L = n / rows;
    if (L % 4 || rows > PTF_MAX_ROWS || !al16(x) || !al16(y) || L / 4 >= ((int64_t)1 << 31)) return CNNQ_ENOTSUP;
    const int64_t L4 = L / 4;
    const int64_t tpr = (L4 + PTF_TILE4 - 1) / PTF_TILE4;
    if (tpr * rows >= ((int64_t)1 << 31) - 64) return CNNQ_ENOTSUP;
    if ((size_t)(tpr * rows) * sizeof(uint4) > gws_bytes - GRP_WS_PAIRS) return CNNQ_ENOTSUP;     // one record per tile
    char* base = reinterpret_cast<char*>(gws) + PTF_OFF;
    PtfWs w;
    w.status = reinterpret_cast<unsigned*>(gws);
    w.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(gws) + GRP_WS_HDR);
    w.epoch = reinterpret_cast<unsigned*>(base);
    w.rows = reinterpret_cast<unsigned*>(base + 256);
    w.recs = reinterpret_cast<uint4*>(reinterpret_cast<char*>(gws) + GRP_WS_PAIRS);   // write-before-read: garbage-tolerant
    static_assert(PTF_OFF + 256 + (size_t)PTF_MAX_ROWS * 16 <= GRP_WS_HDR, "the fused per-tensor region must fit the header");
    const int cus = chip_cus();
    // a co-resident grid: 4 workgroups per CU of the 5 that fit (83 VGPRs, 13 KB of LDS), fewer when the tensor is small
    const int64_t ntiles = tpr * rows;
    int64_t G = (int64_t)cus * 4;
    if (G > GRP_GS_MAX * 2) G = GRP_GS_MAX * 2;
    if (G > ntiles) G = ntiles;
    const int64_t per = (ntiles + G - 1) / G;
    G = (ntiles + per - 1) / per;
    if (grp_lines_per_group_host((int)G) > GRP_MAX_LINES) return CNNQ_ENOTSUP;
    const dim3 grid((unsigned)G), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    // beyond the Infinity Cache the first sweep leaves nothing behind for the second: stream it
    if (n * 4 > ((int64_t)192 << 20))
        hipLaunchKernelGGL(k_pt_fused<true>, grid, block, 0, st, x, y, rows, (unsigned)L4, (unsigned)tpr, (unsigned)per, w,
                           rows_mode, zero_min, num_bits, int_exp, enforce_true_zero, ptp_out);
    else
        hipLaunchKernelGGL(k_pt_fused<false>, grid, block, 0, st, x, y, rows, (unsigned)L4, (unsigned)tpr, (unsigned)per, w,
                           rows_mode, zero_min, num_bits, int_exp, enforce_true_zero, ptp_out);
    return launch_status();
}

int cnnq_pt_qdq(const float* x, float* y, int64_t n, const float* ptp, const float* noise, void* stream) {
    if (!x || !y || !ptp || n <= 0) return CNNQ_EINVAL;
    const bool vec = al16(x) && al16(y) && (!noise || al16(noise));
    const int64_t work = vec ? (n + 3) / 4 : n;
    const int64_t blocks = (work + TPB - 1) / TPB;
    if (blocks >= ((int64_t)1 << 31)) return CNNQ_ERANGE;
    const dim3 grid((unsigned)blocks), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        if (noise) hipLaunchKernelGGL((k_pt_qdq<4, true>), grid, block, 0, st, x, y, n, ptp, noise);
        else hipLaunchKernelGGL((k_pt_qdq<4, false>), grid, block, 0, st, x, y, n, ptp, noise);
    } else {
        if (noise) hipLaunchKernelGGL((k_pt_qdq<1, true>), grid, block, 0, st, x, y, n, ptp, noise);
        else hipLaunchKernelGGL((k_pt_qdq<1, false>), grid, block, 0, st, x, y, n, ptp, noise);
    }
    return launch_status();
}

int cnnq_kld_hist(const float* x, int64_t rows, int64_t len, const float* rowmm, uint32_t* hist, void* stream) {
    if (!x || !rowmm || !hist || rows <= 0 || len <= 0) return CNNQ_EINVAL;
    if (rows > 65535 || len >= ((int64_t)1 << 31)) return CNNQ_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(hist, 0, (size_t)rows * KB * sizeof(uint32_t), st) != hipSuccess) return launch_status();
    const dim3 grid((unsigned)((len + KCHUNK - 1) / KCHUNK), (unsigned)rows), block(TPB);
    if (al16(x) && len % 4 == 0)
        hipLaunchKernelGGL((k_kld_hist<4>), grid, block, 0, st, x, len, rowmm, (int)rows, hist);
    else
        hipLaunchKernelGGL((k_kld_hist<1>), grid, block, 0, st, x, len, rowmm, (int)rows, hist);
    return launch_status();
}

int cnnq_kld_search(const uint32_t* hist, int64_t rows, const float* rowmm, double* div, double* out, void* stream) {
    if (!hist || !rowmm || !div || !out || rows <= 0) return CNNQ_EINVAL;
    if (rows > 65535) return CNNQ_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_kld_search, dim3(KC, (unsigned)rows), dim3(TPB), 0, st, hist, div);
    hipLaunchKernelGGL(k_kld_pick, dim3((unsigned)rows), dim3(64), 0, st, div, rowmm, (int)rows, out);
    return launch_status();
}

// ---- the windows of the in-launch cross-rank exchange: a peer's window mapped from its hipIpc handle, unmapped, the own
//      one released (with cnnq_xrank_alloc the only entry points of the multi-GPU path that allocate or synchronise)
int cnnq_xrank_open(const unsigned char handle[64], void** window) {
    if (!handle || !window) return CNNQ_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    return (int)hipIpcOpenMemHandle(window, h, hipIpcMemLazyEnablePeerAccess);
}
int cnnq_xrank_close(void* window) { return window ? (int)hipIpcCloseMemHandle(window) : CNNQ_EINVAL; }
int cnnq_xrank_free(void* window) { return window ? (int)hipFree(window) : CNNQ_EINVAL; }

}  // extern "C"
