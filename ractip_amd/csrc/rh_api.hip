// rh_api.hip -- context, batch management and the C ABI of include/ractip_hot.h.
//
// Host side of the drop-in boundary.  One rh_ctx owns one
// GPU's streams, the score model in HBM and the DP tables of the current batch;
// tables are kept and reused across batches of equal or smaller shape (the z-score
// loop, /root/reference/src/ractip.cpp:1638-1657, shuffles preserve lengths).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ractip_hot.h"
#include "batch.h"
#include "score_model.h"
#include "lin_model.h"
#include "vienna_model.h"

namespace rh {
__global__ void mc_init(McBatch B);
__global__ void mc_inside_diag(McBatch B, const ScoreModel* __restrict__ M, int d, int pin);
__global__ void mc_outside_diag(McBatch B, const ScoreModel* __restrict__ M, int d, int pin);
__global__ void mc_unpaired(McBatch B);
__global__ void dx_sweep_diag(DxBatch B, const ScoreModel* __restrict__ M, int t);
__global__ void dx_logz(DxBatch B, const ScoreModel* __restrict__ M);
__global__ void dx_posterior(DxBatch B);
__global__ void lin_init(McBatch B, const LinModel* __restrict__ L, int* __restrict__ bad);
template <int W, int BS, int MODE> __global__ void lin_inside_diag(McBatch B, const LinModel* __restrict__ L, int d, double lam_d, int pin);
template <int W, int BS> __global__ void lin_outside_diag(McBatch B, const LinModel* __restrict__ L, int d, int pin, int* __restrict__ bad);
template <int W, int BS> __global__ void lin_outside_pair(McBatch B, const LinModel* __restrict__ L, int d, int khi, int pin, int* __restrict__ bad);
template <int BS> __global__ void lin_far_inside(McBatch B, int D);
template <int BS> __global__ void lin_far_outside(McBatch B, int D);
__global__ void lin_far_inside_mfma(McBatch B, int D);
__global__ void lin_far_outside_mfma(McBatch B, int D);
template <int SWEEP> __global__ void lin_pack_tiles(McBatch B, int Dblk, int outside, int banded);
template <int KD, int W, int FILT> __global__ void lin_inside_strip(McBatch B, const LinModel* __restrict__ L, const double* __restrict__ wT, int d0, int f5_lo, double lam_d0, int pin);
template <int KD, int W, int FILT> __global__ void lin_outside_strip(McBatch B, const LinModel* __restrict__ L, const double* __restrict__ wT, int d0, int f5_hi, int f5_lo, int pin, int* __restrict__ bad);
__global__ void lin_f5i_tail(McBatch B, const LinModel* __restrict__ L, int jlo);
void launch_lin_small(const McBatch& B, const LinModel* L, const double* wpad, const int* list, int nlist, int* bad, hipStream_t stream);   // mccaskill_small.hip
__global__ void lin_f5o_head(McBatch B, const LinModel* __restrict__ L, int khi, int klo);
__global__ void lin_far_inside_pk(McBatch B, int D, int l2);
__global__ void lin_far_outside_pk(McBatch B, int D, int l2);
__global__ void lin_far2_inside(McBatch B, int D2, int l2);
__global__ void lin_far2_outside(McBatch B, int D2, int l2);
__global__ void lin_finish(McBatch B, const LinModel* __restrict__ L, double* __restrict__ logz, int* __restrict__ bad);
template <int W> __global__ void dxl_sweep(DxLinBatch B, const DxLinModel* __restrict__ L, int step, int groups);
__global__ void dxl_sweep4(DxLinBatch B, const DxLinModel* __restrict__ L, int step, int groups);
__global__ void dxl_strip8(DxLinBatch B, const DxLinModel* __restrict__ L, int step);
__global__ void dxvl_sweep4(DxLinBatch B, const VLinModel* __restrict__ L, const VDxLin* __restrict__ D, int step);
__global__ void dxvl_logz_part(DxLinBatch B, const VLinModel* __restrict__ L, const VDxLin* __restrict__ D, double* __restrict__ zpart,
                               int* __restrict__ cpart, int nchunk);
__global__ void dxvl_logz_final(DxLinBatch B, double s, const double* __restrict__ zpart, const int* __restrict__ cpart, int nchunk,
                                double* __restrict__ zbar, double* __restrict__ logz, int* __restrict__ bad);
__global__ void dxl_logz_part(DxLinBatch B, const DxLinModel* __restrict__ L, double* __restrict__ zpart, int* __restrict__ cpart, int nchunk);
__global__ void dxl_logz_final(DxLinBatch B, const DxLinModel* __restrict__ L, const double* __restrict__ zpart, const int* __restrict__ cpart, int nchunk,
                               double* __restrict__ zbar, double* __restrict__ logz, int* __restrict__ bad);
__global__ void dxl_posterior(DxLinBatch B, const double* __restrict__ zbar, int* __restrict__ bad);
__global__ void dxv_sweep_diag(DxBatch B, const ViennaDx* __restrict__ V, int t);
__global__ void dxv_logz(DxBatch B, const ViennaDx* __restrict__ V);
__global__ void dxv_posterior(DxBatch B);
__global__ void mcv_init(McBatch B);
__global__ void mcv_inside_diag(McBatch B, const ViennaDx* __restrict__ V, int d, int pin);
__global__ void mcv_outside_diag(McBatch B, const ViennaDx* __restrict__ V, int d, int pin);
__global__ void mcv_acc_prep(McBatch B, const ViennaDx* __restrict__ V);
__global__ void mcv_acc_hscan(McBatch B, int slot);
__global__ void vlin_acc_prep(McBatch B, const VLinModel* __restrict__ L, const double* __restrict__ hplen);
__global__ void vlin_acc_gaps(McBatch B, const VLinModel* __restrict__ L, double* __restrict__ gaps, int ng, int nchunk, double* __restrict__ part);
__global__ void vlin_acc_gsum(McBatch B, double* __restrict__ gaps, const double* __restrict__ part, int ng, int nchunk);
__global__ void vlin_acc_gaps_wide(McBatch B, const VLinModel* __restrict__ L, double* __restrict__ gaps);
__global__ void vlin_acc_final_t(McBatch B, const VLinModel* __restrict__ L, const double* __restrict__ gaps, int max_w);
__global__ void vlin_acc_hsum(McBatch B, int max_w);
__global__ void vlin_acc_gsuf(McBatch B, double* __restrict__ gaps);
__global__ void vlin_acc_final(McBatch B, const VLinModel* __restrict__ L, const double* __restrict__ gaps, int max_w);
__global__ void mcv_acc_gaps(McBatch B, const ViennaDx* __restrict__ V, double* __restrict__ gaps);
__global__ void mcv_acc_final(McBatch B, const ViennaDx* __restrict__ V, const double* __restrict__ gaps, int max_w);
__global__ void mcv_finish(McBatch B, double* __restrict__ logz);
__global__ void vlin_init(McBatch B, int* __restrict__ bad);
__global__ void vlin_co_seed(McBatch B, McBatch S);
template <int W, int BS, bool CUT, int MODE> __global__ void vlin_inside_diag(McBatch B, const VLinModel* __restrict__ L, int d, double hp_d, int pin);
template <int W, int BS, bool CUT, int MODE> __global__ void vlin_outside_diag(McBatch B, const VLinModel* __restrict__ L, int d, int pin, int* __restrict__ bad);
__global__ void vlin_finish(McBatch B, const VLinModel* __restrict__ L, double* __restrict__ logz, int* __restrict__ bad);
__global__ void mcv_extract_hp(McBatch B, double* __restrict__ hp, size_t hp_stride, int ldd, double* __restrict__ logz, double lin_s, int* __restrict__ bad);
}  // namespace rh

using namespace rh;

// out[3p..3p+2] = F5i[n] of sequences 2p, 2p+1 and the duplex logZ of pair p
__global__ void collect_logz(const double* __restrict__ mc_logz, DxBatch D, double* __restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.np) return;
    out[3 * p + 0] = mc_logz[2 * p];
    out[3 * p + 1] = mc_logz[2 * p + 1];
    out[3 * p + 2] = D.logz[p];
}

// log-space path: logZ = F5i[n] (InferenceEngine.ipp:4089-4094)
__global__ void log_finish(McBatch B, double* __restrict__ logz)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq < B.ns) logz[sq] = B.f5i[(size_t)sq * B.ld + B.n[sq]];
}

// ---- ordered threshold compaction (the scans of /root/reference/src/ractip.cpp:557-568, 578-589,
//      598-608, 621-627 done on device): one wavefront per matrix row; probabilities are narrowed
//      to float BEFORE the comparison, as the reference's VF containers do (src/ractip.cpp:82-83).
struct CandView {
    const double* base;
    int kind;   // 0 = bp triangle (row i: j = i+1..n), 1 = hp matrix (row i: j = 1..n2), 2 = up vector
    int n, n2, ld;
};
struct RowSpan {
    const double* p;  // row base: element j of the row is p[j]
    int i, j0, j1;
    bool ok;
};
// kind 2 (up): n2 = max_w, the row is the whole n x max_w matrix, entry j = position*max_w + width index
__device__ __forceinline__ RowSpan cand_row(const double* base, int kind, int n, int n2, int ld, int row)
{
    RowSpan r;
    if (kind == 0) {
        r.i = row + 1; r.j0 = r.i + 1; r.j1 = n;
        r.p = base + (size_t)r.i * (size_t)(2 * (n + 1) - r.i - 1) / 2;
        r.ok = r.i <= n;
    } else if (kind == 1) {
        r.i = row + 1; r.j0 = 1; r.j1 = n2;
        r.p = base + (size_t)r.i * (size_t)ld;
        r.ok = r.i <= n;
    } else {
        r.i = 0; r.j0 = 0; r.j1 = n * n2 - 1; r.p = base;
        r.ok = row == 0;
    }
    return r;
}
__global__ __launch_bounds__(256) void cand_count(const double* __restrict__ base, int kind, int n, int n2, int ld, float th,
                                                  int nrows, int* __restrict__ counts)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const RowSpan r = cand_row(base, kind, n, n2, ld, row);
    if (!r.ok) return;
    int c = 0;
    for (int j = r.j0 + lane; j <= r.j1; j += 64) c += ((float)r.p[j] > th) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) counts[row] = c;
}
__global__ __launch_bounds__(256) void cand_write(const double* __restrict__ base, int kind, int n, int n2, int ld, float th,
                                                  int nrows, const int* __restrict__ offsets, rh_cand* __restrict__ out, int cap)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const RowSpan r = cand_row(base, kind, n, n2, ld, row);
    if (!r.ok) return;
    int pos = offsets[row];
    for (int jb = r.j0; jb <= r.j1; jb += 64) {
        const int j = jb + lane;
        const float pf = j <= r.j1 ? (float)r.p[j] : 0.0f;
        const bool hit = j <= r.j1 && pf > th;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int k = pos + __popcll(m & ((1ull << lane) - 1ull));
            if (k < cap) {
                rh_cand e;
                e.i = kind == 2 ? j / n2 : r.i;
                e.j = kind == 2 ? j % n2 : j;
                e.p = pf;
                out[k] = e;
            }
        }
        pos += __popcll(m);
    }
}

// ---- the same compaction for every pair of the batch at once: row index = p*rmax + r
struct CandAll {
    const double* bp; const double* hp; const double* up;
    const int* n;            // [2*np]
    size_t tri_stride, hp_stride;
    int up_ld, hp_ld, which, rmax, np;
};
__device__ __forceinline__ RowSpan cand_row_all(const double* bp, const double* hp, const double* up, const int* __restrict__ nn,
                                                size_t tri_stride, size_t hp_stride, int up_ld, int hp_ld, int which, int p, int r)
{
    if (which <= 1) {
        const int sq = 2 * p + which;
        return cand_row(bp + (size_t)sq * tri_stride, 0, nn[sq], 0, 0, r);
    }
    if (which == 2) return cand_row(hp + (size_t)p * hp_stride, 1, nn[2 * p], nn[2 * p + 1], hp_ld, r);
    const int sq = 2 * p + (which - 3);
    return cand_row(up + (size_t)sq * up_ld, 2, nn[sq], hp_ld /* = max_w for the up scans */, 0, r);
}
__global__ __launch_bounds__(256) void cand_count_all(const double* __restrict__ bp, const double* __restrict__ hp, const double* __restrict__ up,
                                                      const int* __restrict__ nn, size_t tri_stride, size_t hp_stride, int up_ld, int hp_ld,
                                                      int which, int rmax, float th, int* __restrict__ counts)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, p = blockIdx.y;
    if (r >= rmax) return;
    const RowSpan rs = cand_row_all(bp, hp, up, nn, tri_stride, hp_stride, up_ld, hp_ld, which, p, r);
    int c = 0;
    if (rs.ok)
        for (int j = rs.j0 + lane; j <= rs.j1; j += 64) c += ((float)rs.p[j] > th) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) counts[(size_t)p * rmax + r] = c;
}
__global__ __launch_bounds__(256) void cand_write_all(const double* __restrict__ bp, const double* __restrict__ hp, const double* __restrict__ up,
                                                      const int* __restrict__ nn, size_t tri_stride, size_t hp_stride, int up_ld, int hp_ld,
                                                      int which, int rmax, float th, const int* __restrict__ offsets,
                                                      rh_cand* __restrict__ out, int cap)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, p = blockIdx.y;
    if (r >= rmax) return;
    const RowSpan rs = cand_row_all(bp, hp, up, nn, tri_stride, hp_stride, up_ld, hp_ld, which, p, r);
    if (!rs.ok) return;
    int pos = offsets[(size_t)p * rmax + r];
    for (int jb = rs.j0; jb <= rs.j1; jb += 64) {
        const int j = jb + lane;
        const float pf = j <= rs.j1 ? (float)rs.p[j] : 0.0f;
        const bool hit = j <= rs.j1 && pf > th;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int k = pos + __popcll(m & ((1ull << lane) - 1ull));
            if (k < cap) {
                rh_cand e;
                e.i = which >= 3 ? j / hp_ld : rs.i;
                e.j = which >= 3 ? j % hp_ld : j;
                e.p = pf;
                out[k] = e;
            }
        }
        pos += __popcll(m);
    }
}

static thread_local std::string g_create_error;

struct GraphSlot {   // one captured launch sequence (see run_graphed)
    hipGraphExec_t exec = nullptr;
    size_t key = 0;
    int launches = 0, far = 0;
};

struct rh_ctx {
    int device = 0;
    int model = 0;
    std::string err;
    hipStream_t s_mc = nullptr, s_dx = nullptr;
    hipEvent_t ev[6] = {};  // mc: start, after inside, after outside ; dx: start, end ; all: end
    ScoreModel* d_model = nullptr;
    LinModel* d_lin = nullptr;
    LinModel h_lin;
    // other scale exponents of the linear McCaskill path, tried on the problems that leave the double range before the log-space
    // kernels are (retry_mc_lin_rungs): built on first use from the host copy of the score model
    static constexpr int kRungs = 3;
    ScoreModel* h_score = nullptr;
    LinModel* h_lin_r = nullptr;               // [kRungs]
    LinModel* d_lin_r[kRungs] = {nullptr, nullptr, nullptr};
    double* d_wT_r[kRungs] = {nullptr, nullptr, nullptr};
    int scale_ladder = 1;                      // RH_SCALE_LADDER=0: flagged problems go straight to the log-space kernels
    int scale_memory = 0;                      // rh_set_scale_memory / RH_SCALE_MEMORY=1: the next batch starts on the exponent most of the last one needed
                                               // (off by default: a sequence's bits then depend on its own letters only, never on the context's history)
    // the default exponent's model (what h_lin / d_lin / d_wT hold unless a pass runs on a rung) and the exponent the NEXT batch starts
    // with: -1 = default, k = rung k -- the one that held more than half of the last batch (a stream of structured RNAs does not pay
    // a failed first pass per batch)
    LinModel h_lin0;
    LinModel* d_lin0 = nullptr;
    double* d_wT0 = nullptr;
    int lin_primary = -1;
    int rescued_by[kRungs + 1] = {0, 0, 0, 0};  // sequences the last ladder moved to the default exponent [0] / rung k [k + 1]
    std::vector<int> rescaled_mc;              // sequences the last compute recomputed on the linear path with another exponent (rh_batch_fallbacks which = 2)
    ViennaDx* d_vienna = nullptr;  // RH_MODEL_VIENNA_BL only
    int vienna_sem = 0;            // kViennaSem18 / kViennaSem20 (0: CONTRAfold model)
    VLinModel* d_vlin = nullptr;   // the same model in scaled linear space
    VLinModel* h_vlin = nullptr;
    // Vienna-BL: other scale exponents of the linear path, tried on the WHOLE batch (single-molecule folds and two-molecule sweeps
    // together: the latter are seeded from the former) before the log-space kernels; see compute().  Model -1 = the default exponent.
    static constexpr int kVRungs = 3;
    ViennaDx* h_vienna = nullptr;              // host copy of the energy tables the rung models are built from
    VLinModel* h_vlin_m[kVRungs + 1] = {nullptr, nullptr, nullptr, nullptr};   // [0] = default, [k + 1] = rung k
    VLinModel* d_vlin_m[kVRungs + 1] = {nullptr, nullptr, nullptr, nullptr};
    int vlin_cur = -1, vlin_primary = -1;      // model selected now / the one a batch starts with
    bool defer_log = false, deferred = false;  // compute_once: a flagged problem ends the attempt instead of starting the log-space kernels
    // Vienna-BL, per-pair route of the ladder (round 3): when at most half of the pairs of a batch are flagged, only THOSE pairs are recomputed --
    // on a helper context of the same model (its own tables, its own whole-batch ladder and log-space fallback) -- and their results are
    // copied into this batch's result buffers; every other pair keeps the result of the first pass bit for bit
    rh_ctx* helper = nullptr;
    bool is_helper = false;
    int pair_helper = 1;           // RH_PAIR_HELPER=0: the whole batch is run again (round-2 behaviour); 2: helper whenever at most half of the pairs are flagged
    std::vector<int> flagged_pairs;            // pairs the deferred attempt flagged (folds, two-molecule sweeps or pf_duplex)
    std::string p_param, p_defaults;           // creation arguments, for the helper
    bool p_has_param = false, p_has_defaults = false;
    int p_use_bl = 1, p_sem = 0;
    bool went_log = false;                     // compute_once (Vienna-BL): the batch was recomputed by the log-space kernels
    std::vector<int> flagged_mc;               // sequences the deferred attempts flagged
    VLinModel* d_vdxl = nullptr;   // the same tables at the duplex scale (duplex_vlin.hip)
    VDxLin* d_vdx = nullptr;
    double vdx_s = 0.27;           // log Z of pf_duplex per unit of a+b: 0.23 (random ACGU) .. 0.32 (70 % GC)
    DxLinModel* d_dxlin = nullptr;
    DxLinModel h_dxlin;
    DxLinModel* d_dxlin_r[4] = {nullptr, nullptr, nullptr, nullptr};   // duplex scale-exponent ladder (retry_dx_lin_rungs), built on first use
    DxLinModel h_dxlin_r[4];
    std::vector<int> rescaled_dx;              // pairs the last compute recomputed on the linear duplex kernels with another exponent (rh_batch_fallbacks which = 3)
    DxLinBatch dxl = {};
    size_t dxl_layout = 0;         // (lda, rows) signature of the zero-padded table image currently in HBM
    int co_seed = 1;               // Vienna-BL, hp from the two-molecule ensemble: copy the one-strand cells from the single folds (RH_CO_SEED=0: sweep them again)
    int dx_strip = 1;              // linear duplex: eight anti-diagonals per launch (dxl_strip8); RH_DX_STRIP=0: four (dxl_sweep4)
    bool dx_quad = true;           // linear duplex: four anti-diagonals per launch (dxl_sweep4, 4 wavefronts per group); RH_DX_QUAD=0: two (dxl_sweep<W>)
    int dx_w = 4;                  // wavefronts per 64-cell group of the linear duplex kernel
    int last_dx_path = 0;
    bool far_mfma = true;          // block products on v_mfma_f64_16x16x4_f64 (BS = 16); RH_FAR_MFMA=0: LDS/FMA kernel
    int lookahead = 2;             // inside sweep: 2 = two diagonals per launch (lin_inside_diag MODE 3), 1 = look-ahead pairs of launches
                                   // (MODE 1/2), 0 = one full launch per diagonal; RH_LOOKAHEAD
    int strip = 3;                 // CONTRAfold linear path: KD = 8 diagonals per launch (mccaskill_strip.hip) with the banded near/far split;
                                   // RH_STRIP=0: the per-diagonal-pair kernels of mccaskill_lin.hip.  Bit 0 = inside sweep, bit 1 = outside sweep
    int far2 = -1;                 // two-level block products: -1 = by size (sequences of n >= 384), 0 / 1 forced (RH_FAR2)
    int far2_next = -1;            // launch-sequence state of far_outside_step
    int strip_w = 8;               // wavefronts per strip workgroup (RH_STRIP_W = 4 | 8)
    // short sequences (kSmallMin <= n <= kSmallMax, CONTRAfold model, scaled linear path): one workgroup per sequence, one launch
    // (mccaskill_small.hip); chosen per sequence by its length alone, so a result does not depend on the rest of the batch.  The sweeps
    // see these sequences with length 0 (d_n_sweep).  Opt-in (RH_SMALL=1): measured slower than the sweeps (6.1 against 4.9 ms per 1000 pairs of 109 + 53 letters).
    int small_on = 0;
    std::vector<int> small_list;
    void* d_small_list = nullptr; size_t cap_small_list = 0;
    void* d_n_sweep = nullptr; size_t cap_n_sweep = 0;
    int nmax_sweep = 0;
    // sequences shorter than 40 letters next to longer ones: the sweeps choose their launch organisation by the longest sequence they
    // see (strips of eight diagonals from 40 letters on), so these get a pass of their own with the organisation they would get alone
    // (d_n_short: their lengths, 0 for everyone else) -- a result then does not depend on what else is in the batch
    void* d_n_short = nullptr; size_t cap_n_short = 0;
    int n_short = 0, nmax_short = 0;
    int strip_filt = 1;            // single-branch filter of the strip kernels: 1 = factored (A(t) B(|l1-l2|) + sparse residual), 0 = dense (RH_STRIP_FILT)
    bool strip_filt_ok = false;    // the model's weights have the factored form (strip_weights verifies it entry by entry)
    int co_cut_min = 0, co_cut_max = 0;   // smallest / largest cut (length of s1) of the two-molecule batch: bounds of the groups its sweeps launch
    int co_window = 1;             // two-molecule sweeps launch only the groups around the cut (RH_CO_WINDOW=0: all groups, most of which return at once)
    int acc_final_t = 1;           // Vienna-BL accessibility: vlin_acc_final_t (one thread per letter, all widths; RH_ACC_FINAL_T=0: one thread per letter and width)
    int acc_wide = 1;              // Vienna-BL accessibility: vlin_acc_gaps_wide for the gap lengths 3..30 (RH_ACC_WIDE=0: vlin_acc_gaps for all)
    int strip_xcd = 1;             // groups of one sequence consecutive on one XCD (RH_STRIP_XCD=0: sequence-major launch order only)
    double* d_wT = nullptr;        // transposed, zero-padded single-branch weights wT[l1][t+1] of the strip kernels
    bool far_pk = true;            // ... on packed operand tiles (lin_pack_tiles + lin_far_*_pk); RH_FAR_PK=0: gather per product
    bool use_graphs = true;        // RH_NO_GRAPH=1 launches every kernel from the host instead
    GraphSlot g_in, g_out, g_dx;
    int mode = RH_MODE_AUTO;       // which McCaskill path rh_batch_compute takes
    int lin_w = 4;                 // wavefronts per 64-cell group of the linear outside kernel (Vienna-BL kernels: 8)
    int lin_w_in = 4;              // ... of the inside kernel (fewer, longer wavefronts: less per-wavefront scalar overhead)
    int lin_bs = 16;               // block size of the far/near split of the O(n^3) terms (0 = off)
    int last_path = 0;             // 1 = linear, 2 = log-space, 3 = linear then log-space fallback
    int max_w = 1;                 // accessibility widths 1..max_w (src/ractip.cpp:370-375); the CONTRAfold path has width 1 only

    // current batch (host mirror)
    int np = 0, ns = 0;
    bool has_mc = false, has_dx = false, computed = false;
    std::vector<int> n;  // [ns]
    McBatch mc = {};
    DxBatch dx = {};
    // owned device buffers + capacities (bytes)
    void* d_seq = nullptr;   size_t cap_seq = 0;
    void* d_n = nullptr;     size_t cap_n = 0;
    void* d_mctab = nullptr; size_t cap_mctab = 0;
    void* d_corowp = nullptr; size_t cap_corowp = 0;
    void* d_rowp = nullptr; size_t cap_rowp = 0;    // look-ahead partial sums of the next inside diagonal
    void* d_pk = nullptr; size_t cap_pk = 0;        // operand tiles of the block products (single-molecule batch)
    void* d_copk = nullptr; size_t cap_copk = 0;    // ... of the s1+s2 batch
    void* d_f5 = nullptr;    size_t cap_f5 = 0;
    void* d_bp = nullptr;    size_t cap_bp = 0;
    void* d_up = nullptr;    size_t cap_up = 0;
    void* d_dxtab = nullptr; size_t cap_dxtab = 0;
    void* d_hp = nullptr;    size_t cap_hp = 0;
    void* d_logz = nullptr;  size_t cap_logz = 0;
    void* d_scal = nullptr;  size_t cap_scal = 0;
    void* d_mclogz = nullptr; size_t cap_mclogz = 0;
    void* d_bad = nullptr;   size_t cap_bad = 0;
    void* d_cnt = nullptr;   size_t cap_cnt = 0;
    void* d_dxbad = nullptr; size_t cap_dxbad = 0;
    void* d_zbar = nullptr;  size_t cap_zbar = 0;
    void* d_zpart = nullptr; size_t cap_zpart = 0;   // per-chunk partial sums of Z~ (+ pairable-cell counts behind them)
    int lz_chunks = 0;
    void* d_cand = nullptr;  size_t cap_cand = 0;
    // compacted sub-batches of the per-problem log-space fallback
    void* d_subseq = nullptr; size_t cap_subseq = 0;
    void* d_subn = nullptr; size_t cap_subn = 0;
    void* d_subbp = nullptr; size_t cap_subbp = 0;
    void* d_subup = nullptr; size_t cap_subup = 0;
    void* d_subdseq = nullptr; size_t cap_subdseq = 0;
    void* d_subdn = nullptr; size_t cap_subdn = 0;
    void* d_subdx = nullptr; size_t cap_subdx = 0;
    bool tables_dirty = false;      // the last compute met values outside the double range: clear the tables before the next batch
    std::vector<uint8_t> h_codes;   // host mirror of d_seq
    std::vector<int> fallback_mc, fallback_dx;   // problems the last compute recomputed in log space (rh_batch_fallbacks)
    void* d_gaps = nullptr;  size_t cap_gaps = 0;
    void* d_allow = nullptr; size_t cap_allow = 0;   // structure-constraint masks [ns][ld*ld] bytes (Vienna-BL, optional)
    void* d_coallow = nullptr; size_t cap_coallow = 0;   // the same for the s1+s2 batch
    void* d_hplen = nullptr; size_t cap_hplen = 0;   // lam^d x hairpin length weight, d = 0..nmax (linear Vienna path)
    std::vector<double> h_hplen;
    // two-molecule (co_pf_fold) form of the hybridization matrix: one concatenated sequence s1+s2 per pair
    int hybrid = RH_HYBRID_DUPLEX;
    McBatch co = {};
    void* d_coseq = nullptr; size_t cap_coseq = 0;
    void* d_con = nullptr;   size_t cap_con = 0;     // [2][np]: lengths, cuts
    void* d_cotab = nullptr; size_t cap_cotab = 0;
    void* d_cof5 = nullptr;  size_t cap_cof5 = 0;    // f5i, f5o, xp, xs, xpo, xso
    void* d_cobp = nullptr;  size_t cap_cobp = 0;
    void* d_cobad = nullptr; size_t cap_cobad = 0;
    double ms[4] = {0, 0, 0, 0};
    int n_launch[3] = {0, 0, 0};
    int n_far[3] = {0, 0, 0};      // of which block-product launches (mccaskill_far.hip)
    bool overlap = true;           // false: duplex, inside and outside sweeps run one after the other (isolated phase timings)
    int time_cls = -1;             // rh_set_kernel_timing: sweep-kernel class whose launches are bracketed by event pairs (-1: none)
    std::vector<hipEvent_t> tev;   // event pool of the timed class (pairs), tev_n used by the last compute
    size_t tev_n = 0;
};

rh_ctx* make_ctx_for_helper(int device, int model, const char* param_file, const char* defaults_file, int use_bl, int semantics);   // = create_ctx (below)

namespace {

int fail(rh_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, e_ == hipErrorOutOfMemory ? RH_ERR_OOM : RH_ERR_HIP, "%s failed: %s", \
                        #call, hipGetErrorString(e_));                                           \
    } while (0)

// launch of one sweep kernel (class = 0 inside, 1 inside block products, 2 outside, 3 outside block products, 4 duplex)
// Measurement aid (rh_set_kernel_timing): launches of class `time_cls` are bracketed by a HIP event pair on their stream, so that
// bench.py can report the average duration of ONE kernel class live (what a kernel trace reports per kernel); off by default.
#define KLAUNCH(c, cls, kern, grid, block, stream, ...)                                                   \
    do {                                                                                                  \
        const bool timed_ = (c)->time_cls == (cls) && (c)->tev_n + 2 <= (c)->tev.size();                  \
        if (timed_) (void)hipEventRecord((c)->tev[(c)->tev_n++], stream);                                 \
        hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                                    \
        if (timed_) (void)hipEventRecord((c)->tev[(c)->tev_n++], stream);                                 \
    } while (0)

// grow-only device buffer
int ensure(rh_ctx* c, void** p, size_t* cap, size_t bytes, bool zero)
{
    if (bytes <= *cap && *p) return RH_OK;
    if (*p) { HIP_TRY(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    if (bytes > free_b) return fail(c, RH_ERR_OOM, "batch needs %zu MiB of HBM, %zu MiB free", bytes >> 20, free_b >> 20);
    HIP_TRY(c, hipMalloc(p, bytes));
    *cap = bytes;
    if (zero) HIP_TRY(c, hipMemset(*p, 0, bytes));
    return RH_OK;
}

uint8_t nuc_code(char ch)
{  // InferenceEngine.ipp:379-384: case-insensitive ACGU, anything else (incl. T, N) is code 4
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'U': case 'u': return 3;
        default: return 4;
    }
}

uint8_t vienna_code(char ch)
{  // ViennaRNA encode_char with energy_set 0: A,C,G,U -> 1..4 (T reads as U), anything else 0
    switch (ch) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'U': case 'u': case 'T': case 't': return 4;
        default: return 0;
    }
}

std::string default_param_path()
{
    Dl_info info;
    if (dladdr((void*)&default_param_path, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        p = (k == std::string::npos) ? std::string(".") : p.substr(0, k);
        return p + "/data/";
    }
    return "ractip_amd/data/";
}

inline size_t tri_size(int n) { return (size_t)(n + 1) * (n + 2) / 2; }
inline size_t tri_offset(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }

// Allowed-pair mask of pf_fold under fold_constrained (ViennaRNA 1.8 make_ptypes), M[a*ld + b], 1 <= a < b <= n:
//   'x' the letter never pairs; '<' it pairs only with a later letter, '>' only with an earlier one; a matched '(' ')'
//   is kept and every pair inconsistent with it (crossing it, or sharing a letter) is removed; '|' and '.' do not
//   restrict the partition function.  Returns false for unbalanced brackets or a forced pair of non-complementary letters.
bool build_allow_mask(const char* seq, int n, const char* cons, int ld, uint8_t* M, std::string* why)
{
    for (int a = 0; a < ld; a++)
        for (int b = 0; b < ld; b++) M[(size_t)a * ld + b] = (a >= 1 && a < b && b <= n) ? 1 : 0;
    const size_t clen = std::strlen(cons);
    std::vector<int> stack;
    for (int j = 1; j <= n; j++) {
        const char ch = (size_t)(j - 1) < clen ? cons[j - 1] : '.';
        if (ch == 'x') {
            for (int l = 1; l <= n; l++) { M[(size_t)l * ld + j] = 0; M[(size_t)j * ld + l] = 0; }
        } else if (ch == '(' || ch == '<') {
            if (ch == '(') stack.push_back(j);
            for (int l = 1; l < j; l++) M[(size_t)l * ld + j] = 0;
        } else if (ch == ')' || ch == '>') {
            if (ch == ')') {
                if (stack.empty()) { *why = "unbalanced ')' in the structure constraint"; return false; }
                const int i = stack.back();
                stack.pop_back();
                const uint8_t keep = M[(size_t)i * ld + j];
                for (int k = i; k <= j; k++) for (int l = j; l <= n; l++) M[(size_t)k * ld + l] = 0;
                for (int k = 1; k <= i; k++) for (int l = i; l <= j; l++) M[(size_t)k * ld + l] = 0;
                M[(size_t)i * ld + j] = keep;
                const uint8_t x = vienna_code(seq[i - 1]), y = vienna_code(seq[j - 1]);
                static const int T[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, 1, 0}, {0, 0, 2, 0, 3}, {0, 6, 0, 4, 0}};
                if (keep && !T[x][y]) { *why = "a forced pair of non-complementary letters (pair type 7) is not supported"; return false; }
            }
            for (int l = j + 1; l <= n; l++) M[(size_t)j * ld + l] = 0;
        }
    }
    if (!stack.empty()) { *why = "unbalanced '(' in the structure constraint"; return false; }
    return true;
}

// Stage `ns` sequences; pairs are (2p, 2p+1) when with_dx.  Allocates what is needed.  cons: per-sequence structure
// constraints (Vienna-BL, single-molecule batch only) or nullptr.
int stage(rh_ctx* c, int ns, const char* const* seqs, const int* lens, bool with_mc, bool with_dx, const char* const* cons = nullptr,
          const char* const* co_cons = nullptr)
{
    HIP_TRY(c, hipSetDevice(c->device));
    if (ns <= 0) return fail(c, RH_ERR_ARG, "empty batch");
    int nmax = 0, n1max = 0, n2max = 0;
    for (int k = 0; k < ns; k++) {
        if (lens[k] < 1) return fail(c, RH_ERR_ARG, "sequence %d has length %d (must be >= 1)", k, lens[k]);
        if (!seqs[k]) return fail(c, RH_ERR_ARG, "sequence %d is NULL", k);
        nmax = std::max(nmax, lens[k]);
        if (with_dx) { if (k & 1) n2max = std::max(n2max, lens[k]); else n1max = std::max(n1max, lens[k]); }
    }
    if (with_dx && (ns & 1)) return fail(c, RH_ERR_ARG, "duplex batch needs an even number of sequences");
    c->ns = ns; c->np = with_dx ? ns / 2 : 0;
    c->has_mc = with_mc; c->has_dx = with_dx; c->computed = false;
    c->n.assign(lens, lens + ns);

    const int lds = (nmax + 3 + 15) & ~15;  // codes 0..n+2 readable
    const bool vienna = c->model == RH_MODEL_VIENNA_BL;
    std::vector<uint8_t> codes((size_t)ns * lds, vienna ? 0 : 4);   // sentinel = the model's "no nucleotide" code
    for (int k = 0; k < ns; k++)
        for (int i = 0; i < lens[k]; i++) codes[(size_t)k * lds + 1 + i] = vienna ? vienna_code(seqs[k][i]) : nuc_code(seqs[k][i]);
    int rc;
    if ((rc = ensure(c, &c->d_seq, &c->cap_seq, codes.size(), false))) return rc;
    if ((rc = ensure(c, &c->d_n, &c->cap_n, sizeof(int) * ns, false))) return rc;
    c->h_codes = codes;
    HIP_TRY(c, hipMemcpyAsync(c->d_seq, codes.data(), codes.size(), hipMemcpyHostToDevice, c->s_mc));
    HIP_TRY(c, hipMemcpyAsync(c->d_n, lens, sizeof(int) * ns, hipMemcpyHostToDevice, c->s_mc));
    c->small_list.clear();
    c->nmax_sweep = nmax;
    c->n_short = 0; c->nmax_short = 0;
    if (with_mc && !vienna && !cons) {
        std::vector<int> nsw(lens, lens + ns), nsh(ns, 0);
        int nmax_rest = 0;
        for (int k = 0; k < ns; k++) {
            if (c->small_on && lens[k] >= kSmallMin && lens[k] <= kSmallMax) { c->small_list.push_back(k); nsw[k] = 0; }
            else nmax_rest = std::max(nmax_rest, lens[k]);
        }
        if (nmax_rest >= kStripMinN)   // some sequence runs in strips: the ones below that length get their own pass
            for (int k = 0; k < ns; k++)
                if (nsw[k] > 0 && nsw[k] < kStripMinN) { nsh[k] = nsw[k]; nsw[k] = 0; c->n_short++; c->nmax_short = std::max(c->nmax_short, nsh[k]); }
        c->nmax_sweep = 0;
        for (int k = 0; k < ns; k++) c->nmax_sweep = std::max(c->nmax_sweep, nsw[k]);
        if (!c->small_list.empty() || c->n_short) {
            // longest first: one workgroup occupies a CU, and workgroups of alternating cost land on alternating CUs
            std::stable_sort(c->small_list.begin(), c->small_list.end(), [&](int a, int b) { return lens[a] > lens[b]; });
            if ((rc = ensure(c, &c->d_small_list, &c->cap_small_list, sizeof(int) * ns, false))) return rc;
            if ((rc = ensure(c, &c->d_n_sweep, &c->cap_n_sweep, sizeof(int) * ns, false))) return rc;
            if ((rc = ensure(c, &c->d_n_short, &c->cap_n_short, sizeof(int) * ns, false))) return rc;
            if (!c->small_list.empty())
                HIP_TRY(c, hipMemcpyAsync(c->d_small_list, c->small_list.data(), sizeof(int) * c->small_list.size(), hipMemcpyHostToDevice, c->s_mc));
            HIP_TRY(c, hipMemcpyAsync(c->d_n_sweep, nsw.data(), sizeof(int) * ns, hipMemcpyHostToDevice, c->s_mc));
            HIP_TRY(c, hipMemcpyAsync(c->d_n_short, nsh.data(), sizeof(int) * ns, hipMemcpyHostToDevice, c->s_mc));
            HIP_TRY(c, hipStreamSynchronize(c->s_mc));   // (the staging vectors die with this scope)
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));  // host staging buffers die with this scope

    if (with_mc) {
        McBatch& B = c->mc;
        B.ns = ns; B.nmax = nmax; B.lds = lds;
        B.ld = (nmax + 2 + 1) & ~1;
        B.tab_stride = (size_t)B.ld * B.ld;
        B.seq_stride = B.tab_stride * (vienna ? (int)kViennaMcTables : (int)T_COUNT);
        B.tri_stride = (tri_size(nmax) + 1) & ~(size_t)1;
        if ((rc = ensure(c, &c->d_mctab, &c->cap_mctab, sizeof(double) * B.seq_stride * ns, false))) return rc;
        B.nb = (nmax - 1) / 16 + 1;
        B.pk_stride = (size_t)B.nb * (B.nb + 1) / 2 * 256;
        if ((rc = ensure(c, &c->d_pk, &c->cap_pk, sizeof(double) * B.pk_stride * kPkCopies * ns, false))) return rc;
        B.pk = (double*)c->d_pk;
        if ((rc = ensure(c, &c->d_rowp, &c->cap_rowp, sizeof(double) * 4 * B.ld * ns, false))) return rc;
        B.rowp = (double*)c->d_rowp;
        if ((rc = ensure(c, &c->d_f5, &c->cap_f5, sizeof(double) * 2 * B.ld * ns, false))) return rc;
        if ((rc = ensure(c, &c->d_up, &c->cap_up, sizeof(double) * B.ld * c->max_w * ns, false))) return rc;
        // gap probabilities [2 ns][32][ld] + the chunk sums of the gap lengths 1, 2 [8][2 ns][2][ld] (launch_mc_vlin)
        if (vienna && (rc = ensure(c, &c->d_gaps, &c->cap_gaps, sizeof(double) * (2 * 32 + 8 * 2 * 2) * B.ld * ns, false))) return rc;
        if (vienna) {
            const VLinModel& H = *c->h_vlin;
            c->h_hplen.resize((size_t)B.ld);
            for (int d = 0; d < B.ld; d++)   // hairpin of d unpaired letters: length weight (beyond 30 as part_func.c extrapolates) x lam^d
                c->h_hplen[d] = (d <= 30 ? H.E_hairpin[d] : std::exp(H.hairpin30 - H.lxc * std::log(d / 30.0))) * std::exp(-H.s * d);
            if ((rc = ensure(c, &c->d_hplen, &c->cap_hplen, sizeof(double) * B.ld, false))) return rc;
            HIP_TRY(c, hipMemcpyAsync(c->d_hplen, c->h_hplen.data(), sizeof(double) * B.ld, hipMemcpyHostToDevice, c->s_mc));
            HIP_TRY(c, hipStreamSynchronize(c->s_mc));
        }
        if ((rc = ensure(c, &c->d_mclogz, &c->cap_mclogz, sizeof(double) * ns, false))) return rc;
        if ((rc = ensure(c, &c->d_bad, &c->cap_bad, sizeof(int) * ns, false))) return rc;
        // bp entries outside 1<=i<j<=n are never written by the sweep: keep them zero
        const size_t bp_bytes = sizeof(double) * B.tri_stride * ns;
        if ((rc = ensure(c, &c->d_bp, &c->cap_bp, bp_bytes, false))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_bp, 0, bp_bytes, c->s_mc));
        B.allow = nullptr;
        if (cons) {
            std::vector<uint8_t> M((size_t)ns * B.ld * B.ld);
            std::string why;
            for (int k = 0; k < ns; k++)
                if (!build_allow_mask(seqs[k], lens[k], cons[k] ? cons[k] : "", B.ld, M.data() + (size_t)k * B.ld * B.ld, &why))
                    return fail(c, RH_ERR_ARG, "sequence %d: %s", k, why.c_str());
            if ((rc = ensure(c, &c->d_allow, &c->cap_allow, M.size(), false))) return rc;
            HIP_TRY(c, hipMemcpy(c->d_allow, M.data(), M.size(), hipMemcpyHostToDevice));
            B.allow = (const uint8_t*)c->d_allow;
        }
        if (c->tables_dirty) {
            // a problem of the previous batch overflowed: its tables hold Inf / NaN, which a later batch must never meet even in
            // cells it masks (0 x Inf).  One clear per such batch; ordinary batches reuse the tables as they are.
            HIP_TRY(c, hipMemsetAsync(c->d_mctab, 0, c->cap_mctab, c->s_mc));
            if (c->d_pk) HIP_TRY(c, hipMemsetAsync(c->d_pk, 0, c->cap_pk, c->s_mc));
            if (c->d_cotab) HIP_TRY(c, hipMemsetAsync(c->d_cotab, 0, c->cap_cotab, c->s_mc));
            if (c->d_copk) HIP_TRY(c, hipMemsetAsync(c->d_copk, 0, c->cap_copk, c->s_mc));
            c->tables_dirty = false;
        }
        B.seq = (const uint8_t*)c->d_seq; B.n = (const int*)c->d_n;
        B.tab = (double*)c->d_mctab;
        B.f5i = (double*)c->d_f5; B.f5o = (double*)c->d_f5 + (size_t)B.ld * ns;
        B.bp = (double*)c->d_bp; B.up = (double*)c->d_up;
    }
    if (with_dx) {
        DxBatch& D = c->dx;
        D.np = ns / 2; D.n1max = n1max; D.n2max = n2max; D.lds = lds;
        D.ldd = (n2max + 2 + 1) & ~1;
        D.tab_stride = (size_t)(n1max + 2) * D.ldd;
        D.pair_stride = D.tab_stride * 6;   // 4 tables (CONTRAfold model) or 6 (Vienna model: IN/OUT + two decorated copies each)
        // the linear path keeps anti-diagonal-major tables in the same buffer (sequential use)
        DxLinBatch& X = c->dxl;
        X.np = D.np; X.n1max = n1max; X.n2max = n2max; X.lds = lds; X.ldd = D.ldd;
        X.lda = (n1max + 2 + 2 * kDxPad + 1) & ~1;
        const size_t rows = (size_t)n1max + n2max + 3;
        X.tab_stride = rows * X.lda + 128;   // slack: the staged 96-column segments may run past the last row
        X.pair_stride = X.tab_stride * (vienna ? 6 : (int)DL_COUNT);   // Vienna-BL: raw + two decorated copies per direction
        const size_t dx_bytes = sizeof(double) * std::max(D.pair_stride, X.pair_stride) * D.np;
        void* before = c->d_dxtab;
        if ((rc = ensure(c, &c->d_dxtab, &c->cap_dxtab, dx_bytes, false))) return rc;
        const size_t layout = ((size_t)X.lda << 32) ^ rows ^ ((size_t)D.np << 48);
        if (c->d_dxtab != before || layout != c->dxl_layout || c->last_dx_path != 1) {
            // pad columns must be zero and a different layout (or the log-space path) leaves arbitrary bytes there
            HIP_TRY(c, hipMemsetAsync(c->d_dxtab, 0, dx_bytes, c->s_dx));
            c->dxl_layout = layout;
        }
        if ((rc = ensure(c, &c->d_dxbad, &c->cap_dxbad, sizeof(int) * D.np, false))) return rc;
        if ((rc = ensure(c, &c->d_zbar, &c->cap_zbar, sizeof(double) * D.np, false))) return rc;
        c->lz_chunks = (n1max + n2max - 1 + 15) / 16;   // kLzRows anti-diagonals per chunk
        if ((rc = ensure(c, &c->d_zpart, &c->cap_zpart, (sizeof(double) + sizeof(int)) * (size_t)D.np * c->lz_chunks, false))) return rc;
        if ((rc = ensure(c, &c->d_logz, &c->cap_logz, sizeof(double) * D.np, false))) return rc;
        const size_t hp_bytes = sizeof(double) * D.tab_stride * D.np;
        if ((rc = ensure(c, &c->d_hp, &c->cap_hp, hp_bytes, false))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_hp, 0, hp_bytes, c->s_dx));  // row 0 / column 0 stay zero
        D.seq = (const uint8_t*)c->d_seq; D.n = (const int*)c->d_n;
        D.tab = (double*)c->d_dxtab; D.hp = (double*)c->d_hp; D.logz = (double*)c->d_logz;
        if (vienna && c->hybrid == RH_HYBRID_COFOLD) {
            // concatenated sequences s1+s2, cut after s1
            McBatch& C = c->co;
            const int np = ns / 2, cmax = n1max + n2max;
            C = McBatch{};
            C.ns = np; C.nmax = cmax;
            C.lds = (cmax + 3 + 15) & ~15;
            C.ld = (cmax + 2 + 1) & ~1;
            C.tab_stride = (size_t)C.ld * C.ld;
            C.seq_stride = C.tab_stride * kViennaMcTables;
            C.tri_stride = (tri_size(cmax) + 1) & ~(size_t)1;
            std::vector<uint8_t> cc((size_t)np * C.lds, 0);
            std::vector<int> nn(2 * (size_t)np);
            for (int p = 0; p < np; p++) {
                const int a = lens[2 * p], b = lens[2 * p + 1];
                for (int i = 0; i < a; i++) cc[(size_t)p * C.lds + 1 + i] = vienna_code(seqs[2 * p][i]);
                for (int i = 0; i < b; i++) cc[(size_t)p * C.lds + 1 + a + i] = vienna_code(seqs[2 * p + 1][i]);
                nn[p] = a + b; nn[np + p] = a;
                c->co_cut_min = p == 0 ? a : std::min(c->co_cut_min, a);
                c->co_cut_max = p == 0 ? a : std::max(c->co_cut_max, a);
            }
            if ((rc = ensure(c, &c->d_coseq, &c->cap_coseq, cc.size(), false))) return rc;
            if ((rc = ensure(c, &c->d_con, &c->cap_con, sizeof(int) * nn.size(), false))) return rc;
            if ((rc = ensure(c, &c->d_cotab, &c->cap_cotab, sizeof(double) * C.seq_stride * np, false))) return rc;
            C.nb = (cmax - 1) / 16 + 1;
            C.pk_stride = (size_t)C.nb * (C.nb + 1) / 2 * 256;
            if ((rc = ensure(c, &c->d_copk, &c->cap_copk, sizeof(double) * C.pk_stride * kPkCopies * np, false))) return rc;
            C.pk = (double*)c->d_copk;
            if ((rc = ensure(c, &c->d_corowp, &c->cap_corowp, sizeof(double) * 4 * C.ld * np, false))) return rc;
            C.rowp = (double*)c->d_corowp;
            if ((rc = ensure(c, &c->d_cof5, &c->cap_cof5, sizeof(double) * 6 * C.ld * np, false))) return rc;
            if ((rc = ensure(c, &c->d_cobp, &c->cap_cobp, sizeof(double) * C.tri_stride * np, false))) return rc;
            if ((rc = ensure(c, &c->d_cobad, &c->cap_cobad, sizeof(int) * np, false))) return rc;
            if ((int)c->h_hplen.size() < C.ld) {   // hairpin length weights up to the joint length (see the single-molecule batch)
                const VLinModel& H = *c->h_vlin;
                c->h_hplen.resize((size_t)C.ld);
                for (int d = 0; d < C.ld; d++)
                    c->h_hplen[d] = (d <= 30 ? H.E_hairpin[d] : std::exp(H.hairpin30 - H.lxc * std::log(d / 30.0))) * std::exp(-H.s * d);
            }
            HIP_TRY(c, hipMemcpyAsync(c->d_coseq, cc.data(), cc.size(), hipMemcpyHostToDevice, c->s_dx));
            HIP_TRY(c, hipMemcpyAsync(c->d_con, nn.data(), sizeof(int) * nn.size(), hipMemcpyHostToDevice, c->s_dx));
            HIP_TRY(c, hipStreamSynchronize(c->s_dx));
            C.seq = (const uint8_t*)c->d_coseq; C.n = (const int*)c->d_con; C.cut = (const int*)c->d_con + np;
            C.tab = (double*)c->d_cotab;
            double* f = (double*)c->d_cof5;
            const size_t fs = (size_t)C.ld * np;
            C.f5i = f; C.f5o = f + fs; C.xp = f + 2 * fs; C.xs = f + 3 * fs; C.xpo = f + 4 * fs; C.xso = f + 5 * fs;
            C.bp = (double*)c->d_cobp; C.up = nullptr;
            if (co_cons) {   // constraints over the concatenation s1+s2 (one string of length n1+n2 per pair)
                std::vector<uint8_t> M((size_t)np * C.ld * C.ld);
                std::string why;
                for (int p = 0; p < np; p++) {
                    const std::string joint = std::string(seqs[2 * p], lens[2 * p]) + std::string(seqs[2 * p + 1], lens[2 * p + 1]);
                    if (!build_allow_mask(joint.c_str(), (int)joint.size(), co_cons[p] ? co_cons[p] : "", C.ld, M.data() + (size_t)p * C.ld * C.ld, &why))
                        return fail(c, RH_ERR_ARG, "pair %d: %s", p, why.c_str());
                }
                if ((rc = ensure(c, &c->d_coallow, &c->cap_coallow, M.size(), false))) return rc;
                HIP_TRY(c, hipMemcpy(c->d_coallow, M.data(), M.size(), hipMemcpyHostToDevice));
                C.allow = (const uint8_t*)c->d_coallow;
            }
        }
        X.seq = D.seq; X.n = D.n; X.tab = D.tab; X.hp = D.hp; X.hp_stride = D.tab_stride;
    }
    return RH_OK;
}

// ---- McCaskill sweeps, log-space path (always valid)
int launch_mc_log(rh_ctx* c, int pin, const McBatch& B, double* logz_out)
{
    hipLaunchKernelGGL(mc_init, dim3((B.ns + 63) / 64), dim3(64), 0, c->s_mc, B);
    for (int d = 0; d <= B.nmax - 1; d++) {
        const int waves = std::max(B.nmax - 1 - d, 0) + 1;
        KLAUNCH(c, 0, mc_inside_diag, pin ? dim3(B.ns, (waves + 3) / 4) : dim3((waves + 3) / 4, B.ns), dim3(256), c->s_mc, B, c->d_model, d, pin);
        c->n_launch[0]++;
    }
    HIP_TRY(c, hipEventRecord(c->ev[1], c->s_mc));
    for (int d = B.nmax - 2; d >= 0; d--) {
        const int waves = (B.nmax - 1 - d) + 1;
        KLAUNCH(c, 2, mc_outside_diag, pin ? dim3(B.ns, (waves + 3) / 4) : dim3((waves + 3) / 4, B.ns), dim3(256), c->s_mc, B, c->d_model, d, pin);
        c->n_launch[1]++;
    }
    hipLaunchKernelGGL(log_finish, dim3((B.ns + 63) / 64), dim3(64), 0, c->s_mc, B, logz_out);
    hipLaunchKernelGGL(mc_unpaired, dim3((B.nmax + 63) / 64, B.ns), dim3(256), 0, c->s_mc, B);
    return RH_OK;
}
int launch_mc_log(rh_ctx* c, int pin) { return launch_mc_log(c, pin, c->mc, (double*)c->d_mclogz); }

// ---- McCaskill sweeps + accessibility, Vienna-BL model (log space; mccaskill_vienna.hip)
int launch_mc_vienna(rh_ctx* c, int pin)
{
    const McBatch& B = c->mc;
    hipLaunchKernelGGL(mcv_init, dim3((B.ns + 63) / 64), dim3(64), 0, c->s_mc, B);
    for (int d = 0; d <= B.nmax - 1; d++) {
        const int waves = std::max(B.nmax - 1 - d, 0) + 1;
        KLAUNCH(c, 0, mcv_inside_diag, pin ? dim3(B.ns, (waves + 3) / 4) : dim3((waves + 3) / 4, B.ns), dim3(256), c->s_mc, B, c->d_vienna, d, pin);
        c->n_launch[0]++;
    }
    HIP_TRY(c, hipEventRecord(c->ev[1], c->s_mc));
    for (int d = B.nmax - 2; d >= 0; d--) {
        const int waves = (B.nmax - 1 - d) + 1;
        KLAUNCH(c, 2, mcv_outside_diag, pin ? dim3(B.ns, (waves + 3) / 4) : dim3((waves + 3) / 4, B.ns), dim3(256), c->s_mc, B, c->d_vienna, d, pin);
        c->n_launch[1]++;
    }
    hipLaunchKernelGGL(mcv_finish, dim3((B.ns + 63) / 64), dim3(64), 0, c->s_mc, B, (double*)c->d_mclogz);
    // accessibility P(i..i+w unpaired), w < max_w, from the finished tables
    const int tiles = (B.ld + 31) / 32;
    hipLaunchKernelGGL(mcv_acc_prep, dim3(tiles * tiles, B.ns, 3), dim3(256), 0, c->s_mc, B, c->d_vienna);
    hipLaunchKernelGGL(mcv_acc_hscan, dim3((B.nmax + 1 + 255) / 256, B.ns), dim3(256), 0, c->s_mc, B, 1 /* VM_FCX */);
    hipLaunchKernelGGL(mcv_acc_gaps, dim3((B.nmax * 30 + 3) / 4, B.ns, 2), dim3(256), 0, c->s_mc, B, c->d_vienna, (double*)c->d_gaps);
    hipLaunchKernelGGL(mcv_acc_final, dim3((B.nmax + 3) / 4, B.ns), dim3(256), 0, c->s_mc, B, c->d_vienna, (const double*)c->d_gaps, c->max_w);
    c->n_launch[1] += 4;
    return RH_OK;
}

// ---- block products (mccaskill_far.hip, BS = 16) on re-laid operand tiles: the tiles of block diagonal Dblk are packed once,
// right after their last cell is final, and then read by every product that uses them as two contiguous 2 KB fragments.
//   inside : far(D) uses FM1/FM tiles of block diagonals 2..D-2; block diagonal D-2 completes with fine diagonal (D-1)*16-1
//   outside: far(D) uses FM2o tiles of block diagonals >= D+2 (final before fine diagonal (D+1)*16-1) and FM1/FM tiles of
//            every block diagonal (the last two are packed when the outside phase starts)
// returns the number of launches it counts: 1 (the pack launch rides with its product; bench.py adds its traffic to the product's)
// two-level products (64x64 macro tiles under the 16x16 tile kernels, mccaskill_far.hip) pay from 6 macro blocks per axis on
// (measured: n = 200, 300 equal, n = 400 +2 %, n = 500 +4 %, n = 2000 +27 %)
// returns the length from which a SEQUENCE takes the two-level form (0: no sequence of this batch does)
static int far_two_level(const rh_ctx* c, const McBatch& B)
{
    const int from = c->far2 >= 0 ? (c->far2 ? 1 : 0) : 384;
    return from > 0 && B.nmax >= from ? from : 0;
}

static int far_inside_step(rh_ctx* c, const McBatch& B, hipStream_t st, int D, int last_block, int banded = 0)
{
    if (!c->far_pk) { KLAUNCH(c, 1, lin_far_inside_mfma, dim3(last_block - D + 1, B.ns), dim3(256), st, B, D); return 1; }
    const int l2 = far_two_level(c, B);
    KLAUNCH(c, 1, lin_pack_tiles<0>, dim3(B.nb - (D - 2), B.ns, 2), dim3(256), st, B, D - 2, 0, banded);
    if (l2 && (D + 3) % 4 == 0) {   // D = 4*D2-3: every operand tile of macro block diagonal D2 is packed now
        const int D2 = (D + 3) / 4, last2 = (B.nmax - 1) / 64;
        if (D2 >= 4 && D2 <= last2) KLAUNCH(c, 1, lin_far2_inside, dim3(last2 - D2 + 1, B.ns), dim3(256), st, B, D2, l2);
    }
    KLAUNCH(c, 1, lin_far_inside_pk, dim3(last_block - D + 1, B.ns), dim3(256), st, B, D, l2);
    return 1;
}
// repack2: the inside sweep left block diagonal 2 packed in the other form (masked for the banded split / plain for the block split)
static int far_outside_begin(rh_ctx* c, const McBatch& B, hipStream_t st, int last_block, int banded = 0, bool repack2 = false)
{
    if (!c->far_pk) return 0;
    c->far2_next = (B.nmax - 1) / 64;   // macro block diagonals whose 64-block products are still to be launched (descending)
    if (repack2 && last_block - 1 > 2) KLAUNCH(c, 3, lin_pack_tiles<1>, dim3(B.nb - 2, B.ns, 2), dim3(256), st, B, 2, 0, banded);
    for (int Dblk = std::max(2, last_block - 1); Dblk <= last_block; Dblk++)
        KLAUNCH(c, 3, lin_pack_tiles<1>, dim3(B.nb - Dblk, B.ns, 2), dim3(256), st, B, Dblk, 0, banded);
    return 0;
}
static int far_outside_step(rh_ctx* c, const McBatch& B, hipStream_t st, int D, int last_block)
{
    if (!c->far_pk) { KLAUNCH(c, 3, lin_far_outside_mfma, dim3(last_block - D + 1, B.ns, 2), dim3(256), st, B, D); return 1; }
    const int l2 = far_two_level(c, B);
    if (D + 2 <= last_block) KLAUNCH(c, 3, lin_pack_tiles<1>, dim3(B.nb - (D + 2), B.ns, 1), dim3(256), st, B, D + 2, 1, 0);
    if (l2) {   // macro block diagonal D2 holds tile block diagonals 4*D2-3 .. 4*D2+3: its products go first, their FM2o tiles (block diagonals >= 4*D2+5) are packed
        const int last2 = (B.nmax - 1) / 64;
        for (; c->far2_next >= 0 && 4 * c->far2_next + 3 >= D; c->far2_next--)
            KLAUNCH(c, 3, lin_far2_outside, dim3(last2 - c->far2_next + 1, B.ns, 2), dim3(256), st, B, c->far2_next, l2);
    }
    KLAUNCH(c, 3, lin_far_outside_pk, dim3(last_block - D + 1, B.ns, 2), dim3(256), st, B, D, l2);
    return 1;
}

// ---- Vienna-BL McCaskill sweeps, scaled linear-space path (mccaskill_vlin.hip) with the block products of mccaskill_far.hip.
// co = false: the single-molecule batch on the McCaskill stream (+ accessibility); co = true: the s1+s2 batch of the
// two-molecule hybridization matrix on the duplex stream (two more groups per launch for the exterior halves XS / XP)
template <int BS>
int launch_mc_vlin(rh_ctx* c, int pin, int phase, bool co)
{
    constexpr int W = 8;
    const McBatch& B = co ? c->co : c->mc;
    hipStream_t st = co ? c->s_dx : c->s_mc;
    int* bad = (int*)(co ? c->d_cobad : c->d_bad);
    int* nl = co ? &c->n_launch[2] : (phase == 0 ? &c->n_launch[0] : &c->n_launch[1]);
    int* nf = co ? &c->n_far[2] : (phase == 0 ? &c->n_far[0] : &c->n_far[1]);
    const int extra = co ? 3 : 1;   // F5 (+ XP, XS)
    const int last_block = BS > 0 ? (B.nmax - 1) / BS : 0;
    // two-molecule sweeps: the groups of diagonal dd with a cell on both strands are 1 + 64 slot <= cut < 1 + 64 slot + 64 + dd; the
    // union over the batch's cuts is launched behind the three F5 / XP / XS groups (window_slot_vl); cells = all cell groups of the launch
    const bool window = co && c->co_window && c->co_cut_min >= 1;
    const auto windowed = [&](int dd, int cells, int* pin_arg) -> int {
        const int t = c->co_cut_min - 65 - dd;
        const int lo = std::max(0, (t >= 0 ? t / 64 : -((-t + 63) / 64)) + 1), hi = std::min(cells - 1, (c->co_cut_max - 1) / 64);
        *pin_arg = pin | 128 | (lo << 8);
        return 3 + std::max(0, hi - lo + 1);
    };
    if (phase == 0) {
        hipLaunchKernelGGL(vlin_init, dim3((B.ns + 63) / 64), dim3(64), 0, st, B, bad);
        if (co && B.seeded) hipLaunchKernelGGL(vlin_co_seed, dim3(c->mc.nmax, B.ns), dim3(256), 0, st, B, c->mc);
        for (int d = 0; d <= B.nmax - 1; d++) {
            const int cells = (std::max(B.nmax - 1 - d, 0) + 63) / 64;
            const bool la1 = BS == 16 && c->lookahead && (d & 1) == 0;   // this launch also feeds diagonal d+1
            int pin_k = pin;
            const int groups = (window && B.seeded) ? windowed(la1 ? d + 1 : d, cells, &pin_k) : cells + extra;
            const double hp_d = c->h_hplen[d];
            const dim3 grid = pin ? dim3(B.ns, groups) : dim3(groups, B.ns);
            bool done = false;
            if constexpr (BS == 16) {
                if (c->lookahead) {   // look-ahead pairs: even diagonal = full launch that also accumulates d+1's sums, odd = one wavefront per group
                    done = true;
                    if ((d & 1) == 0) {
                        if (co) KLAUNCH(c, 0, (vlin_inside_diag<W, 16, true, 1>), grid, dim3(64 * W), st, B, c->d_vlin, d, hp_d, pin_k);
                        else KLAUNCH(c, 0, (vlin_inside_diag<W, 16, false, 1>), grid, dim3(64 * W), st, B, c->d_vlin, d, hp_d, pin);
                    } else {
                        if (co) KLAUNCH(c, 0, (vlin_inside_diag<W, 16, true, 2>), grid, dim3(64), st, B, c->d_vlin, d, hp_d, pin_k);
                        else KLAUNCH(c, 0, (vlin_inside_diag<W, 16, false, 2>), grid, dim3(64), st, B, c->d_vlin, d, hp_d, pin);
                    }
                }
            }
            if (!done) {
                if (co) KLAUNCH(c, 0, (vlin_inside_diag<W, BS, true, 0>), grid, dim3(64 * W), st, B, c->d_vlin, d, hp_d, pin_k);
                else KLAUNCH(c, 0, (vlin_inside_diag<W, BS, false, 0>), grid, dim3(64 * W), st, B, c->d_vlin, d, hp_d, pin);
            }
            (*nl)++;
            if (BS > 0 && (d + 1) % BS == 0) {
                const int D = (d + 1) / BS + 1;
                if (D >= 4 && D <= last_block) { (*nl) += far_inside_step(c, B, st, D, last_block); (*nf)++; }
            }
        }
        return RH_OK;
    }
    if (BS > 0) {
        (*nl) += far_outside_begin(c, B, st, last_block);
        for (int D = last_block; D >= 0 && (D + 1) * BS - 1 > B.nmax - 2; D--) { (*nl) += far_outside_step(c, B, st, D, last_block); (*nf)++; }
    }
    const bool la = BS == 16 && c->lookahead;   // look-ahead pairs (odd diagonal: full launch + the sums of the next, even: one wavefront per group)
    for (int d = la ? ((B.nmax - 2) | 1) : B.nmax - 2; d >= 0; d--) {
        if (BS > 0 && (d + 1) % BS == 0 && d <= B.nmax - 2) {
            const int D = (d + 1) / BS - 1;
            if (D >= 0 && D <= last_block) { (*nl) += far_outside_step(c, B, st, D, last_block); (*nf)++; }
        }
        bool done = false;
        if constexpr (BS == 16) {
            if (la) {
                done = true;
                int pin_k = pin;
                if (d & 1) {
                    const int cells = (B.nmax - d + 63) / 64;   // cells of diagonal d-1
                    const int groups = window ? windowed(d, cells, &pin_k) : cells + extra;
                    const dim3 grid = pin ? dim3(B.ns, groups) : dim3(groups, B.ns);
                    if (co) KLAUNCH(c, 2, (vlin_outside_diag<W, 16, true, 1>), grid, dim3(64 * W), st, B, c->d_vlin, d, pin_k, bad);
                    else KLAUNCH(c, 2, (vlin_outside_diag<W, 16, false, 1>), grid, dim3(64 * W), st, B, c->d_vlin, d, pin, bad);
                } else {
                    const int cells = (B.nmax - 1 - d + 63) / 64;
                    const int groups = window ? windowed(d, cells, &pin_k) : cells + extra;
                    const dim3 grid = pin ? dim3(B.ns, groups) : dim3(groups, B.ns);
                    if (co) KLAUNCH(c, 2, (vlin_outside_diag<W, 16, true, 2>), grid, dim3(64), st, B, c->d_vlin, d, pin_k, bad);
                    else KLAUNCH(c, 2, (vlin_outside_diag<W, 16, false, 2>), grid, dim3(64), st, B, c->d_vlin, d, pin, bad);
                }
            }
        }
        if (!done) {
            const int cells = (B.nmax - 1 - d + 63) / 64;
            int pin_k = pin;
            const int groups = window ? windowed(d, cells, &pin_k) : cells + extra;
            if (co) KLAUNCH(c, 2, (vlin_outside_diag<W, BS, true, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), st, B, c->d_vlin, d, pin_k, bad);
            else KLAUNCH(c, 2, (vlin_outside_diag<W, BS, false, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), st, B, c->d_vlin, d, pin, bad);
        }
        (*nl)++;
    }
    if (co) {
        const DxBatch& D = c->dx;
        hipLaunchKernelGGL(mcv_extract_hp, dim3((D.n1max * D.n2max + 255) / 256, B.ns), dim3(256), 0, st, B, D.hp, D.tab_stride, D.ldd, D.logz,
                           c->h_vlin->s, bad);
        return RH_OK;
    }
    hipLaunchKernelGGL(vlin_finish, dim3((B.ns + 63) / 64), dim3(64), 0, st, B, c->d_vlin, (double*)c->d_mclogz, bad);
    // accessibility P(i..i+w unpaired), w < max_w
    const int tiles = (B.ld + 31) / 32;
    hipLaunchKernelGGL(vlin_acc_prep, dim3(tiles * tiles, B.ns), dim3(256), 0, st, B, c->d_vlin, (const double*)c->d_hplen);
    hipLaunchKernelGGL(mcv_acc_hscan, dim3((B.nmax + 1 + 255) / 256, B.ns), dim3(256), 0, st, B, 10 /* VL_FM2F */);
    hipLaunchKernelGGL(vlin_acc_hsum, dim3((B.nmax + 3) / 4, B.ns), dim3(256), 0, st, B, c->max_w);
    if (c->acc_wide && (size_t)kViennaMcTables * B.tab_stride * sizeof(double) < ((size_t)1 << 32)) {   // (vlin_acc_gaps_wide addresses a sequence's tables with 32-bit offsets)
        // gap lengths 1, 2 (the tabulated shapes: six times the loads of a generic length): one thread per letter and length, the inner
        // spans in 8 chunks; 3..30: the lanes over the gap length (vlin_acc_gaps_wide)
        constexpr int NG = 2, NCH = 8;
        double* part = (double*)c->d_gaps + (size_t)2 * 32 * B.ld * B.ns;
        // (one wavefront per workgroup: the loop runs to the longest inner span of the workgroup's letters, which differ by the block width)
        hipLaunchKernelGGL(vlin_acc_gaps, dim3((B.nmax + 63) / 64, B.ns, 2 * NG * NCH), dim3(64), 0, st, B, c->d_vlin, (double*)c->d_gaps, NG, NCH, part);
        hipLaunchKernelGGL(vlin_acc_gsum, dim3((B.nmax + 255) / 256, B.ns, 2 * NG), dim3(256), 0, st, B, (double*)c->d_gaps, (const double*)part, NG, NCH);
        hipLaunchKernelGGL(vlin_acc_gaps_wide, dim3((B.nmax + 3) / 4, B.ns, 2), dim3(256), 0, st, B, c->d_vlin, (double*)c->d_gaps);
    } else
        hipLaunchKernelGGL(vlin_acc_gaps, dim3((B.nmax + 255) / 256, B.ns, 60), dim3(256), 0, st, B, c->d_vlin, (double*)c->d_gaps, 30, 1, (double*)nullptr);
    hipLaunchKernelGGL(vlin_acc_gsuf, dim3((B.nmax + 255) / 256, B.ns, 2), dim3(256), 0, st, B, (double*)c->d_gaps);
    if (c->acc_final_t && c->max_w <= 15)   // one thread per letter, all widths (the operands of the fifteen widths overlap)
        hipLaunchKernelGGL(vlin_acc_final_t, dim3((B.nmax + 255) / 256, B.ns), dim3(256), 0, st, B, c->d_vlin, (const double*)c->d_gaps, c->max_w);
    else
        hipLaunchKernelGGL(vlin_acc_final, dim3((B.nmax + 255) / 256, B.ns, c->max_w), dim3(256), 0, st, B, c->d_vlin, (const double*)c->d_gaps, c->max_w);
    c->n_launch[1] += 6;
    return RH_OK;
}

// ---- hybridization matrix from the two-molecule ensemble (co_pf_fold semantics): the same sweeps over s1+s2 with a cut
int launch_cofold(rh_ctx* c)
{
    const McBatch& B = c->co;
    const DxBatch& D = c->dx;
    const int pin = B.ns % 8 == 0 ? 1 : 0;
    hipLaunchKernelGGL(mcv_init, dim3((B.ns + 63) / 64), dim3(64), 0, c->s_dx, B);
    for (int d = 0; d <= B.nmax - 1; d++) {
        const int waves = std::max(B.nmax - 1 - d, 0) + 3;   // cells, F5i, XP, XS
        KLAUNCH(c, 4, mcv_inside_diag, pin ? dim3(B.ns, (waves + 3) / 4) : dim3((waves + 3) / 4, B.ns), dim3(256), c->s_dx, B, c->d_vienna, d, pin);
        c->n_launch[2]++;
    }
    for (int d = B.nmax - 2; d >= 0; d--) {
        const int waves = (B.nmax - 1 - d) + 3;
        KLAUNCH(c, 4, mcv_outside_diag, pin ? dim3(B.ns, (waves + 3) / 4) : dim3((waves + 3) / 4, B.ns), dim3(256), c->s_dx, B, c->d_vienna, d, pin);
        c->n_launch[2]++;
    }
    hipLaunchKernelGGL(mcv_extract_hp, dim3((D.n1max * D.n2max + 255) / 256, B.ns), dim3(256), 0, c->s_dx, B, D.hp, D.tab_stride, D.ldd, D.logz,
                       -1.0, (int*)nullptr);
    return RH_OK;
}

// ---- McCaskill sweeps, scaled linear-space path (fast; flags sequences that left the double range)
// BS > 0: block products (mccaskill_far.hip) take the k-terms of complete blocks; schedule:
//   inside : far(D) right after fine diagonal (D-1)*BS-1  (its operands are final, tile (I,I+D) starts at (D-1)*BS+1)
//   outside: far(D) right before fine diagonal (D+1)*BS-1 (operands: spans >= (D+1)*BS+1, already final)
// the strip kernels need the packed block products (masked tiles) and at least one strip behind the 32 bootstrap diagonals
static bool strip_inside(const rh_ctx* c, const McBatch& B) { return (c->strip & 1) && c->far_pk && c->far_mfma && c->lin_bs == 16 && B.nmax >= kStripMinN; }
static bool strip_outside(const rh_ctx* c, const McBatch& B) { return (c->strip & 2) && c->far_pk && c->far_mfma && c->lin_bs == 16 && B.nmax >= kStripMinN; }

// the sweeps of one phase over the sequences B shows (lengths 0 hide a sequence); BR: the batch as uploaded (lin_init / lin_finish /
// mc_unpaired see every sequence)
template <int W, int BS>
int launch_mc_lin_body(rh_ctx* c, int pin, int phase, const McBatch& B, const McBatch& BR, bool init, bool finish)
{
    int* bad = (int*)c->d_bad;
    const int last_block = BS > 0 ? (B.nmax - 1) / BS : 0;
    if (phase == 0) {
    if (init) hipLaunchKernelGGL(lin_init, dim3((BR.ns + 63) / 64), dim3(64), 0, c->s_mc, BR, c->d_lin, bad);
    if constexpr (W == 4 && BS == 16) {
        if (strip_inside(c, B)) {
            // diagonals 0..31 by pairs (every row is "near" there), then strips of kStripKD diagonals (mccaskill_strip.hip)
            constexpr int KD = 8, GS = 64 - (KD - 1);
            for (int d = 0; d < 32; d += 2) {
                const int groups = (std::max(B.nmax - 1 - d, 0) + 62) / 63 + 1;
                KLAUNCH(c, 0, (lin_inside_diag<4, 16, 3>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B, c->d_lin, d,
                        std::exp(-c->h_lin.s * d), pin);
                c->n_launch[0]++;
            }
            int d0 = 32;
            for (; d0 <= B.nmax - 2; d0 += KD) {
                const int groups = (std::max(B.nmax - 1 - d0, 0) + GS - 1) / GS + 1;
                if (c->strip_w == 4)
                    KLAUNCH(c, 0, (lin_inside_strip<KD, 4, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(256), c->s_mc, B, c->d_lin, c->d_wT, d0,
                            d0 == 32 ? 32 : d0 - KD + 2, std::exp(-c->h_lin.s * d0), (pin && c->strip_xcd) ? 2 : pin);
                else if (c->strip_filt && c->strip_filt_ok)
                    KLAUNCH(c, 0, (lin_inside_strip<KD, 8, 1>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(512), c->s_mc, B, c->d_lin, c->d_wT, d0,
                            d0 == 32 ? 32 : d0 - KD + 2, std::exp(-c->h_lin.s * d0), (pin && c->strip_xcd) ? 2 : pin);
                else
                    KLAUNCH(c, 0, (lin_inside_strip<KD, 8, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(512), c->s_mc, B, c->d_lin, c->d_wT, d0,
                            d0 == 32 ? 32 : d0 - KD + 2, std::exp(-c->h_lin.s * d0), (pin && c->strip_xcd) ? 2 : pin);
                c->n_launch[0]++;
                if ((d0 + KD) % BS == 0) {
                    const int D = (d0 + KD) / BS + 1;
                    if (D >= 4 && D <= last_block) { c->n_launch[0] += far_inside_step(c, B, c->s_mc, D, last_block, 1); c->n_far[0]++; }
                }
            }
            hipLaunchKernelGGL(lin_f5i_tail, dim3(B.ns), dim3(256), 0, c->s_mc, B, c->d_lin, d0 - KD + 2);
            return RH_OK;
        }
        if (c->lookahead == 2) {   // two diagonals per launch (lin_inside_diag MODE 3); the last launch may hold only F5i[nmax]
            for (int d = 0; d <= B.nmax; d += 2) {
                const int groups = (std::max(B.nmax - 1 - d, 0) + 62) / 63 + 1;
                KLAUNCH(c, 0, (lin_inside_diag<4, 16, 3>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B, c->d_lin, d,
                        std::exp(-c->h_lin.s * d), pin);
                c->n_launch[0]++;
                if ((d + 2) % BS == 0) {
                    const int D = (d + 2) / BS + 1;
                    if (D >= 4 && D <= last_block) { c->n_launch[0] += far_inside_step(c, B, c->s_mc, D, last_block); c->n_far[0]++; }
                }
            }
            return RH_OK;
        }
    }
    for (int d = 0; d <= B.nmax - 1; d++) {
        const int groups = (std::max(B.nmax - 1 - d, 0) + 63) / 64 + 1;
        if constexpr (W == 4 && BS == 16) {
            if (c->lookahead && (d & 1) == 0)      // even diagonal: also accumulates the look-ahead sums of d+1 ...
                KLAUNCH(c, 0, (lin_inside_diag<4, 16, 1>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B, c->d_lin, d,
                        std::exp(-c->h_lin.s * d), pin);
            else if (c->lookahead)                 // ... which then needs one wavefront per group
                KLAUNCH(c, 0, (lin_inside_diag<4, 16, 2>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64), c->s_mc, B, c->d_lin, d,
                        std::exp(-c->h_lin.s * d), pin);
            else
                KLAUNCH(c, 0, (lin_inside_diag<W, BS, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B, c->d_lin, d,
                        std::exp(-c->h_lin.s * d), pin);
        } else {
            KLAUNCH(c, 0, (lin_inside_diag<W, BS, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B, c->d_lin, d,
                    std::exp(-c->h_lin.s * d), pin);
        }
        c->n_launch[0]++;
        if (BS > 0 && (d + 1) % BS == 0) {
            const int D = (d + 1) / BS + 1;
            if (D >= 4 && D <= last_block) {
                if (BS == 16 && c->far_mfma) c->n_launch[0] += far_inside_step(c, B, c->s_mc, D, last_block);
                else {
                    KLAUNCH(c, 1, lin_far_inside<(BS > 0 ? BS : 16)>, dim3(last_block - D + 1, B.ns), dim3(256), c->s_mc, B, D);
                    c->n_launch[0]++;
                }
                c->n_far[0]++;
            }
        }
    }
    return RH_OK;
    }
    const bool in_banded = (W == 4 && BS == 16) && strip_inside(c, B);
    if constexpr (W == 4 && BS == 16) {
        if (strip_outside(c, B)) {
            // strips of KD diagonals from the top (mccaskill_strip.hip), banded near/far split: block diagonal 2 of FM1 / FM is packed masked
            constexpr int KD = 8, GS = 64 - (KD - 1);
            c->n_launch[1] += far_outside_begin(c, B, c->s_mc, last_block, 1, !in_banded);
            const int d0_top = (B.nmax - 2) | (KD - 1);
            for (int D = last_block; D >= 0 && (D + 1) * BS - 1 > d0_top; D--) { c->n_launch[1] += far_outside_step(c, B, c->s_mc, D, last_block); c->n_far[1]++; }
            hipLaunchKernelGGL(lin_f5o_head, dim3(B.ns), dim3(256), 0, c->s_mc, B, c->d_lin, B.nmax - 1, d0_top - 5);
            for (int d0 = d0_top; d0 >= KD - 1; d0 -= KD) {
                if ((d0 + 1) % BS == 0) {
                    const int D = (d0 + 1) / BS - 1;
                    if (D >= 0 && D <= last_block) { c->n_launch[1] += far_outside_step(c, B, c->s_mc, D, last_block); c->n_far[1]++; }
                }
                const int groups = (std::max(B.nmax - 1 - (d0 - (KD - 1)), 0) + GS - 1) / GS + 1;
                if (c->strip_w == 4)
                    KLAUNCH(c, 2, (lin_outside_strip<KD, 4, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(256), c->s_mc, B, c->d_lin, c->d_wT, d0,
                            d0 - 6, d0 - 13, (pin && c->strip_xcd) ? 2 : pin, bad);
                else if (c->strip_filt && c->strip_filt_ok)
                    KLAUNCH(c, 2, (lin_outside_strip<KD, 8, 1>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(512), c->s_mc, B, c->d_lin, c->d_wT, d0,
                            d0 - 6, d0 - 13, (pin && c->strip_xcd) ? 2 : pin, bad);
                else
                    KLAUNCH(c, 2, (lin_outside_strip<KD, 8, 0>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(512), c->s_mc, B, c->d_lin, c->d_wT, d0,
                            d0 - 6, d0 - 13, (pin && c->strip_xcd) ? 2 : pin, bad);
                c->n_launch[1]++;
            }
            if (finish) hipLaunchKernelGGL(lin_finish, dim3((BR.ns + 63) / 64), dim3(64), 0, c->s_mc, BR, c->d_lin, (double*)c->d_mclogz, bad);
            if (finish) hipLaunchKernelGGL(mc_unpaired, dim3((BR.nmax + 63) / 64, BR.ns), dim3(256), 0, c->s_mc, BR);
            return RH_OK;
        }
    }
    if (BS == 16 && c->far_mfma) c->n_launch[1] += far_outside_begin(c, B, c->s_mc, last_block, 0, in_banded);
    if (BS > 0)  // tiles whose first cell would come before the first outside diagonal: their far sums are empty
        for (int D = last_block; D >= 0 && (D + 1) * BS - 1 > B.nmax - 2; D--) {
            if (BS == 16 && c->far_mfma) c->n_launch[1] += far_outside_step(c, B, c->s_mc, D, last_block);
            else {
                KLAUNCH(c, 3, lin_far_outside<(BS > 0 ? BS : 16)>, dim3(last_block - D + 1, B.ns, 2), dim3(256), c->s_mc, B, D);
                c->n_launch[1]++;
            }
            c->n_far[1]++;
        }
    if constexpr (BS == 16 && (W == 8 || W == 4)) {
        if (c->lookahead == 2 && c->far_mfma) {   // two diagonals per launch (lin_outside_pair)
            // pairs are (odd, even) whatever the batch: a sequence's results do not depend on its neighbours' lengths
            for (int d = (B.nmax - 2) | 1; d >= 0; d -= 2) {
                for (int r = d; r >= d - 1 && r >= 0; r--)   // block products whose tiles start on either diagonal of the pair
                    if ((r + 1) % BS == 0) {
                        const int D = (r + 1) / BS - 1;
                        if (D >= 0 && D <= last_block) { c->n_launch[1] += far_outside_step(c, B, c->s_mc, D, last_block); c->n_far[1]++; }
                    }
                const int ncol = B.nmax - 1 - d + 1;          // columns of the longer diagonal d-1 (d = 0: diagonal 0 alone, one less)
                const int groups = std::max(1, (ncol - 1 + 62) / 63) + 1;
                KLAUNCH(c, 2, (lin_outside_pair<W, BS>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B, c->d_lin, d,
                        d, pin, bad);
                c->n_launch[1]++;
            }
            if (finish) hipLaunchKernelGGL(lin_finish, dim3((BR.ns + 63) / 64), dim3(64), 0, c->s_mc, BR, c->d_lin, (double*)c->d_mclogz, bad);
            if (finish) hipLaunchKernelGGL(mc_unpaired, dim3((BR.nmax + 63) / 64, BR.ns), dim3(256), 0, c->s_mc, BR);
            return RH_OK;
        }
    }
    for (int d = B.nmax - 2; d >= 0; d--) {
        if (BS > 0 && (d + 1) % BS == 0) {
            const int D = (d + 1) / BS - 1;
            if (D >= 0 && D <= last_block) {
                if (BS == 16 && c->far_mfma) c->n_launch[1] += far_outside_step(c, B, c->s_mc, D, last_block);
                else {
                    KLAUNCH(c, 3, lin_far_outside<(BS > 0 ? BS : 16)>, dim3(last_block - D + 1, B.ns, 2), dim3(256), c->s_mc, B, D);
                    c->n_launch[1]++;
                }
                c->n_far[1]++;
            }
        }
        const int groups = (B.nmax - 1 - d + 63) / 64 + 1;
        KLAUNCH(c, 2, (lin_outside_diag<W, BS>), pin ? dim3(B.ns, groups) : dim3(groups, B.ns), dim3(64 * W), c->s_mc, B,
                           c->d_lin, d, pin, bad);
        c->n_launch[1]++;
    }
    if (finish) hipLaunchKernelGGL(lin_finish, dim3((BR.ns + 63) / 64), dim3(64), 0, c->s_mc, BR, c->d_lin, (double*)c->d_mclogz, bad);
    if (finish) hipLaunchKernelGGL(mc_unpaired, dim3((BR.nmax + 63) / 64, BR.ns), dim3(256), 0, c->s_mc, BR);
    return RH_OK;
}


// Which sequences run where is decided per sequence, by its length alone: 8..109 letters by their own workgroup when RH_SMALL=1
// (mccaskill_small.hip), fewer than kStripMinN letters next to longer ones in a pass of their own (the organisation they would get
// alone), everyone else in the sweeps.  Sub-batches of the scale ladder (c->mc.n is not the upload's length array) run as they are.
template <int W, int BS>
int launch_mc_lin(rh_ctx* c, int pin, int phase)
{
    const McBatch BR = c->mc;
    const bool routed = (const void*)BR.n == c->d_n && (!c->small_list.empty() || c->n_short > 0);
    if (!routed) return launch_mc_lin_body<W, BS>(c, pin, phase, BR, BR, true, true);
    int* bad = (int*)c->d_bad;
    McBatch BL = BR, BSH = BR;
    BL.n = (const int*)c->d_n_sweep; BL.nmax = c->nmax_sweep;
    BSH.n = (const int*)c->d_n_short; BSH.nmax = c->nmax_short;
    int rc;
    if (phase == 0) {
        hipLaunchKernelGGL(lin_init, dim3((BR.ns + 63) / 64), dim3(64), 0, c->s_mc, BR, c->d_lin, bad);
        if (!c->small_list.empty()) {
            launch_lin_small(BR, c->d_lin, c->d_wT + kStripFiltOff + kStripFiltLen, (const int*)c->d_small_list, (int)c->small_list.size(), bad, c->s_mc);
            c->n_launch[0]++;
        }
        if (BL.nmax > 0 && (rc = launch_mc_lin_body<W, BS>(c, pin, 0, BL, BR, false, false))) return rc;
        if (c->n_short > 0 && (rc = launch_mc_lin_body<W, BS>(c, pin, 0, BSH, BR, false, false))) return rc;
        return RH_OK;
    }
    if (BL.nmax > 0 && (rc = launch_mc_lin_body<W, BS>(c, pin, 1, BL, BR, false, false))) return rc;
    if (c->n_short > 0 && (rc = launch_mc_lin_body<W, BS>(c, pin, 1, BSH, BR, false, false))) return rc;
    hipLaunchKernelGGL(lin_finish, dim3((BR.ns + 63) / 64), dim3(64), 0, c->s_mc, BR, c->d_lin, (double*)c->d_mclogz, bad);
    hipLaunchKernelGGL(mc_unpaired, dim3((BR.nmax + 63) / 64, BR.ns), dim3(256), 0, c->s_mc, BR);
    return RH_OK;
}

template <int BS>
int launch_mc_lin_w(rh_ctx* c, int pin, int phase)
{
    switch (phase == 0 ? c->lin_w_in : c->lin_w) {
        case 16: return launch_mc_lin<16, BS>(c, pin, phase);
        case 4: if (BS == 16) return launch_mc_lin<4, 16>(c, pin, phase); else return launch_mc_lin<8, BS>(c, pin, phase);
        default: return launch_mc_lin<8, BS>(c, pin, phase);
    }
}
int launch_mc_lin_any(rh_ctx* c, int pin, int phase)
{
    switch (c->lin_bs) {
        case 0: return launch_mc_lin_w<0>(c, pin, phase);
        case 32: return launch_mc_lin_w<32>(c, pin, phase);
        default: return launch_mc_lin_w<16>(c, pin, phase);
    }
}

// ---- hipGraph replay of the fast path.  The launch sequence of a batch depends only on its shape (and on the
// buffer addresses baked into the kernel arguments), so it is captured once per shape and replayed: the ~1000
// launches per sweep then cost the GPU-side ~1.5 us boundary instead of a host launch each.
size_t shape_key(const rh_ctx* c, int which);

template <class F>
int run_graphed(rh_ctx* c, GraphSlot& g, size_t key, hipStream_t stream, int* launch_counter, int* far_counter, F&& launch)
{
    if (!c->use_graphs || c->time_cls >= 0) return launch();   // (timed launches are host launches: events between graph nodes would be captured)
    if (!g.exec || g.key != key) {
        if (g.exec) { HIP_TRY(c, hipGraphExecDestroy(g.exec)); g.exec = nullptr; }
        hipGraph_t graph = nullptr;
        const int before = *launch_counter, far_before = *far_counter;
        HIP_TRY(c, hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        const int rc = launch();
        hipError_t e = hipStreamEndCapture(stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }   // (the captured graph is not leaked on the error paths)
        if (e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); return fail(c, RH_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e)); }
        e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { g.exec = nullptr; return fail(c, RH_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
        g.key = key;
        g.launches = *launch_counter - before;
        g.far = *far_counter - far_before;
        *launch_counter = before;
        *far_counter = far_before;
    }
    HIP_TRY(c, hipGraphLaunch(g.exec, stream));
    *launch_counter += g.launches;
    *far_counter += g.far;
    return RH_OK;
}

// ---- duplex sweeps, log-space path
int launch_dx_log(rh_ctx* c, const DxBatch& D)
{
    const int smax = D.n1max + D.n2max;
    const int steps = smax / 2;
    const int waves = 2 * std::min(D.n1max, D.n2max);
    for (int t = 0; t < steps; t++) {
        KLAUNCH(c, 4, dx_sweep_diag, dim3((waves + 3) / 4, D.np, 2), dim3(256), c->s_dx, D, c->d_model, t);
        c->n_launch[2]++;
    }
    hipLaunchKernelGGL(dx_logz, dim3(D.np), dim3(1024), 0, c->s_dx, D, c->d_model);
    const int cells = D.n1max * D.n2max;
    hipLaunchKernelGGL(dx_posterior, dim3((cells + 255) / 256, D.np), dim3(256), 0, c->s_dx, D);
    return RH_OK;
}
int launch_dx_log(rh_ctx* c) { return launch_dx_log(c, c->dx); }

// ---- per-problem fallback (CONTRAfold model): only the sequences / pairs whose scaled values left the double range are
// recomputed by the log-space kernels, as a compacted sub-batch; everyone else keeps the linear-path result, bit for bit.
// (The reference has no such cliff at all: its log-space arithmetic, LogSpace.hpp:232-244, is always valid.)
int recompute_mc_subset_log(rh_ctx* c, const std::vector<int>& F)
{
    const McBatch& B = c->mc;
    const int nsub = (int)F.size();
    std::vector<uint8_t> codes((size_t)nsub * B.lds);
    std::vector<int> lens(nsub);
    int nmax = 0;
    for (int k = 0; k < nsub; k++) {
        std::memcpy(codes.data() + (size_t)k * B.lds, c->h_codes.data() + (size_t)F[k] * B.lds, B.lds);
        lens[k] = c->n[F[k]];
        nmax = std::max(nmax, lens[k]);
    }
    int rc;
    const size_t up_per = (size_t)B.ld * c->max_w;
    if ((rc = ensure(c, &c->d_subseq, &c->cap_subseq, codes.size(), false))) return rc;
    if ((rc = ensure(c, &c->d_subn, &c->cap_subn, sizeof(int) * nsub, false))) return rc;
    if ((rc = ensure(c, &c->d_subbp, &c->cap_subbp, sizeof(double) * B.tri_stride * nsub, false))) return rc;
    if ((rc = ensure(c, &c->d_subup, &c->cap_subup, sizeof(double) * (up_per + 1) * nsub, false))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_subseq, codes.data(), codes.size(), hipMemcpyHostToDevice, c->s_mc));
    HIP_TRY(c, hipMemcpyAsync(c->d_subn, lens.data(), sizeof(int) * nsub, hipMemcpyHostToDevice, c->s_mc));
    HIP_TRY(c, hipMemsetAsync(c->d_subbp, 0, sizeof(double) * B.tri_stride * nsub, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));   // the host staging vectors die with this scope
    McBatch S = B;                 // same ld / strides: the tables of the linear pass are dead and are reused
    S.ns = nsub; S.nmax = nmax;
    S.seq = (const uint8_t*)c->d_subseq; S.n = (const int*)c->d_subn;
    S.f5i = (double*)c->d_f5; S.f5o = (double*)c->d_f5 + (size_t)B.ld * nsub;
    S.bp = (double*)c->d_subbp; S.up = (double*)c->d_subup;
    double* sub_logz = (double*)c->d_subup + up_per * nsub;
    if ((rc = launch_mc_log(c, nsub % 8 == 0 ? 1 : 0, S, sub_logz))) return rc;
    for (int k = 0; k < nsub; k++) {   // scatter to the flagged sequences' slots
        HIP_TRY(c, hipMemcpyAsync((double*)c->d_bp + (size_t)F[k] * B.tri_stride, (double*)c->d_subbp + (size_t)k * B.tri_stride,
                                  sizeof(double) * B.tri_stride, hipMemcpyDeviceToDevice, c->s_mc));
        HIP_TRY(c, hipMemcpyAsync((double*)c->d_up + (size_t)F[k] * up_per, (double*)c->d_subup + (size_t)k * up_per, sizeof(double) * up_per,
                                  hipMemcpyDeviceToDevice, c->s_mc));
        HIP_TRY(c, hipMemcpyAsync((double*)c->d_mclogz + F[k], sub_logz + k, sizeof(double), hipMemcpyDeviceToDevice, c->s_mc));
    }
    return RH_OK;
}

// single-branch weights of the strip kernels: wT[l1*40 + t+1] = shape_w(l1, t-l1), zero where the shape does not exist (the dense
// filter, FILT = 0), followed by the FACTORED form of the same weights at offset kStripFiltOff (FILT = 1, mccaskill_strip.hip):
// cache_score_single[l1][l2] (InferenceEngine.ipp:1161-1197) of an interior loop is length term(l1+l2) + asymmetry term(|l1-l2|)
// plus corrections on a sparse set (bulges l1 = 0 | l2 = 0, the symmetric term on l1 == l2, the explicit terms for l1, l2 <= 4), so
//   w(l1, t-l1) = A(t) * B(|2 l1 - t|) + R(l1, t),   R != 0 only for bulge ends, the centre tap and a few (l1, l2 <= 4) shapes,
// and the B-weighted row sums obey S_{t+2}[i-1] = S_t[i] + B(t) (x[i+1] + x[i+t+1]): two diagonals later the same table row needs
// two more taps instead of a whole pass.  A, B (any gauge) and R are taken from the weights themselves and the reconstruction is
// verified entry by entry; a weight set without this structure keeps the dense filter (`*ok` = false).
//   F[0..159]   W4[t+1][4] = {A(t), bulge weight wb(t), Bstep(t), centre residual Rc(t)}, t = -1..38 (zero outside 0..30)
//   F[160..191] Bp[parity][j] = B(parity + 2j)
//   F[192..231] Rx[t][l1-1], t = 0..9, l1 = 1..4: residuals of the shapes with 1 <= l1 <= 4 that are neither bulge end nor centre
std::vector<double> strip_weights(const LinModel& L, bool* ok_out = nullptr)
{
    std::vector<double> wT(kStripFiltOff + kStripFiltLen + kSmallWLen, 0.0);
    for (int t = 0; t <= kMaxSingle; t++)   // zero-padded rows for mccaskill_small.hip
        for (int l1 = 0; l1 <= t; l1++) wT[kStripFiltOff + kStripFiltLen + t * 32 + l1] = L.shape_w[t * (t + 1) / 2 + l1];
    double W[31][31] = {};
    for (int t = 0; t <= kMaxSingle; t++)
        for (int l1 = 0; l1 <= t; l1++) { W[t][l1] = L.shape_w[t * (t + 1) / 2 + l1]; wT[(size_t)l1 * 40 + t + 1] = W[t][l1]; }
    double* F = wT.data() + kStripFiltOff;
    double B[40] = {}, A[40] = {};
    bool ok = W[30][14] > 0.0 && W[29][14] > 0.0;
    if (ok) {
        for (int k = 2; k <= 28; k += 2) B[k] = W[30][15 - k / 2] / W[30][14];     // gauge B(2) = 1 on the even, B(1) = 1 on the odd differences
        B[0] = B[2];                                                                  // (the centre tap carries the symmetric term: residual)
        for (int k = 1; k <= 27; k += 2) B[k] = W[29][(29 - k) / 2] / W[29][14];
        for (int t = 3; t <= 30; t++) A[t] = B[t - 2] > 0.0 ? W[t][1] / B[t - 2] : 0.0;
    }
    for (int t = 0; ok && t <= 30; t++) {
        for (int l1 = 0; l1 <= t; l1++) {
            const int l2 = t - l1;
            const double ab = (l1 >= 1 && l2 >= 1) ? A[t] * B[std::abs(l1 - l2)] : 0.0;
            double R = W[t][l1] - ab;
            if (std::fabs(R) <= 1e-13 * std::fabs(W[t][l1])) R = 0.0;
            if (R == 0.0) continue;
            if (l1 == 0 || l2 == 0) { if (W[t][0] != W[t][t]) ok = false; F[(t + 1) * 4 + 1] = W[t][0]; }   // one bulge weight per length
            else if (l1 == l2) F[(t + 1) * 4 + 3] = R;
            else if (l1 <= 4 && l2 <= 4) F[192 + t * 4 + (l1 - 1)] = R;                                      // (t <= 8)
            else ok = false;
        }
        F[(t + 1) * 4 + 0] = A[t];
        F[(t + 1) * 4 + 2] = t == 0 ? 0.5 * B[0] : (t <= 28 ? B[t] : 0.0);
    }
    for (int j = 0; j < 16; j++) { F[160 + j] = 2 * j <= 28 ? B[2 * j] : 0.0; F[176 + j] = 2 * j + 1 <= 27 ? B[2 * j + 1] : 0.0; }
    // verification: the factored form reproduces every weight
    for (int t = 0; ok && t <= 30; t++)
        for (int l1 = 0; l1 <= t; l1++) {
            const int l2 = t - l1;
            double w = (l1 >= 1 && l2 >= 1) ? A[t] * B[std::abs(l1 - l2)] : 0.0;
            if (l1 == 0 || l2 == 0) w += t >= 1 ? F[(t + 1) * 4 + 1] : 0.0;
            else if (l1 == l2) w += F[(t + 1) * 4 + 3];
            else if (l1 <= 4 && l2 <= 4) w += F[192 + t * 4 + (l1 - 1)];
            if (std::fabs(w - W[t][l1]) > 1e-12 * std::fabs(W[t][l1])) ok = false;
        }
    if (ok && (F[(0 + 1) * 4 + 1] != 0.0 || F[(1 + 1) * 4 + 1] != 0.0)) ok = false;   // shapes (0,0), (0,1), (1,0) are not filter taps (weight 0 here)
    if (ok_out) *ok_out = ok;
    return wT;
}

// ---- other scale exponents before the log-space kernels (CONTRAfold model).  The scaled linear path stores Q * exp(-s * span): with
// s = 0.12 (random ACGU: log Z per nucleotide 0.11-0.13) a 1100-nt RNA made of stable hairpins (0.58 per nucleotide) passes 1e200.
// Such a sequence is recomputed on the SAME kernels with a larger exponent -- rung 0: s = 0.45, rung 1: s = 1.5 -- and one whose
// values vanish (log Z per nucleotide far below 0.12 on a long sequence) with rung 2: s = 0; only what leaves the range there too
// goes to the log-space kernels (3x slower).  An exponent costs dynamic range only, never accuracy: a cell that underflows to 0
// while Z~ stays inside (1e-200, 1e200) is below 1e-100 of the terms that make up Z.  Flagged sequences are compacted into a
// sub-batch that reuses the (dead) tables of the main pass; results are scattered back into their slots.
constexpr double kRungS[rh_ctx::kRungs] = {0.45, 1.5, 0.0};

int ensure_rungs(rh_ctx* c)
{
    if (c->h_lin_r) return RH_OK;
    c->h_lin_r = new LinModel[rh_ctx::kRungs];
    for (int r = 0; r < rh_ctx::kRungs; r++) {
        build_lin_model(*c->h_score, kRungS[r], &c->h_lin_r[r]);
        bool fok = false;
        const std::vector<double> wT = strip_weights(c->h_lin_r[r], &fok);
        if (!fok) c->strip_filt_ok = false;   // (the structure does not depend on the exponent; kept as a guard)
        HIP_TRY(c, hipMalloc((void**)&c->d_lin_r[r], sizeof(LinModel)));
        HIP_TRY(c, hipMemcpy(c->d_lin_r[r], &c->h_lin_r[r], sizeof(LinModel), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMalloc((void**)&c->d_wT_r[r], sizeof(double) * wT.size()));
        HIP_TRY(c, hipMemcpy(c->d_wT_r[r], wT.data(), sizeof(double) * wT.size(), hipMemcpyHostToDevice));
    }
    return RH_OK;
}

// `rest`: the sequences the main pass flagged; on return those that no rung could hold (for the log-space kernels)
int retry_mc_lin_rungs(rh_ctx* c, std::vector<int>* rest)
{
    if (!c->scale_ladder || rest->empty()) return RH_OK;
    int rc;
    if ((rc = ensure_rungs(c))) return rc;
    const McBatch B = c->mc;
    const size_t up_per = (size_t)B.ld * c->max_w;
    // direction: log Z = log(Z~) + s n of the failed pass is +Inf / NaN / large after an overflow, -Inf or below s n after an underflow
    std::vector<double> lz(B.ns);
    HIP_TRY(c, hipMemcpyAsync(lz.data(), c->d_mclogz, sizeof(double) * B.ns, hipMemcpyDeviceToHost, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    std::vector<int> over, under;
    for (int k : *rest) ((lz[k] == lz[k] && lz[k] < c->h_lin.s * c->n[k]) ? under : over).push_back(k);
    const LinModel saved_h = c->h_lin;
    LinModel* const saved_d = c->d_lin;
    double* const saved_wT = c->d_wT;
    for (int& q : c->rescued_by) q = 0;
    // the exponents to try: larger ones in ascending order for the overflows, smaller ones in descending order for the underflows
    // (model -1 = the default exponent, when this pass ran on a rung)
    struct Try { int model; bool up; };
    std::vector<Try> tries;
    {
        std::vector<std::pair<double, int>> all = {{c->h_lin0.s, -1}};
        for (int r = 0; r < rh_ctx::kRungs; r++) all.push_back({kRungS[r], r});
        std::sort(all.begin(), all.end());
        for (const auto& e : all) if (e.first > saved_h.s + 1e-12) tries.push_back({e.second, true});
        for (auto it = all.rbegin(); it != all.rend(); ++it) if (it->first < saved_h.s - 1e-12) tries.push_back({it->second, false});
    }
    const int saved_nl[3] = {c->n_launch[0], c->n_launch[1], c->n_launch[2]}, saved_nf[3] = {c->n_far[0], c->n_far[1], c->n_far[2]};
    const auto restore = [&] {
        c->mc = B; c->h_lin = saved_h; c->d_lin = saved_d; c->d_wT = saved_wT;
        for (int q = 0; q < 3; q++) { c->n_launch[q] = saved_nl[q]; c->n_far[q] = saved_nf[q]; }
    };
    for (const Try& t : tries) {
        const int rung = t.model;
        std::vector<int>& F = t.up ? over : under;
        const int nsub = (int)F.size();
        if (!nsub) continue;
        std::vector<uint8_t> codes((size_t)nsub * B.lds);
        std::vector<int> lens(nsub);
        int nmax = 0;
        for (int k = 0; k < nsub; k++) {
            std::memcpy(codes.data() + (size_t)k * B.lds, c->h_codes.data() + (size_t)F[k] * B.lds, B.lds);
            lens[k] = c->n[F[k]];
            nmax = std::max(nmax, lens[k]);
        }
        if ((rc = ensure(c, &c->d_subseq, &c->cap_subseq, codes.size(), false))) return rc;
        if ((rc = ensure(c, &c->d_subn, &c->cap_subn, sizeof(int) * nsub, false))) return rc;
        if ((rc = ensure(c, &c->d_subbp, &c->cap_subbp, sizeof(double) * B.tri_stride * nsub, false))) return rc;
        if ((rc = ensure(c, &c->d_subup, &c->cap_subup, sizeof(double) * (up_per + 1) * nsub, false))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->d_subseq, codes.data(), codes.size(), hipMemcpyHostToDevice, c->s_mc));
        HIP_TRY(c, hipMemcpyAsync(c->d_subn, lens.data(), sizeof(int) * nsub, hipMemcpyHostToDevice, c->s_mc));
        HIP_TRY(c, hipMemsetAsync(c->d_subbp, 0, sizeof(double) * B.tri_stride * nsub, c->s_mc));
        // the tables (and packed operand tiles) the sub-batch reuses hold Inf / NaN of the failed pass: the fast path masks operands by
        // multiplying with 0 in places
        HIP_TRY(c, hipMemsetAsync(c->d_mctab, 0, sizeof(double) * B.seq_stride * nsub, c->s_mc));
        if (c->d_pk) HIP_TRY(c, hipMemsetAsync(c->d_pk, 0, sizeof(double) * B.pk_stride * kPkCopies * nsub, c->s_mc));
        HIP_TRY(c, hipStreamSynchronize(c->s_mc));   // the host staging vectors die with this scope
        McBatch S = B;                 // same ld / strides
        S.ns = nsub; S.nmax = nmax;
        S.seq = (const uint8_t*)c->d_subseq; S.n = (const int*)c->d_subn;
        S.f5i = (double*)c->d_f5; S.f5o = (double*)c->d_f5 + (size_t)B.ld * nsub;
        S.bp = (double*)c->d_subbp; S.up = (double*)c->d_subup;
        c->mc = S;
        if (rung < 0) { c->h_lin = c->h_lin0; c->d_lin = c->d_lin0; c->d_wT = c->d_wT0; }
        else { c->h_lin = c->h_lin_r[rung]; c->d_lin = c->d_lin_r[rung]; c->d_wT = c->d_wT_r[rung]; }
        const int spin = nsub % 8 == 0 ? 1 : 0;
        rc = launch_mc_lin_any(c, spin, 0);
        if (!rc) rc = launch_mc_lin_any(c, spin, 1);
        if (rc) { restore(); return rc; }
        // flags and log Z of the sub-batch are the first nsub entries of d_bad / d_mclogz
        std::vector<int> sbad(nsub);
        std::vector<double> slz(nsub);
        hipError_t e = hipMemcpyAsync(sbad.data(), c->d_bad, sizeof(int) * nsub, hipMemcpyDeviceToHost, c->s_mc);
        if (e == hipSuccess) e = hipMemcpyAsync(slz.data(), c->d_mclogz, sizeof(double) * nsub, hipMemcpyDeviceToHost, c->s_mc);
        if (e == hipSuccess) e = hipStreamSynchronize(c->s_mc);
        if (e != hipSuccess) { restore(); return fail(c, RH_ERR_HIP, "scale ladder: %s", hipGetErrorString(e)); }
        std::vector<int> still;
        for (int k = 0; k < nsub; k++) {
            if (sbad[k]) { still.push_back(F[k]); continue; }
            e = hipMemcpyAsync((double*)c->d_bp + (size_t)F[k] * B.tri_stride, (double*)c->d_subbp + (size_t)k * B.tri_stride,
                               sizeof(double) * B.tri_stride, hipMemcpyDeviceToDevice, c->s_mc);
            if (e == hipSuccess)
                e = hipMemcpyAsync((double*)c->d_up + (size_t)F[k] * up_per, (double*)c->d_subup + (size_t)k * up_per, sizeof(double) * up_per,
                                   hipMemcpyDeviceToDevice, c->s_mc);
            if (e != hipSuccess) { restore(); return fail(c, RH_ERR_HIP, "scale ladder: %s", hipGetErrorString(e)); }
            lz[F[k]] = slz[k];
            c->rescaled_mc.push_back(F[k]);
            c->rescued_by[rung + 1]++;
        }
        F.swap(still);
        restore();   // (every early return below finds the context as it was)
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_mclogz, lz.data(), sizeof(double) * B.ns, hipMemcpyHostToDevice, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));   // (lz dies with this scope)
    rest->clear();
    rest->insert(rest->end(), over.begin(), over.end());
    rest->insert(rest->end(), under.begin(), under.end());
    std::sort(rest->begin(), rest->end());
    std::sort(c->rescaled_mc.begin(), c->rescaled_mc.end());
    return RH_OK;
}

int recompute_dx_subset_log(rh_ctx* c, const std::vector<int>& F)
{
    const DxBatch& D = c->dx;
    const int nsub = (int)F.size();
    std::vector<uint8_t> codes((size_t)2 * nsub * D.lds);
    std::vector<int> lens(2 * (size_t)nsub);
    int n1max = 0, n2max = 0;
    for (int k = 0; k < nsub; k++)
        for (int h = 0; h < 2; h++) {
            std::memcpy(codes.data() + (size_t)(2 * k + h) * D.lds, c->h_codes.data() + (size_t)(2 * F[k] + h) * D.lds, D.lds);
            lens[2 * k + h] = c->n[2 * F[k] + h];
            (h ? n2max : n1max) = std::max(h ? n2max : n1max, lens[2 * k + h]);
        }
    int rc;
    if ((rc = ensure(c, &c->d_subdseq, &c->cap_subdseq, codes.size(), false))) return rc;
    if ((rc = ensure(c, &c->d_subdn, &c->cap_subdn, sizeof(int) * lens.size(), false))) return rc;
    // own tables: the linear duplex image keeps zero pad columns between runs, which the log-space layout would overwrite
    if ((rc = ensure(c, &c->d_subdx, &c->cap_subdx, sizeof(double) * (D.pair_stride + D.tab_stride + 1) * nsub, false))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_subdseq, codes.data(), codes.size(), hipMemcpyHostToDevice, c->s_dx));
    HIP_TRY(c, hipMemcpyAsync(c->d_subdn, lens.data(), sizeof(int) * lens.size(), hipMemcpyHostToDevice, c->s_dx));
    double* sub_hp = (double*)c->d_subdx + D.pair_stride * nsub;
    double* sub_logz = sub_hp + D.tab_stride * nsub;
    HIP_TRY(c, hipMemsetAsync(sub_hp, 0, sizeof(double) * D.tab_stride * nsub, c->s_dx));   // row 0 / column 0 stay zero
    HIP_TRY(c, hipStreamSynchronize(c->s_dx));
    DxBatch S = D;
    S.np = nsub; S.n1max = n1max; S.n2max = n2max;
    S.seq = (const uint8_t*)c->d_subdseq; S.n = (const int*)c->d_subdn;
    S.tab = (double*)c->d_subdx; S.hp = sub_hp; S.logz = sub_logz;
    if ((rc = launch_dx_log(c, S))) return rc;
    for (int k = 0; k < nsub; k++) {
        HIP_TRY(c, hipMemcpyAsync((double*)c->d_hp + (size_t)F[k] * D.tab_stride, sub_hp + (size_t)k * D.tab_stride, sizeof(double) * D.tab_stride,
                                  hipMemcpyDeviceToDevice, c->s_dx));
        HIP_TRY(c, hipMemcpyAsync((double*)c->d_logz + F[k], sub_logz + k, sizeof(double), hipMemcpyDeviceToDevice, c->s_dx));
    }
    return RH_OK;
}

// ---- duplex sweeps, scaled linear path
// X: the batch (the whole one, or a compacted sub-batch of the scale-exponent ladder with its own tables); dm / hm: the model at the
// scale exponent of this pass; logz_out / bad: per pair of X
template <int W>
int launch_dx_lin_on(rh_ctx* c, DxLinBatch X, const DxLinModel* dm, const DxLinModel& hm, double* logz_out, int* bad)
{
    const int smax = X.n1max + X.n2max;
    const int steps = smax / 2;
    const int groups = (X.n1max + 2 + 63) / 64;
    const double leu = hm.lam_eu, l2 = hm.lam_pow[2];
    if (W == 4 && c->dx_quad && c->dx_strip) {   // eight anti-diagonals per launch (dxl_strip8)
        const int groups8 = (X.n1max + 2 + 57) / 58;
        for (int t = 0; 8 * t < smax - 1; t++) {
            for (int k = 0; k < 8; k++) X.pw8[k] = std::pow(leu, 8.0 * t + k) * l2;
            KLAUNCH(c, 4, dxl_strip8, dim3(groups8, X.np, 2), dim3(512), c->s_dx, X, dm, t);
            c->n_launch[2]++;
        }
    } else
    if (W == 4 && c->dx_quad) {   // four anti-diagonals per launch (dxl_sweep4)
        const int groups4 = (X.n1max + 2 + 61) / 62;
        for (int t = 0; 4 * t < smax - 1; t++) {
            for (int k = 0; k < 4; k++) X.pw4[k] = std::pow(leu, 4.0 * t + k) * l2;
            KLAUNCH(c, 4, dxl_sweep4, dim3(groups4, X.np, 2), dim3(256), c->s_dx, X, dm, t, groups4);
            c->n_launch[2]++;
        }
    } else
    for (int t = 0; t < steps; t++) {
        // inside diagonal sd = 2+2t+k: (lam e^eu)^(sd-2) lam^2 ; outside sd = Smax-2t-1+k: (lam e^eu)^(2t+1-k) lam^2
        X.pw_in[0] = std::pow(leu, 2.0 * t) * l2;      X.pw_in[1] = X.pw_in[0] * leu;
        X.pw_out[1] = std::pow(leu, 2.0 * t) * l2;     X.pw_out[0] = X.pw_out[1] * leu;
        KLAUNCH(c, 4, dxl_sweep<W>, dim3(groups, X.np, 2), dim3(64 * W), c->s_dx, X, dm, t, groups);
        c->n_launch[2]++;
    }
    double* zpart = (double*)c->d_zpart;
    int* cpart = (int*)(zpart + (size_t)X.np * c->lz_chunks);
    hipLaunchKernelGGL(dxl_logz_part, dim3(c->lz_chunks, X.np), dim3(256), 0, c->s_dx, X, dm, zpart, cpart, c->lz_chunks);
    hipLaunchKernelGGL(dxl_logz_final, dim3((X.np + 63) / 64), dim3(64), 0, c->s_dx, X, dm, (const double*)zpart, (const int*)cpart,
                       c->lz_chunks, (double*)c->d_zbar, logz_out, bad);
    hipLaunchKernelGGL(dxl_posterior, dim3((X.n1max + 31) / 32, (smax - 1 + 31) / 32, X.np), dim3(256), 0, c->s_dx, X, (const double*)c->d_zbar, bad);
    return RH_OK;
}
template <int W>
int launch_dx_lin(rh_ctx* c) { return launch_dx_lin_on<W>(c, c->dxl, c->d_dxlin, c->h_dxlin, (double*)c->d_logz, (int*)c->d_dxbad); }

// ---- other scale exponents for the duplex sweeps before the log-space kernels (CONTRAfold model): the duplex counterpart of
// retry_mc_lin_rungs.  IN~ = IN * exp(-s (a+b)) with s = 0.65 per unit of a+b (random ACGU: log Z grows by 0.6 per unit); a pair of
// long complementary strands (1.0 - 1.5 per unit) leaves the double range.  The flagged pairs are recomputed as a compacted sub-batch on
// the SAME linear kernels with s = 1.3, then 2.2 (overflow), then 0.3, then 0 (underflow), with their own tables (zero pad columns),
// and only what no exponent holds goes to the log-space kernels.  An exponent costs dynamic range only, never accuracy.
constexpr int kDxRungs = 4;
constexpr double kDxRungS[kDxRungs] = {1.3, 2.2, 0.3, 0.0};

int retry_dx_lin_rungs(rh_ctx* c, std::vector<int>* rest)
{
    if (!c->scale_ladder || rest->empty() || !c->h_score) return RH_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const DxBatch& D = c->dx;
    int rc;
    if (!c->d_dxlin_r[0]) {
        for (int r = 0; r < kDxRungs; r++) {
            build_dx_lin_model(*c->h_score, kDxRungS[r], &c->h_dxlin_r[r]);
            HIP_TRY(c, hipMalloc((void**)&c->d_dxlin_r[r], sizeof(DxLinModel)));
            HIP_TRY(c, hipMemcpy(c->d_dxlin_r[r], &c->h_dxlin_r[r], sizeof(DxLinModel), hipMemcpyHostToDevice));
        }
    }
    std::vector<int> F = *rest;
    for (int r = 0; r < kDxRungs && !F.empty(); r++) {
        const int nsub = (int)F.size();
        std::vector<uint8_t> codes((size_t)2 * nsub * D.lds);
        std::vector<int> lens(2 * (size_t)nsub);
        int n1max = 0, n2max = 0;
        for (int k = 0; k < nsub; k++)
            for (int h = 0; h < 2; h++) {
                std::memcpy(codes.data() + (size_t)(2 * k + h) * D.lds, c->h_codes.data() + (size_t)(2 * F[k] + h) * D.lds, D.lds);
                lens[2 * k + h] = c->n[2 * F[k] + h];
                (h ? n2max : n1max) = std::max(h ? n2max : n1max, lens[2 * k + h]);
            }
        DxLinBatch X = c->dxl;   // lds, ldd, hp_stride as in the main batch: results scatter back row for row
        X.np = nsub; X.n1max = n1max; X.n2max = n2max;
        X.lda = (n1max + 2 + 2 * kDxPad + 1) & ~1;
        const size_t rows = (size_t)n1max + n2max + 3;
        X.tab_stride = rows * X.lda + 128;
        X.pair_stride = X.tab_stride * (int)DL_COUNT;
        const size_t tab_d = X.pair_stride * nsub, hp_d = D.tab_stride * nsub;
        if ((rc = ensure(c, &c->d_subdseq, &c->cap_subdseq, codes.size(), false))) return rc;
        if ((rc = ensure(c, &c->d_subdn, &c->cap_subdn, sizeof(int) * lens.size(), false))) return rc;
        if ((rc = ensure(c, &c->d_subdx, &c->cap_subdx, sizeof(double) * (tab_d + hp_d + nsub + 1) + sizeof(int) * nsub, false))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->d_subdseq, codes.data(), codes.size(), hipMemcpyHostToDevice, c->s_dx));
        HIP_TRY(c, hipMemcpyAsync(c->d_subdn, lens.data(), sizeof(int) * lens.size(), hipMemcpyHostToDevice, c->s_dx));
        double* sub_tab = (double*)c->d_subdx;
        double* sub_hp = sub_tab + tab_d;
        double* sub_logz = sub_hp + hp_d;
        int* sub_bad = (int*)(sub_logz + nsub + 1);
        HIP_TRY(c, hipMemsetAsync(c->d_subdx, 0, sizeof(double) * (tab_d + hp_d + nsub + 1) + sizeof(int) * nsub, c->s_dx));   // pad columns, row 0 / column 0 of hp
        HIP_TRY(c, hipStreamSynchronize(c->s_dx));   // (codes / lens die with this iteration)
        X.seq = (const uint8_t*)c->d_subdseq; X.n = (const int*)c->d_subdn; X.tab = sub_tab; X.hp = sub_hp;
        const int saved_chunks = c->lz_chunks;
        c->lz_chunks = (n1max + n2max - 1 + 15) / 16;   // (<= the main batch's: d_zpart / d_zbar are large enough)
        rc = launch_dx_lin_on<4>(c, X, c->d_dxlin_r[r], c->h_dxlin_r[r], sub_logz, sub_bad);
        c->lz_chunks = saved_chunks;
        if (rc) return rc;
        std::vector<int> bad(nsub);
        HIP_TRY(c, hipMemcpyAsync(bad.data(), sub_bad, sizeof(int) * nsub, hipMemcpyDeviceToHost, c->s_dx));
        HIP_TRY(c, hipStreamSynchronize(c->s_dx));
        std::vector<int> still;
        for (int k = 0; k < nsub; k++) {
            if (bad[k]) { still.push_back(F[k]); continue; }
            HIP_TRY(c, hipMemcpyAsync((double*)c->d_hp + (size_t)F[k] * D.tab_stride, sub_hp + (size_t)k * D.tab_stride, sizeof(double) * D.tab_stride,
                                      hipMemcpyDeviceToDevice, c->s_dx));
            HIP_TRY(c, hipMemcpyAsync((double*)c->d_logz + F[k], sub_logz + k, sizeof(double), hipMemcpyDeviceToDevice, c->s_dx));
            c->rescaled_dx.push_back(F[k]);
        }
        HIP_TRY(c, hipStreamSynchronize(c->s_dx));   // the sub-batch buffers are reused by the next rung
        F.swap(still);
    }
    std::sort(c->rescaled_dx.begin(), c->rescaled_dx.end());
    *rest = F;
    return RH_OK;
}
// Vienna-BL pf_duplex, scaled linear space (duplex_vlin.hip)
int launch_dx_vlin(rh_ctx* c)
{
    DxLinBatch X = c->dxl;
    const int smax = X.n1max + X.n2max;
    const int groups4 = (X.n1max + 2 + 61) / 62;
    const double lam = std::exp(-c->vdx_s);
    for (int t = 0; 4 * t < smax - 1; t++) {
        for (int k = 0; k < 4; k++) X.pw4[k] = std::pow(lam, 2.0 + 4.0 * t + k);
        KLAUNCH(c, 4, dxvl_sweep4, dim3(groups4, X.np, 2), dim3(256), c->s_dx, X, c->d_vdxl, c->d_vdx, t);
        c->n_launch[2]++;
    }
    double* zpart = (double*)c->d_zpart;
    int* cpart = (int*)(zpart + (size_t)X.np * c->lz_chunks);
    hipLaunchKernelGGL(dxvl_logz_part, dim3(c->lz_chunks, X.np), dim3(256), 0, c->s_dx, X, c->d_vdxl, c->d_vdx, zpart, cpart, c->lz_chunks);
    hipLaunchKernelGGL(dxvl_logz_final, dim3((X.np + 63) / 64), dim3(64), 0, c->s_dx, X, c->vdx_s, (const double*)zpart, (const int*)cpart,
                       c->lz_chunks, (double*)c->d_zbar, (double*)c->d_logz, (int*)c->d_dxbad);
    hipLaunchKernelGGL(dxl_posterior, dim3((X.n1max + 31) / 32, (smax - 1 + 31) / 32, X.np), dim3(256), 0, c->s_dx, X, (const double*)c->d_zbar,
                       (int*)c->d_dxbad);
    return RH_OK;
}
int launch_dx_vlog(rh_ctx* c)
{
    const DxBatch& D = c->dx;
    const int steps = (D.n1max + D.n2max) / 2;
    const int waves = 2 * std::min(D.n1max, D.n2max);
    for (int t = 0; t < steps; t++) {
        KLAUNCH(c, 4, dxv_sweep_diag, dim3((waves + 3) / 4, D.np, 2), dim3(256), c->s_dx, D, c->d_vienna, t);
        c->n_launch[2]++;
    }
    hipLaunchKernelGGL(dxv_logz, dim3(D.np), dim3(1024), 0, c->s_dx, D, c->d_vienna);
    hipLaunchKernelGGL(dxv_posterior, dim3((D.n1max * D.n2max + 255) / 256, D.np), dim3(256), 0, c->s_dx, D);
    return RH_OK;
}
int launch_dx_lin_any(rh_ctx* c)
{
    switch (c->dx_w) {
        case 2: return launch_dx_lin<2>(c);
        case 8: return launch_dx_lin<8>(c);
        default: return launch_dx_lin<4>(c);
    }
}

size_t shape_key(const rh_ctx* c, int which)
{
    auto mix = [](size_t h, size_t v) { return (h ^ v) * 0x100000001b3ull + 0x9e3779b97f4a7c15ull; };
    size_t h = 1469598103934665603ull + which;
    if (which == 3) {
        const McBatch& B = c->co;
        for (size_t v : {(size_t)B.ns, (size_t)B.nmax, (size_t)B.ld, (size_t)B.lds, (size_t)B.tab, (size_t)B.seq, (size_t)B.n, (size_t)B.f5i,
                         (size_t)B.bp, (size_t)c->d_cobad, (size_t)c->lin_bs, (size_t)B.tri_stride, (size_t)c->dx.hp, (size_t)c->dx.logz,
                         (size_t)c->dx.ldd, (size_t)c->dx.tab_stride, (size_t)c->dx.n1max, (size_t)c->dx.n2max, (size_t)B.allow, (size_t)B.pk,
                         (size_t)c->far_pk, (size_t)B.rowp, (size_t)c->lookahead, (size_t)B.seeded, (size_t)c->mc.tab, (size_t)c->mc.ld, (size_t)c->d_vlin,
                         // the windowed grid and its pin offset are baked into the captured launches (launch_mc_vlin)
                         (size_t)c->co_window, (size_t)(c->co_cut_min + 1), (size_t)(c->co_cut_max + 1)})
            h = mix(h, v);
    } else if (which <= 1) {
        const McBatch& B = c->mc;
        for (size_t v : {(size_t)B.ns, (size_t)B.nmax, (size_t)B.ld, (size_t)B.lds, (size_t)B.tab, (size_t)B.seq, (size_t)B.n,
                         (size_t)B.f5i, (size_t)B.bp, (size_t)B.up, (size_t)c->d_bad, (size_t)c->d_mclogz, (size_t)c->lin_w, (size_t)c->lin_w_in,
                         (size_t)c->lin_bs, (size_t)B.tri_stride, (size_t)c->far_mfma, (size_t)c->max_w, (size_t)c->d_gaps,
                         (size_t)c->d_hplen, (size_t)B.allow, (size_t)B.pk, (size_t)c->far_pk, (size_t)B.rowp, (size_t)c->lookahead, (size_t)c->strip, (size_t)c->d_wT, (size_t)c->strip_w, (size_t)(c->strip_filt && c->strip_filt_ok), (size_t)c->strip_xcd, (size_t)(c->far2 + 2), (size_t)c->acc_wide, (size_t)c->acc_final_t, (size_t)c->d_vlin,
                         (size_t)c->small_on, c->small_list.size(), (size_t)c->nmax_sweep, (size_t)c->d_small_list, (size_t)c->d_n_sweep,
                         (size_t)c->n_short, (size_t)c->nmax_short, (size_t)c->d_n_short})
            h = mix(h, v);
    } else {
        const DxLinBatch& X = c->dxl;
        for (size_t v : {(size_t)X.np, (size_t)X.n1max, (size_t)X.n2max, (size_t)X.lda, (size_t)X.ldd, (size_t)X.tab, (size_t)X.hp,
                         (size_t)X.seq, (size_t)X.n, (size_t)c->d_zbar, (size_t)c->d_logz, (size_t)c->d_dxbad, (size_t)c->dx_w, (size_t)c->d_zpart, (size_t)c->dx_quad, (size_t)c->dx_strip})
            h = mix(h, v);
    }
    return h;
}

int compute_once(rh_ctx* c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    c->deferred = false;
    c->went_log = false;
    c->tev_n = 0;
    c->n_launch[0] = c->n_launch[1] = c->n_launch[2] = 0;
    c->n_far[0] = c->n_far[1] = c->n_far[2] = 0;
    c->last_path = 0;
    c->fallback_mc.clear(); c->fallback_dx.clear(); c->rescaled_mc.clear(); c->rescaled_dx.clear();
    // sequence -> XCD affinity only when the batch spreads evenly over the 8 XCDs (speed only)
    const int pin = (c->has_mc && c->mc.ns % 8 == 0) ? 1 : 0;
    int rc;
    // duplex first on its own stream: it is independent of the McCaskill sweeps and overlaps them
    HIP_TRY(c, hipEventRecord(c->ev[3], c->s_dx));
    bool dx_lin_launched = false, co_lin_launched = false, co_seed_bad = false, out_from_ev5 = false;
    // two-molecule sweeps in linear space next to the single-molecule folds of the same pairs (no structure constraints): the cells
    // on one strand are copied from those folds (vlin_co_seed), so the sweeps over s1+s2 start when their inside tables are final
    const bool co_seed = c->has_dx && c->has_mc && c->model == RH_MODEL_VIENNA_BL && c->hybrid == RH_HYBRID_COFOLD && c->mode != RH_MODE_LOG &&
                         c->co_seed && !c->mc.allow && !c->co.allow && c->mc.ns == 2 * c->co.ns;
    c->co.seeded = co_seed ? 1 : 0;
    auto launch_co_lin = [&]() -> int {   // scaled linear sweeps over s1+s2; out-of-range values send the batch to the log-space kernels
        const int cpin = c->co.ns % 8 == 0 ? 1 : 0;
        const bool far = c->lin_bs != 0;
        return run_graphed(c, c->g_dx, shape_key(c, 3), c->s_dx, &c->n_launch[2], &c->n_far[2], [&] {
            int r = far ? launch_mc_vlin<16>(c, cpin, 0, true) : launch_mc_vlin<0>(c, cpin, 0, true);
            return r ? r : (far ? launch_mc_vlin<16>(c, cpin, 1, true) : launch_mc_vlin<0>(c, cpin, 1, true));
        });
    };
    if (c->has_dx && c->model == RH_MODEL_VIENNA_BL && c->hybrid == RH_HYBRID_COFOLD) {
        HIP_TRY(c, hipMemsetAsync(c->d_cobp, 0, sizeof(double) * c->co.tri_stride * c->co.ns, c->s_dx));
        bool co_log = c->mode == RH_MODE_LOG;
        if (!co_log && !co_seed) {
            if ((rc = launch_co_lin())) return rc;
            c->last_dx_path = 1;
            co_lin_launched = true;   // its overflow flags are read after the McCaskill stream has been fed (the two overlap)
        }
        if (co_log) {
            if ((rc = launch_cofold(c))) return rc;
            if (c->last_dx_path != 3) c->last_dx_path = 2;
        }
    } else if (c->has_dx && c->model == RH_MODEL_VIENNA_BL) {
        if (c->mode != RH_MODE_LOG) {   // scaled linear sweeps; pairs outside the double range send the batch to the log-space kernels
            if ((rc = run_graphed(c, c->g_dx, shape_key(c, 2), c->s_dx, &c->n_launch[2], &c->n_far[2], [&] { return launch_dx_vlin(c); }))) return rc;
            dx_lin_launched = true;
        } else {
            if ((rc = launch_dx_vlog(c))) return rc;
            c->last_dx_path = 2;
        }
    } else if (c->has_dx) {
        if (c->mode != RH_MODE_LOG) {
            if ((rc = run_graphed(c, c->g_dx, shape_key(c, 2), c->s_dx, &c->n_launch[2], &c->n_far[2], [&] { return launch_dx_lin_any(c); }))) return rc;
            dx_lin_launched = true;
        } else {
            if ((rc = launch_dx_log(c))) return rc;
            c->last_dx_path = 2;
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev[4], c->s_dx));
    if (!c->overlap) HIP_TRY(c, hipStreamSynchronize(c->s_dx));   // isolated phase timings: nothing else on the device

    HIP_TRY(c, hipEventRecord(c->ev[0], c->s_mc));
    bool need_log = c->has_mc && c->mode == RH_MODE_LOG && c->model != RH_MODEL_VIENNA_BL;
    if (c->has_mc && c->model == RH_MODEL_VIENNA_BL) {
        bool log_path = c->mode == RH_MODE_LOG;
        if (!log_path) {   // scaled linear sweeps; a sequence that leaves the double range sends the batch to the log-space kernels
            const bool far = c->lin_bs != 0;
            if ((rc = run_graphed(c, c->g_in, shape_key(c, 0), c->s_mc, &c->n_launch[0], &c->n_far[0],
                                  [&] { return far ? launch_mc_vlin<16>(c, pin, 0, false) : launch_mc_vlin<0>(c, pin, 0, false); }))) return rc;
            HIP_TRY(c, hipEventRecord(c->ev[1], c->s_mc));
            if (co_seed) {   // the inside tables of both molecules are final behind ev[1]
                HIP_TRY(c, hipStreamWaitEvent(c->s_dx, c->ev[1], 0));
                HIP_TRY(c, hipEventRecord(c->ev[3], c->s_dx));
                if ((rc = launch_co_lin())) return rc;
                HIP_TRY(c, hipEventRecord(c->ev[4], c->s_dx));
                c->last_dx_path = 1;
                co_lin_launched = true;
                if (!c->overlap) HIP_TRY(c, hipStreamSynchronize(c->s_dx));   // isolated phase timings: nothing else on the device
            }
            HIP_TRY(c, hipEventRecord(c->ev[5], c->s_mc));   // start of the outside phase (= ev[1] unless the seeded sweeps ran in between)
            out_from_ev5 = true;
            if ((rc = run_graphed(c, c->g_out, shape_key(c, 1), c->s_mc, &c->n_launch[1], &c->n_far[1],
                                  [&] { return far ? launch_mc_vlin<16>(c, pin, 1, false) : launch_mc_vlin<0>(c, pin, 1, false); }))) return rc;
            c->last_path = 1;
            if (c->mode == RH_MODE_AUTO) {
                std::vector<int> bad(c->mc.ns);
                HIP_TRY(c, hipMemcpyAsync(bad.data(), c->d_bad, sizeof(int) * c->mc.ns, hipMemcpyDeviceToHost, c->s_mc));
                HIP_TRY(c, hipStreamSynchronize(c->s_mc));
                for (int b : bad) log_path |= (b != 0);
                if (log_path) { c->last_path = 3; c->tables_dirty = true; co_seed_bad = co_seed; }
                if (log_path && c->defer_log) {   // another exponent first (compute): this attempt ends here
                    for (int k = 0; k < c->mc.ns; k++) if (bad[k]) { c->flagged_mc.push_back(k); if (c->has_dx) c->flagged_pairs.push_back(k / 2); }
                    c->deferred = true;
                    log_path = false;
                }
            }
        }
        if (log_path) {
            c->went_log = true;
            out_from_ev5 = false;
            c->n_launch[0] = c->n_launch[1] = 0;
            c->n_far[0] = c->n_far[1] = 0;
            HIP_TRY(c, hipEventRecord(c->ev[0], c->s_mc));
            HIP_TRY(c, hipMemsetAsync(c->d_bp, 0, sizeof(double) * c->mc.tri_stride * c->mc.ns, c->s_mc));
            if ((rc = launch_mc_vienna(c, pin))) return rc;
            if (c->last_path == 0) c->last_path = 2;
        }
    } else if (c->has_mc && c->mode != RH_MODE_LOG) {
        // the exponent most of the last batch needed (scale-exponent ladder); h_lin / d_lin / d_wT are the default's again afterwards
        const bool on_rung = c->mode == RH_MODE_AUTO && c->scale_ladder && c->lin_primary >= 0 && c->h_lin_r;
        if (on_rung) { c->h_lin = c->h_lin_r[c->lin_primary]; c->d_lin = c->d_lin_r[c->lin_primary]; c->d_wT = c->d_wT_r[c->lin_primary]; }
        struct Back { rh_ctx* c; ~Back() { c->h_lin = c->h_lin0; c->d_lin = c->d_lin0; c->d_wT = c->d_wT0; } } back{c};
        if ((rc = run_graphed(c, c->g_in, shape_key(c, 0), c->s_mc, &c->n_launch[0], &c->n_far[0], [&] { return launch_mc_lin_any(c, pin, 0); }))) return rc;
        HIP_TRY(c, hipEventRecord(c->ev[1], c->s_mc));
        if ((rc = run_graphed(c, c->g_out, shape_key(c, 1), c->s_mc, &c->n_launch[1], &c->n_far[1], [&] { return launch_mc_lin_any(c, pin, 1); }))) return rc;
        c->last_path = 1;
        if (c->mode == RH_MODE_AUTO) {  // did every sequence stay inside the double range?
            std::vector<int> bad(c->mc.ns);
            HIP_TRY(c, hipMemcpyAsync(bad.data(), c->d_bad, sizeof(int) * c->mc.ns, hipMemcpyDeviceToHost, c->s_mc));
            HIP_TRY(c, hipStreamSynchronize(c->s_mc));
            for (int k = 0; k < c->mc.ns; k++) if (bad[k]) c->fallback_mc.push_back(k);
            if (!c->fallback_mc.empty()) {
                c->last_path = 3;
                c->tables_dirty = true;
                if ((rc = retry_mc_lin_rungs(c, &c->fallback_mc))) return rc;   // another exponent first; what is left goes to log space
                for (int q = 0; q <= rh_ctx::kRungs; q++)   // more than half of the batch on one exponent: the next batch starts there
                    if (c->scale_memory && c->mc.ns >= 8 && 2 * c->rescued_by[q] > c->mc.ns) c->lin_primary = q - 1;   // (a batch, not a single call)
                if (c->fallback_mc.empty()) { }
                else if (2 * c->fallback_mc.size() > (size_t)c->mc.ns) need_log = true;   // most of the batch: redo it whole
                else if ((rc = recompute_mc_subset_log(c, c->fallback_mc))) return rc;
            }
        }
    } else {
        HIP_TRY(c, hipEventRecord(c->ev[1], c->s_mc));
    }
    if (need_log) {
        c->n_launch[0] = c->n_launch[1] = 0;
        c->n_far[0] = c->n_far[1] = 0;
        HIP_TRY(c, hipEventRecord(c->ev[0], c->s_mc));
        HIP_TRY(c, hipMemsetAsync(c->d_bp, 0, sizeof(double) * c->mc.tri_stride * c->mc.ns, c->s_mc));
        if ((rc = launch_mc_log(c, pin))) return rc;
        if (c->last_path == 0) c->last_path = 2;
    }
    HIP_TRY(c, hipEventRecord(c->ev[2], c->s_mc));
    if (co_lin_launched && c->mode == RH_MODE_AUTO) {
        std::vector<int> bad(c->co.ns);
        HIP_TRY(c, hipMemcpyAsync(bad.data(), c->d_cobad, sizeof(int) * c->co.ns, hipMemcpyDeviceToHost, c->s_dx));
        HIP_TRY(c, hipStreamSynchronize(c->s_dx));
        bool redo = co_seed_bad;   // a molecule left the double range on its own: what was copied from its fold is not usable
        for (int b : bad) redo |= (b != 0);
        if (redo && c->defer_log) {
            c->tables_dirty = true; c->deferred = true;
            for (int k = 0; k < c->co.ns; k++) if (bad[k]) c->flagged_pairs.push_back(k);
        }
        else if (redo) {   // some pair left the double range: recompute the two-molecule sweeps in log space
            c->went_log = true;
            c->tables_dirty = true;
            c->n_launch[2] = 0; c->n_far[2] = 0;
            HIP_TRY(c, hipEventRecord(c->ev[3], c->s_dx));
            HIP_TRY(c, hipMemsetAsync(c->d_cobp, 0, sizeof(double) * c->co.tri_stride * c->co.ns, c->s_dx));
            if ((rc = launch_cofold(c))) return rc;
            HIP_TRY(c, hipEventRecord(c->ev[4], c->s_dx));
            c->last_dx_path = 3;
        }
    }
    if (dx_lin_launched) {
        c->last_dx_path = 1;
        if (c->mode == RH_MODE_AUTO) {
            std::vector<int> bad(c->dx.np);
            HIP_TRY(c, hipMemcpyAsync(bad.data(), c->d_dxbad, sizeof(int) * c->dx.np, hipMemcpyDeviceToHost, c->s_dx));
            HIP_TRY(c, hipStreamSynchronize(c->s_dx));
            bool redo = false;
            for (int k = 0; k < c->dx.np; k++) if (bad[k]) { redo = true; c->fallback_dx.push_back(k); }
            if (redo && c->model != RH_MODEL_VIENNA_BL && 2 * c->fallback_dx.size() <= (size_t)c->dx.np) {
                // only the flagged pairs, as a compacted sub-batch with its own tables: another scale exponent on the linear kernels
                // first (retry_dx_lin_rungs), the log-space kernels for what is left
                if ((rc = retry_dx_lin_rungs(c, &c->fallback_dx))) return rc;
                if (!c->fallback_dx.empty() && (rc = recompute_dx_subset_log(c, c->fallback_dx))) return rc;
                HIP_TRY(c, hipEventRecord(c->ev[4], c->s_dx));
                c->last_dx_path = 3;
            } else if (redo && c->model == RH_MODEL_VIENNA_BL && c->defer_log) {   // (compute: the flagged pairs go to the helper context)
                c->tables_dirty = true; c->deferred = true;
                c->flagged_pairs.insert(c->flagged_pairs.end(), c->fallback_dx.begin(), c->fallback_dx.end());
                c->fallback_dx.clear();
            } else if (redo) {  // most pairs (or the Vienna-BL model): recompute the batch with the log-space kernels
                c->n_launch[2] = 0;
                HIP_TRY(c, hipEventRecord(c->ev[3], c->s_dx));
                if ((rc = (c->model == RH_MODEL_VIENNA_BL ? launch_dx_vlog(c) : launch_dx_log(c)))) return rc;
                HIP_TRY(c, hipEventRecord(c->ev[4], c->s_dx));
                c->last_dx_path = 3;
            }
        }
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_dx));
    float t01 = 0, t12 = 0, t34 = 0, t02 = 0;
    HIP_TRY(c, hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
    HIP_TRY(c, hipEventElapsedTime(&t12, c->ev[out_from_ev5 && !c->overlap ? 5 : 1], c->ev[2]));
    HIP_TRY(c, hipEventElapsedTime(&t02, c->ev[0], c->ev[2]));
    HIP_TRY(c, hipEventElapsedTime(&t34, c->ev[3], c->ev[4]));
    c->ms[0] = t01; c->ms[1] = t12; c->ms[2] = t34; c->ms[3] = std::max(t02, t34);
    c->computed = true;
    return RH_OK;
}

// ---- Vienna-BL: other scale exponents before the log-space kernels.  The linear path stores Q * exp(-s * span) with s = 0.28 (random ACGU
// under the BL* energies: 0.21-0.33 per nucleotide); a 900-nt chain of stable hairpins (0.87) or a ribosomal RNA (~0.6 at 1500 nt)
// passes 1e200, and one such sequence used to send the whole batch -- single-molecule folds, accessibility and the two-molecule
// sweeps -- to the log-space kernels.  Now the batch is run again on the linear kernels with s = 0.7, then 1.8, then 0 (models built
// on first use), and goes to log space only when every exponent left some problem outside the range.  Whole batches, not problems
// (the two-molecule sweeps are seeded from the single folds of the same pass); the exponent that worked is where the next batch of at
// least eight sequences starts.  rh_last_path = 3 and rh_batch_fallbacks(which = 2) = the sequences the first attempts flagged.
constexpr double kVRungS[rh_ctx::kVRungs] = {0.7, 1.8, 0.0};

int select_vlin(rh_ctx* c, int model)
{
    if (model == c->vlin_cur) return RH_OK;
    const int k = model + 1;
    if (!c->h_vlin_m[k]) {
        c->h_vlin_m[k] = new VLinModel;
        build_vlin_model(*c->h_vienna, kVRungS[model], c->h_vlin_m[k]);
        HIP_TRY(c, hipMalloc((void**)&c->d_vlin_m[k], sizeof(VLinModel)));
        HIP_TRY(c, hipMemcpy(c->d_vlin_m[k], c->h_vlin_m[k], sizeof(VLinModel), hipMemcpyHostToDevice));
    }
    c->h_vlin = c->h_vlin_m[k];
    c->d_vlin = c->d_vlin_m[k];
    c->vlin_cur = model;
    // hairpin length weights x lam^d (kernel arguments of the inside sweeps; the device copy serves the accessibility)
    const VLinModel& H = *c->h_vlin;
    for (size_t d = 0; d < c->h_hplen.size(); d++)
        c->h_hplen[d] = (d <= 30 ? H.E_hairpin[d] : std::exp(H.hairpin30 - H.lxc * std::log(d / 30.0))) * std::exp(-H.s * (double)d);
    if (c->d_hplen && c->has_mc) {
        HIP_TRY(c, hipMemcpyAsync(c->d_hplen, c->h_hplen.data(), sizeof(double) * std::min((size_t)c->mc.ld, c->h_hplen.size()), hipMemcpyHostToDevice, c->s_mc));
        HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    }
    return RH_OK;
}

int compute(rh_ctx* c);

// The flagged pairs P of a Vienna-BL batch, recomputed on the helper context and copied into this batch's result buffers (bp, up, hp and
// the three log partition functions of each pair); the layouts differ only in their strides.
int recompute_pairs_on_helper(rh_ctx* c, const std::vector<int>& P)
{
    if (!c->helper) {
        c->helper = make_ctx_for_helper(c->device, c->model, c->p_has_param ? c->p_param.c_str() : nullptr, c->p_has_defaults ? c->p_defaults.c_str() : nullptr,
                               c->p_use_bl, c->p_sem);
        if (!c->helper) return fail(c, RH_ERR_HIP, "helper context for the per-pair fallback could not be created");
        c->helper->is_helper = true;
    }
    rh_ctx* h = c->helper;
    h->max_w = c->max_w; h->hybrid = c->hybrid; h->mode = RH_MODE_AUTO; h->scale_ladder = c->scale_ladder; h->scale_memory = 0;
    h->vlin_primary = c->vlin_primary < 0 ? 0 : -1;   // the exponent of the first pass is known to fail for these pairs: start on the next one
    const int nsub = (int)P.size();
    const int lds = c->mc.lds;
    std::vector<std::string> text(2 * (size_t)nsub);
    std::vector<const char*> ptr(2 * (size_t)nsub);
    std::vector<int> lens(2 * (size_t)nsub);
    for (int k = 0; k < nsub; k++)
        for (int hh = 0; hh < 2; hh++) {
            const int sq = 2 * P[k] + hh, n = c->n[sq];
            std::string& t = text[2 * k + hh];
            t.resize(n);
            for (int i = 0; i < n; i++) t[i] = "NACGU"[c->h_codes[(size_t)sq * lds + 1 + i] <= 4 ? c->h_codes[(size_t)sq * lds + 1 + i] : 0];   // vienna_code^-1
            ptr[2 * k + hh] = t.c_str(); lens[2 * k + hh] = n;
        }
    int rc;
    if ((rc = stage(h, 2 * nsub, ptr.data(), lens.data(), true, true, nullptr, nullptr))) return fail(c, rc, "helper upload: %s", h->err.c_str());
    if ((rc = compute(h))) return fail(c, rc, "helper compute: %s", h->err.c_str());
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_dx));
    const size_t up_c = (size_t)c->mc.ld * c->max_w, up_h = (size_t)h->mc.ld * h->max_w;
    for (int k = 0; k < nsub; k++) {
        for (int hh = 0; hh < 2; hh++) {
            const int sq = 2 * P[k] + hh, sh = 2 * k + hh, n = c->n[sq];
            HIP_TRY(c, hipMemcpyAsync((double*)c->d_bp + (size_t)sq * c->mc.tri_stride, (const double*)h->d_bp + (size_t)sh * h->mc.tri_stride,
                                      sizeof(double) * tri_size(n), hipMemcpyDeviceToDevice, c->s_mc));
            HIP_TRY(c, hipMemcpyAsync((double*)c->d_up + (size_t)sq * up_c, (const double*)h->d_up + (size_t)sh * up_h, sizeof(double) * (size_t)n * c->max_w,
                                      hipMemcpyDeviceToDevice, c->s_mc));
            HIP_TRY(c, hipMemcpyAsync((double*)c->d_mclogz + sq, (const double*)h->d_mclogz + sh, sizeof(double), hipMemcpyDeviceToDevice, c->s_mc));
        }
        const int n1 = c->n[2 * P[k]], n2 = c->n[2 * P[k] + 1];
        HIP_TRY(c, hipMemcpy2DAsync((double*)c->d_hp + (size_t)P[k] * c->dx.tab_stride, sizeof(double) * c->dx.ldd,
                                    (const double*)h->d_hp + (size_t)k * h->dx.tab_stride, sizeof(double) * h->dx.ldd,
                                    sizeof(double) * (n2 + 1), (size_t)n1 + 1, hipMemcpyDeviceToDevice, c->s_mc));
        HIP_TRY(c, hipMemcpyAsync((double*)c->d_logz + P[k], (const double*)h->d_logz + k, sizeof(double), hipMemcpyDeviceToDevice, c->s_mc));
    }
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    // what happened to them: held by another exponent on the helper's linear kernels (which = 2), or recomputed in log space (which = 0 / 1)
    c->last_path = 3;
    if (c->last_dx_path == 1 || c->last_dx_path == 0) c->last_dx_path = 3;
    c->rescaled_mc.clear(); c->fallback_mc.clear(); c->fallback_dx.clear();
    const bool h_log = h->went_log || h->last_path == 2 || (h->last_path == 3 && h->rescaled_mc.empty() && !h->fallback_mc.empty());
    for (int k = 0; k < nsub; k++) {
        if (h_log) { c->fallback_mc.push_back(2 * P[k]); c->fallback_mc.push_back(2 * P[k] + 1); c->fallback_dx.push_back(P[k]); }
        else { c->rescaled_mc.push_back(2 * P[k]); c->rescaled_mc.push_back(2 * P[k] + 1); }
    }
    c->deferred = false;
    c->tables_dirty = true;
    c->computed = true;
    return RH_OK;
}

int compute(rh_ctx* c)
{
    c->defer_log = false;
    c->flagged_mc.clear();
    const bool ladder = c->model == RH_MODEL_VIENNA_BL && c->mode == RH_MODE_AUTO && c->scale_ladder && c->has_mc && c->h_vienna &&
                        c->vienna_sem != kViennaSem20 && !std::getenv("RH_VLIN_S");
    if (!ladder) return compute_once(c);
    HIP_TRY(c, hipSetDevice(c->device));
    // the exponent the batch starts with, then the others: larger ones ascending, smaller ones descending
    std::vector<int> order = {c->vlin_primary};
    {
        std::vector<std::pair<double, int>> all = {{c->h_vlin_m[0]->s, -1}};
        for (int r = 0; r < rh_ctx::kVRungs; r++) all.push_back({kVRungS[r], r});
        std::sort(all.begin(), all.end());
        const double s0 = c->vlin_primary < 0 ? c->h_vlin_m[0]->s : kVRungS[c->vlin_primary];
        for (const auto& e : all) if (e.first > s0 + 1e-12) order.push_back(e.second);
        for (auto it = all.rbegin(); it != all.rend(); ++it) if (it->first < s0 - 1e-12) order.push_back(it->second);
    }
    // (An attempt that failed leaves Inf / NaN in the tables of the flagged sequences; the next attempt runs over them without a clear.
    //  That is sound because the vlin kernels mask every operand by SELECT (`ok ? x : 0.0`), never by a multiplication with 0, and
    //  rewrite every interior cell they read before reading it -- the invariant `tests: test_vienna_bl_scale_exponent_ladder` and
    //  tools/fuzz_ladder.py exercise: chains of hairpins that overflow the first exponent, results equal to the log-space path's.)
    int rc = RH_OK;
    const bool per_pair = !c->is_helper && c->pair_helper && c->has_dx && c->np >= 4 && !c->mc.allow && !c->co.allow;
    for (size_t a = 0; a < order.size(); a++) {
        if ((rc = select_vlin(c, order[a]))) break;
        c->defer_log = a + 1 < order.size();
        c->flagged_pairs.clear();
        if ((rc = compute_once(c))) break;
        if (c->deferred && a == 0 && per_pair) {
            std::sort(c->flagged_pairs.begin(), c->flagged_pairs.end());
            c->flagged_pairs.erase(std::unique(c->flagged_pairs.begin(), c->flagged_pairs.end()), c->flagged_pairs.end());
            // cost: the helper pays the launch latency of a few pairs (tens of ms per attempt at n = 500 - 1000, whatever the batch), a
            // second pass over the batch pays its whole device time again: the helper wins when the flagged pairs are a small share
            // (measured at n = 500: equal at 64 pairs and one flagged pair, 2 x at 256).  RH_PAIR_HELPER=2: whenever at most half are flagged
            const size_t share = c->pair_helper >= 2 ? 2 : 16;
            if (!c->flagged_pairs.empty() && share * c->flagged_pairs.size() <= (size_t)c->np) {
                rc = recompute_pairs_on_helper(c, c->flagged_pairs);
                break;
            }
        }
        if (!c->deferred) {
            if (a > 0) {   // held by another exponent
                c->last_path = 3;
                std::sort(c->flagged_mc.begin(), c->flagged_mc.end());
                c->flagged_mc.erase(std::unique(c->flagged_mc.begin(), c->flagged_mc.end()), c->flagged_mc.end());
                if (!c->went_log) {   // (the last attempt may still have ended in log space)
                    c->rescaled_mc = c->flagged_mc;
                    if (c->scale_memory && c->mc.ns >= 8) c->vlin_primary = order[a];   // a batch, not a single call: the next one starts here
                }
            }
            break;
        }
    }
    c->defer_log = false;
    const int back = select_vlin(c, c->vlin_primary);
    return rc ? rc : back;
}

// copy one sequence's posterior out of the (nmax-strided) device buffer
int fetch_bp(rh_ctx* c, int sq, double* out)
{
    const int n = c->n[sq];
    HIP_TRY(c, hipMemcpy(out, (const double*)c->d_bp + (size_t)sq * c->mc.tri_stride, sizeof(double) * tri_size(n),
                         hipMemcpyDeviceToHost));
    return RH_OK;
}
int fetch_up(rh_ctx* c, int sq, double* out)
{
    HIP_TRY(c, hipMemcpy(out, (const double*)c->d_up + (size_t)sq * c->mc.ld * c->max_w, sizeof(double) * c->n[sq] * c->max_w,
                         hipMemcpyDeviceToHost));
    return RH_OK;
}
int fetch_logz(rh_ctx* c, int sq, double* out)
{
    HIP_TRY(c, hipMemcpy(out, (const double*)c->d_mclogz + sq, sizeof(double), hipMemcpyDeviceToHost));
    return RH_OK;
}
int fetch_hp(rh_ctx* c, int p, double* out, double* logz)
{
    const int n1 = c->n[2 * p], n2 = c->n[2 * p + 1];
    if (out)
        HIP_TRY(c, hipMemcpy2D(out, sizeof(double) * (n2 + 1), (const double*)c->d_hp + (size_t)p * c->dx.tab_stride,
                               sizeof(double) * c->dx.ldd, sizeof(double) * (n2 + 1), n1 + 1, hipMemcpyDeviceToHost));
    if (logz) HIP_TRY(c, hipMemcpy(logz, (const double*)c->d_logz + p, sizeof(double), hipMemcpyDeviceToHost));
    return RH_OK;
}

}  // namespace

extern "C" {

static rh_ctx* create_ctx(int device, int model, const char* param_file, const char* defaults_file, int use_bl, int semantics);

rh_ctx* rh_create(int device, int model, const char* param_file) { return create_ctx(device, model, param_file, nullptr, 1, 0); }
rh_ctx* rh_create_vienna(int device, const char* defaults_file, int use_bl_param, const char* param_file, int semantics)
{
    return create_ctx(device, RH_MODEL_VIENNA_BL, param_file, defaults_file, use_bl_param, semantics);
}
int rh_vienna_semantics(const rh_ctx* c) { return c ? c->vienna_sem : RH_ERR_ARG; }

static rh_ctx* create_ctx(int device, int model, const char* param_file, const char* defaults_file, int use_bl, int semantics)
{
    if (model != RH_MODEL_CONTRAFOLD && model != RH_MODEL_VIENNA_BL) {
        fail(nullptr, RH_ERR_UNSUPPORTED, "unknown model %d", model);
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fail(nullptr, RH_ERR_HIP, "no HIP device available (%s): this library has no CPU fallback",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(nullptr, RH_ERR_ARG, "device %d out of range (have %d)", device, ndev);
        return nullptr;
    }
    ScoreModel host_model;
    ViennaDx* host_vienna = nullptr;
    char err[256];
    if (model == RH_MODEL_CONTRAFOLD) {
        const std::string path = param_file ? std::string(param_file) : default_param_path() + "contrafold_complementary.params";
        if (!load_score_model(path.c_str(), &host_model, err, sizeof err)) {
            fail(nullptr, RH_ERR_PARAM, "%s", err);
            return nullptr;
        }
    } else {
        const std::string bl = default_param_path() + "vienna_bl_star.params";
        host_vienna = new ViennaDx;
        if (!load_vienna_dx_ex(defaults_file, use_bl != 0, bl.c_str(), param_file, semantics, host_vienna, err, sizeof err)) {
            delete host_vienna;
            fail(nullptr, RH_ERR_PARAM, "%s", err);
            return nullptr;
        }
        std::memset(&host_model, 0, sizeof host_model);
    }
    rh_ctx* c = new rh_ctx;
    c->device = device; c->model = model;
    c->max_w = model == RH_MODEL_VIENNA_BL ? 15 : 1;   // RactIP's default --max-w (src/cmdline.c:151-186) / contrafold's width 1
    c->vienna_sem = host_vienna ? host_vienna->semantics : 0;
    if (c->vienna_sem == kViennaSem20) c->mode = RH_MODE_LOG;   // the scaled linear kernels hold the 1.8 semantics only
    // scale exponent of the linear fast path: log Z per nucleotide of typical sequences under this model
    // (random ACGU: 0.107..0.129 for n = 200..2000); deviations only cost dynamic range, never accuracy
    build_lin_model(host_model, 0.12, &c->h_lin);
    c->h_score = new ScoreModel(host_model);
    // duplex: log Z per unit of (i + L2+1-j) is 0.62..0.82 on the bundled pairs, 0.645 for random sequences
    build_dx_lin_model(host_model, 0.65, &c->h_dxlin);
    if (const char* e = std::getenv("RH_LIN_W")) c->lin_w = c->lin_w_in = std::atoi(e);
    if (const char* e = std::getenv("RH_LIN_W_IN")) c->lin_w_in = std::atoi(e);
    if (const char* e = std::getenv("RH_LIN_BS")) c->lin_bs = std::atoi(e);
    if (const char* e = std::getenv("RH_NO_GRAPH")) c->use_graphs = std::atoi(e) == 0;
    if (const char* e = std::getenv("RH_FAR_MFMA")) c->far_mfma = std::atoi(e) != 0;
    if (const char* e = std::getenv("RH_FAR_PK")) c->far_pk = std::atoi(e) != 0;
    if (const char* e = std::getenv("RH_LOOKAHEAD")) c->lookahead = std::atoi(e);
    if (const char* e = std::getenv("RH_STRIP")) c->strip = std::atoi(e);
    if (const char* e = std::getenv("RH_STRIP_W")) c->strip_w = std::atoi(e) == 4 ? 4 : 8;
    if (const char* e = std::getenv("RH_STRIP_FILT")) c->strip_filt = std::atoi(e) != 0;
    if (const char* e = std::getenv("RH_SMALL")) c->small_on = std::atoi(e) != 0;
    if (const char* e = std::getenv("RH_FAR2")) c->far2 = std::atoi(e);
    if (const char* e = std::getenv("RH_STRIP_XCD")) c->strip_xcd = std::atoi(e);
    if (const char* e = std::getenv("RH_ACC_WIDE")) c->acc_wide = std::atoi(e);
    if (const char* e = std::getenv("RH_ACC_FINAL_T")) c->acc_final_t = std::atoi(e);
    if (const char* e = std::getenv("RH_CO_WINDOW")) c->co_window = std::atoi(e);
    if (const char* e = std::getenv("RH_SCALE_LADDER")) c->scale_ladder = std::atoi(e);
    if (const char* e = std::getenv("RH_SCALE_MEMORY")) c->scale_memory = std::atoi(e);
    if (const char* e = std::getenv("RH_PAIR_HELPER")) c->pair_helper = std::atoi(e);
    c->p_has_param = param_file != nullptr; if (param_file) c->p_param = param_file;
    c->p_has_defaults = defaults_file != nullptr; if (defaults_file) c->p_defaults = defaults_file;
    c->p_use_bl = use_bl; c->p_sem = semantics;
    if (const char* e = std::getenv("RH_CO_SEED")) c->co_seed = std::atoi(e);
    if (const char* e = std::getenv("RH_DX_W")) c->dx_w = std::atoi(e);
    if (const char* e = std::getenv("RH_DX_QUAD")) c->dx_quad = std::atoi(e) != 0;
    if (const char* e = std::getenv("RH_DX_STRIP")) c->dx_strip = std::atoi(e);
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&c->s_mc, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->s_dx, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&c->d_model, sizeof(ScoreModel)) == hipSuccess &&
              hipMemcpy(c->d_model, &host_model, sizeof(ScoreModel), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&c->d_lin, sizeof(LinModel)) == hipSuccess &&
              hipMemcpy(c->d_lin, &c->h_lin, sizeof(LinModel), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&c->d_dxlin, sizeof(DxLinModel)) == hipSuccess &&
              hipMemcpy(c->d_dxlin, &c->h_dxlin, sizeof(DxLinModel), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        const std::vector<double> wT = strip_weights(c->h_lin, &c->strip_filt_ok);
        ok = hipMalloc((void**)&c->d_wT, sizeof(double) * wT.size()) == hipSuccess &&
             hipMemcpy(c->d_wT, wT.data(), sizeof(double) * wT.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    c->h_lin0 = c->h_lin; c->d_lin0 = c->d_lin; c->d_wT0 = c->d_wT;
    if (ok && host_vienna) {
        c->h_vlin = new VLinModel;
        // scale exponent: log Z per nucleotide of random ACGU under the BL* energies is 0.21..0.33 for n = 200..500 (up to 0.45 on the bundled RNAs)
        build_vlin_model(*host_vienna, 0.28, c->h_vlin);
        if (const char* e = std::getenv("RH_VLIN_S")) build_vlin_model(*host_vienna, std::atof(e), c->h_vlin);
        ok = hipMalloc((void**)&c->d_vienna, sizeof(ViennaDx)) == hipSuccess &&
             hipMemcpy(c->d_vienna, host_vienna, sizeof(ViennaDx), hipMemcpyHostToDevice) == hipSuccess &&
             hipMalloc((void**)&c->d_vlin, sizeof(VLinModel)) == hipSuccess &&
             hipMemcpy(c->d_vlin, c->h_vlin, sizeof(VLinModel), hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {   // pf_duplex in scaled linear space: the loop tables at the duplex scale + its own end / mismatch weights
            if (const char* e = std::getenv("RH_VDX_S")) c->vdx_s = std::atof(e);
            VLinModel* tmp = new VLinModel;
            VDxLin hd;
            build_vlin_model(*host_vienna, c->vdx_s, tmp);
            build_vdx_lin(*host_vienna, c->vdx_s, &hd);
            ok = hipMalloc((void**)&c->d_vdxl, sizeof(VLinModel)) == hipSuccess &&
                 hipMemcpy(c->d_vdxl, tmp, sizeof(VLinModel), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMalloc((void**)&c->d_vdx, sizeof(VDxLin)) == hipSuccess &&
                 hipMemcpy(c->d_vdx, &hd, sizeof(VDxLin), hipMemcpyHostToDevice) == hipSuccess;
            delete tmp;
        }
    }
    c->h_vienna = host_vienna;   // (kept: the rung models of the scale-exponent ladder are built from it)
    c->h_vlin_m[0] = c->h_vlin; c->d_vlin_m[0] = c->d_vlin;
    for (int k = 0; ok && k < 6; k++) ok = hipEventCreate(&c->ev[k]) == hipSuccess;
    if (!ok) {
        fail(nullptr, RH_ERR_HIP, "context setup failed: %s", hipGetErrorString(hipGetLastError()));
        rh_destroy(c);
        return nullptr;
    }
    return c;
}

void rh_destroy(rh_ctx* c)
{
    if (!c) return;
    if (c->helper) { rh_destroy(c->helper); c->helper = nullptr; }
    (void)hipSetDevice(c->device);
    void* bufs[] = {c->d_seq, c->d_n, c->d_mctab, c->d_f5, c->d_bp, c->d_up, c->d_dxtab, c->d_hp, c->d_logz, c->d_scal, c->d_mclogz, c->d_bad, c->d_cnt, c->d_cand, c->d_dxbad, c->d_zbar, c->d_zpart, c->d_gaps, c->d_coseq, c->d_con, c->d_cotab, c->d_pk, c->d_copk, c->d_rowp, c->d_corowp, c->d_cof5, c->d_cobp, c->d_cobad, c->d_allow, c->d_coallow, c->d_vlin, c->d_vdxl, c->d_vdx, c->d_hplen, c->d_model, c->d_lin, c->d_dxlin, c->d_vienna, c->d_wT, c->d_subseq, c->d_subn, c->d_subbp, c->d_subup, c->d_subdseq, c->d_subdn, c->d_subdx, c->d_small_list, c->d_n_sweep, c->d_n_short, c->d_dxlin_r[0], c->d_dxlin_r[1], c->d_dxlin_r[2], c->d_dxlin_r[3]};
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (int r = 0; r < rh_ctx::kRungs; r++) { if (c->d_lin_r[r]) (void)hipFree(c->d_lin_r[r]); if (c->d_wT_r[r]) (void)hipFree(c->d_wT_r[r]); }
    for (GraphSlot* g : {&c->g_in, &c->g_out, &c->g_dx}) if (g->exec) (void)hipGraphExecDestroy(g->exec);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->tev) if (e) (void)hipEventDestroy(e);
    if (c->s_mc) (void)hipStreamDestroy(c->s_mc);
    if (c->s_dx) (void)hipStreamDestroy(c->s_dx);
    for (int k = 0; k <= rh_ctx::kVRungs; k++) {
        if (c->d_vlin_m[k] && c->d_vlin_m[k] != c->d_vlin) (void)hipFree(c->d_vlin_m[k]);   // (the selected one went with `bufs`)
        if (c->h_vlin_m[k] && c->h_vlin_m[k] != c->h_vlin) delete c->h_vlin_m[k];
    }
    delete c->h_vlin;
    delete c->h_vienna;
    delete c->h_score;
    delete[] c->h_lin_r;
    delete c;
}

const char* rh_last_error(const rh_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int rh_bpp(rh_ctx* c, const char* seq, int n, const char* constraint, double* bp_tri, double* logZ)
{
    if (!c) return RH_ERR_ARG;
    if (constraint && c->model != RH_MODEL_VIENNA_BL)
        return fail(c, RH_ERR_UNSUPPORTED, "structure constraints apply to the Vienna-BL model only (RactIP::contrafold takes none)");
    if (!seq || n < 0) return fail(c, RH_ERR_ARG, "bad sequence");
    if (n == 0) { if (bp_tri) bp_tri[0] = 0.0; if (logZ) *logZ = 0.0; return RH_OK; }
    int rc;
    if ((rc = stage(c, 1, &seq, &n, true, false, constraint ? &constraint : nullptr))) return rc;
    if ((rc = compute(c))) return rc;
    if (bp_tri && (rc = fetch_bp(c, 0, bp_tri))) return rc;
    if (logZ && (rc = fetch_logz(c, 0, logZ))) return rc;
    return RH_OK;
}

int rh_unpaired(rh_ctx* c, const char* seq, int n, int max_w, double* up)
{
    if (!c) return RH_ERR_ARG;
    if (!seq || n < 0 || !up) return fail(c, RH_ERR_ARG, "bad argument");
    if (n == 0) return RH_OK;
    int rc;
    if ((rc = rh_set_max_w(c, max_w))) return rc;
    if ((rc = stage(c, 1, &seq, &n, true, false))) return rc;
    if ((rc = compute(c))) return rc;
    return fetch_up(c, 0, up);
}

int rh_fold(rh_ctx* c, const char* seq, int n, double* bp_tri, double* up, double* logZ)
{
    if (!c) return RH_ERR_ARG;
    if (!seq || n < 0) return fail(c, RH_ERR_ARG, "bad sequence");
    if (n == 0) { if (bp_tri) bp_tri[0] = 0.0; if (logZ) *logZ = 0.0; return RH_OK; }
    int rc;
    if ((rc = stage(c, 1, &seq, &n, true, false))) return rc;
    if ((rc = compute(c))) return rc;
    if (bp_tri && (rc = fetch_bp(c, 0, bp_tri))) return rc;
    if (up && (rc = fetch_up(c, 0, up))) return rc;
    if (logZ && (rc = fetch_logz(c, 0, logZ))) return rc;
    return RH_OK;
}

int rh_fold_constrained(rh_ctx* c, const char* seq, int n, const char* constraint, double* bp_tri, double* up, double* logZ)
{
    if (!c) return RH_ERR_ARG;
    if (constraint && c->model != RH_MODEL_VIENNA_BL)
        return fail(c, RH_ERR_UNSUPPORTED, "structure constraints apply to the Vienna-BL model only (RactIP::contrafold takes none)");
    if (!seq || n < 0) return fail(c, RH_ERR_ARG, "bad sequence");
    if (n == 0) { if (bp_tri) bp_tri[0] = 0.0; if (logZ) *logZ = 0.0; return RH_OK; }
    int rc;
    if ((rc = stage(c, 1, &seq, &n, true, false, constraint ? &constraint : nullptr))) return rc;
    if ((rc = compute(c))) return rc;
    if (bp_tri && (rc = fetch_bp(c, 0, bp_tri))) return rc;
    if (up && (rc = fetch_up(c, 0, up))) return rc;
    if (logZ && (rc = fetch_logz(c, 0, logZ))) return rc;
    return RH_OK;
}

int rh_duplex(rh_ctx* c, const char* s1, int n1, const char* s2, int n2, double* hp, double* logZ)
{
    if (!c) return RH_ERR_ARG;
    if (!s1 || !s2 || n1 < 1 || n2 < 1) return fail(c, RH_ERR_ARG, "bad sequence");
    const char* seqs[2] = {s1, s2};
    const int lens[2] = {n1, n2};
    int rc;
    if ((rc = stage(c, 2, seqs, lens, false, true))) return rc;
    if ((rc = compute(c))) return rc;
    return fetch_hp(c, 0, hp, logZ);
}

int rh_cofold_constrained(rh_ctx* c, const char* s1, int n1, const char* s2, int n2, const char* constraint, double* hp, double* logZ)
{
    if (!c) return RH_ERR_ARG;
    if (c->model != RH_MODEL_VIENNA_BL) return fail(c, RH_ERR_UNSUPPORTED, "the two-molecule ensemble needs RH_MODEL_VIENNA_BL");
    if (!s1 || !s2 || n1 < 1 || n2 < 1) return fail(c, RH_ERR_ARG, "bad sequence");
    const char* seqs[2] = {s1, s2};
    const int lens[2] = {n1, n2};
    const int keep = c->hybrid;
    c->hybrid = RH_HYBRID_COFOLD;
    int rc = stage(c, 2, seqs, lens, false, true, nullptr, constraint ? &constraint : nullptr);
    if (!rc) rc = compute(c);
    if (!rc) rc = fetch_hp(c, 0, hp, logZ);
    c->hybrid = keep;
    c->ns = 0; c->computed = false;   // the staged batch belongs to the other hybridization mode
    return rc;
}

int rh_batch_upload(rh_ctx* c, int npairs, const char* const* s1, const int* n1, const char* const* s2, const int* n2)
{
    if (!c) return RH_ERR_ARG;
    if (npairs < 1 || !s1 || !s2 || !n1 || !n2) return fail(c, RH_ERR_ARG, "bad batch");
    std::vector<const char*> seqs(2 * (size_t)npairs);
    std::vector<int> lens(2 * (size_t)npairs);
    for (int p = 0; p < npairs; p++) {
        seqs[2 * p] = s1[p]; seqs[2 * p + 1] = s2[p];
        lens[2 * p] = n1[p]; lens[2 * p + 1] = n2[p];
    }
    return stage(c, 2 * npairs, seqs.data(), lens.data(), true, true);
}

int rh_batch_compute(rh_ctx* c)
{
    if (!c) return RH_ERR_ARG;
    if (c->ns == 0) return fail(c, RH_ERR_ARG, "no batch uploaded");
    return compute(c);
}

int rh_batch_results(rh_ctx* c, int p, double* bp1, double* bp2, double* up1, double* up2, double* hp, double* logZ3)
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed || !c->has_mc || !c->has_dx) return fail(c, RH_ERR_ARG, "no computed pair batch");
    if (p < 0 || p >= c->np) return fail(c, RH_ERR_ARG, "pair %d out of range", p);
    int rc;
    if (bp1 && (rc = fetch_bp(c, 2 * p, bp1))) return rc;
    if (bp2 && (rc = fetch_bp(c, 2 * p + 1, bp2))) return rc;
    if (up1 && (rc = fetch_up(c, 2 * p, up1))) return rc;
    if (up2 && (rc = fetch_up(c, 2 * p + 1, up2))) return rc;
    if ((hp || logZ3) && (rc = fetch_hp(c, p, hp, logZ3 ? logZ3 + 2 : nullptr))) return rc;
    if (logZ3) {
        if ((rc = fetch_logz(c, 2 * p, logZ3))) return rc;
        if ((rc = fetch_logz(c, 2 * p + 1, logZ3 + 1))) return rc;
    }
    return RH_OK;
}

int rh_batch_logz(rh_ctx* c, double* out)
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed || !c->has_mc || !c->has_dx || !out) return fail(c, RH_ERR_ARG, "no computed pair batch");
    int rc;
    if ((rc = ensure(c, &c->d_scal, &c->cap_scal, sizeof(double) * 3 * c->np, false))) return rc;
    hipLaunchKernelGGL(collect_logz, dim3((c->np + 63) / 64), dim3(64), 0, c->s_mc, (const double*)c->d_mclogz, c->dx, (double*)c->d_scal);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scal, sizeof(double) * 3 * c->np, hipMemcpyDeviceToHost, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    return RH_OK;
}

int rh_batch_candidates(rh_ctx* c, int p, int which, float threshold, rh_cand* out, int cap)
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed) return fail(c, RH_ERR_ARG, "no computed batch");
    if (p < 0 || p >= c->np || which < 0 || which > 4 || cap < 0 || (cap > 0 && !out)) return fail(c, RH_ERR_ARG, "bad pair/which/cap");
    HIP_TRY(c, hipSetDevice(c->device));
    CandView v{};
    int nrows;
    if (which <= 1) {
        const int sq = 2 * p + which;
        v = CandView{(const double*)c->d_bp + (size_t)sq * c->mc.tri_stride, 0, c->n[sq], 0, 0};
        nrows = c->n[sq];
    } else if (which == 2) {
        v = CandView{(const double*)c->d_hp + (size_t)p * c->dx.tab_stride, 1, c->n[2 * p], c->n[2 * p + 1], c->dx.ldd};
        nrows = c->n[2 * p];
    } else {
        const int sq = 2 * p + (which - 3);
        v = CandView{(const double*)c->d_up + (size_t)sq * c->mc.ld * c->max_w, 2, c->n[sq], c->max_w, 0};
        nrows = 1;
    }
    int rc;
    if ((rc = ensure(c, &c->d_cnt, &c->cap_cnt, sizeof(int) * 2 * (size_t)(nrows + 1), false))) return rc;
    int* d_counts = (int*)c->d_cnt;
    int* d_offsets = d_counts + (nrows + 1);
    hipLaunchKernelGGL(cand_count, dim3((nrows + 3) / 4), dim3(256), 0, c->s_mc, v.base, v.kind, v.n, v.n2, v.ld, threshold, nrows, d_counts);
    std::vector<int> counts(nrows), offsets(nrows);
    HIP_TRY(c, hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * nrows, hipMemcpyDeviceToHost, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    int found = 0;
    for (int r = 0; r < nrows; r++) { offsets[r] = found; found += counts[r]; }
    const int take = std::min(found, cap);
    if (take > 0) {
        if ((rc = ensure(c, &c->d_cand, &c->cap_cand, sizeof(rh_cand) * (size_t)take, false))) return rc;
        HIP_TRY(c, hipMemcpyAsync(d_offsets, offsets.data(), sizeof(int) * nrows, hipMemcpyHostToDevice, c->s_mc));
        hipLaunchKernelGGL(cand_write, dim3((nrows + 3) / 4), dim3(256), 0, c->s_mc, v.base, v.kind, v.n, v.n2, v.ld, threshold,
                           nrows, d_offsets, (rh_cand*)c->d_cand, take);
        HIP_TRY(c, hipMemcpyAsync(out, c->d_cand, sizeof(rh_cand) * (size_t)take, hipMemcpyDeviceToHost, c->s_mc));
        HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    }
    return found;
}

int rh_batch_candidates_all(rh_ctx* c, int which, float threshold, rh_cand* out, int cap, int* first)
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed || !c->has_mc || !c->has_dx) return fail(c, RH_ERR_ARG, "no computed pair batch");
    if (which < 0 || which > 4 || cap < 0 || (cap > 0 && !out) || !first) return fail(c, RH_ERR_ARG, "bad which/cap/first");
    HIP_TRY(c, hipSetDevice(c->device));
    const int np = c->np;
    const int rmax = which >= 3 ? 1 : (which == 2 ? c->dx.n1max : c->mc.nmax);
    const size_t nrows = (size_t)np * rmax;
    int rc;
    if ((rc = ensure(c, &c->d_cnt, &c->cap_cnt, sizeof(int) * 2 * (nrows + 1), false))) return rc;
    int* d_counts = (int*)c->d_cnt;
    int* d_offsets = d_counts + (nrows + 1);
    const dim3 grid((rmax + 3) / 4, np);
    hipLaunchKernelGGL(cand_count_all, grid, dim3(256), 0, c->s_mc, (const double*)c->d_bp, (const double*)c->d_hp, (const double*)c->d_up,
                       (const int*)c->d_n, c->mc.tri_stride, c->dx.tab_stride, c->mc.ld * c->max_w, which >= 3 ? c->max_w : c->dx.ldd, which, rmax, threshold, d_counts);
    std::vector<int> counts(nrows), offsets(nrows);
    HIP_TRY(c, hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * nrows, hipMemcpyDeviceToHost, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    int found = 0;
    for (int p = 0; p < np; p++) {
        first[p] = found;
        for (int r = 0; r < rmax; r++) { offsets[(size_t)p * rmax + r] = found; found += counts[(size_t)p * rmax + r]; }
    }
    first[np] = found;
    const int take = std::min(found, cap);
    if (take > 0) {
        if ((rc = ensure(c, &c->d_cand, &c->cap_cand, sizeof(rh_cand) * (size_t)take, false))) return rc;
        HIP_TRY(c, hipMemcpyAsync(d_offsets, offsets.data(), sizeof(int) * nrows, hipMemcpyHostToDevice, c->s_mc));
        hipLaunchKernelGGL(cand_write_all, grid, dim3(256), 0, c->s_mc, (const double*)c->d_bp, (const double*)c->d_hp, (const double*)c->d_up,
                           (const int*)c->d_n, c->mc.tri_stride, c->dx.tab_stride, c->mc.ld * c->max_w, which >= 3 ? c->max_w : c->dx.ldd, which, rmax, threshold, d_offsets,
                           (rh_cand*)c->d_cand, take);
        HIP_TRY(c, hipMemcpyAsync(out, c->d_cand, sizeof(rh_cand) * (size_t)take, hipMemcpyDeviceToHost, c->s_mc));
        HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    }
    return found;
}

int rh_batch_layout(rh_ctx* c, size_t* tri_stride, int* up_ld, size_t* hp_stride, int* hp_ld)
{
    if (!c) return RH_ERR_ARG;
    if (!c->has_mc || !c->has_dx) return fail(c, RH_ERR_ARG, "no pair batch");
    if (tri_stride) *tri_stride = c->mc.tri_stride;
    if (up_ld) *up_ld = c->mc.ld * c->max_w;
    if (hp_stride) *hp_stride = c->dx.tab_stride;
    if (hp_ld) *hp_ld = c->dx.ldd;
    return RH_OK;
}

int rh_batch_results_all(rh_ctx* c, double* bp, double* up, double* hp, double* logz)
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed || !c->has_mc || !c->has_dx) return fail(c, RH_ERR_ARG, "no computed pair batch");
    HIP_TRY(c, hipSetDevice(c->device));
    if (bp) HIP_TRY(c, hipMemcpyAsync(bp, c->d_bp, sizeof(double) * c->mc.tri_stride * c->ns, hipMemcpyDeviceToHost, c->s_mc));
    if (up) HIP_TRY(c, hipMemcpyAsync(up, c->d_up, sizeof(double) * c->mc.ld * c->max_w * c->ns, hipMemcpyDeviceToHost, c->s_mc));
    if (hp) HIP_TRY(c, hipMemcpyAsync(hp, c->d_hp, sizeof(double) * c->dx.tab_stride * c->np, hipMemcpyDeviceToHost, c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    if (logz) return rh_batch_logz(c, logz);
    return RH_OK;
}

void* rh_host_alloc(rh_ctx* c, size_t bytes)
{
    if (!c || bytes == 0) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        fail(c, RH_ERR_OOM, "hipHostMalloc of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}

void rh_host_free(rh_ctx* c, void* p)
{
    if (c && p) { (void)hipSetDevice(c->device); (void)hipHostFree(p); }
}

int rh_set_max_w(rh_ctx* c, int max_w)
{
    if (!c) return RH_ERR_ARG;
    if (max_w < 1 || max_w > 64) return fail(c, RH_ERR_ARG, "max_w=%d out of range [1,64]", max_w);
    if (c->model == RH_MODEL_CONTRAFOLD && max_w != 1)
        return fail(c, RH_ERR_UNSUPPORTED, "max_w=%d: the CONTRAfold path has width-1 accessibility only (src/ractip.cpp:213-222)", max_w);
    if (max_w != c->max_w) { c->max_w = max_w; c->computed = false; c->ns = 0; }   // buffers are sized at upload
    return RH_OK;
}
int rh_get_max_w(const rh_ctx* c) { return c ? c->max_w : RH_ERR_ARG; }

int rh_set_hybrid(rh_ctx* c, int hybrid)
{
    if (!c) return RH_ERR_ARG;
    if (hybrid != RH_HYBRID_DUPLEX && hybrid != RH_HYBRID_COFOLD) return fail(c, RH_ERR_ARG, "unknown hybridization mode %d", hybrid);
    if (hybrid == RH_HYBRID_COFOLD && c->model != RH_MODEL_VIENNA_BL)
        return fail(c, RH_ERR_UNSUPPORTED, "the two-molecule (co_pf_fold) hybridization matrix needs RH_MODEL_VIENNA_BL");
    if (hybrid != c->hybrid) { c->hybrid = hybrid; c->computed = false; c->ns = 0; }
    return RH_OK;
}

int rh_set_mode(rh_ctx* c, int mode)
{
    if (!c) return RH_ERR_ARG;
    if (mode < RH_MODE_AUTO || mode > RH_MODE_LINEAR) return fail(c, RH_ERR_ARG, "unknown mode %d", mode);
    if (c->vienna_sem == kViennaSem20) {   // ViennaRNA-2.x semantics: log-space kernels only (AUTO means LOG)
        if (mode == RH_MODE_LINEAR) return fail(c, RH_ERR_UNSUPPORTED, "RH_VIENNA_SEM_20 runs on the log-space kernels only");
        c->mode = RH_MODE_LOG;
        return RH_OK;
    }
    c->mode = mode;
    return RH_OK;
}

int rh_last_path(const rh_ctx* c) { return c ? c->last_path : RH_ERR_ARG; }
int rh_last_hybrid_path(const rh_ctx* c) { return c ? c->last_dx_path : RH_ERR_ARG; }

int rh_batch_kernels(rh_ctx* c, const char* fine[3], const char* far[3], int n_far[3])
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed) return fail(c, RH_ERR_ARG, "no computed batch");
    static thread_local std::string names[6];
    const bool vienna0 = c->model == RH_MODEL_VIENNA_BL;
    const std::string w = std::to_string(c->lin_w == 16 ? 16 : (!vienna0 && c->lin_w == 4 && c->lin_bs != 0 && c->lin_bs != 32 ? 4 : 8)), w_in = std::to_string(c->lin_w_in == 16 ? 16 : (c->lin_w_in == 4 && c->lin_bs != 0 && c->lin_bs != 32 ? 4 : 8)), bs = std::to_string(c->lin_bs == 0 || c->lin_bs == 32 ? c->lin_bs : 16);
    const bool vienna = c->model == RH_MODEL_VIENNA_BL, lin = c->last_path == 1;
    const bool pairs = !vienna && lin && c->lookahead == 2 && c->far_mfma && c->lin_bs == 16;   // two diagonals per launch
    const std::string pre = vienna ? "vlin_" : "lin_";
    const std::string targs = vienna ? bs + ", false" : bs;
    names[0] = !c->has_mc ? "" : lin ? pre + "inside_diag<" + (vienna ? "8" : w_in) + ", " + targs + (vienna ? "" : (pairs && c->lin_w_in == 4) ? ", 3" : (c->lookahead == 1 && c->lin_w_in == 4 && c->lin_bs == 16) ? ", 1" : ", 0") + ">"
                                     : vienna ? "mcv_inside_diag" : "mc_inside_diag";
    if (c->has_mc && lin && !vienna && strip_inside(c, c->mc)) names[0] = c->strip_w == 4 ? "lin_inside_strip<8, 4, 0>" : ((c->strip_filt && c->strip_filt_ok) ? "lin_inside_strip<8, 8, 1>" : "lin_inside_strip<8, 8, 0>");
    const bool ostrip = c->has_mc && lin && !vienna && strip_outside(c, c->mc);
    names[1] = ostrip ? (c->strip_w == 4 ? "lin_outside_strip<8, 4, 0>" : ((c->strip_filt && c->strip_filt_ok) ? "lin_outside_strip<8, 8, 1>" : "lin_outside_strip<8, 8, 0>")) : !c->has_mc ? "" : lin ? ((pairs && (c->lin_w == 4 || c->lin_w == 8)) ? "lin_outside_pair<" + w + ", " + targs + ">" : pre + "outside_diag<" + (vienna ? "8" : w) + ", " + targs + ">")
                                     : vienna ? "mcv_outside_diag" : "mc_outside_diag";
    names[2] = !c->has_dx ? "" : vienna ? (c->hybrid == RH_HYBRID_COFOLD ? (c->last_dx_path == 1 ? "vlin_inside_diag<8, 16, true> + vlin_outside_diag<8, 16, true> (s1+s2)"
                                                                                         : "mcv_inside_diag + mcv_outside_diag (s1+s2)") : (c->last_dx_path == 1 ? "dxvl_sweep4" : "dxv_sweep_diag")) : c->last_dx_path == 1 ? ((c->dx_quad && c->dx_w != 2 && c->dx_w != 8) ? std::string(c->dx_strip ? "dxl_strip8" : "dxl_sweep4") : "dxl_sweep<" + std::to_string(c->dx_w == 2 || c->dx_w == 8 ? c->dx_w : 4) + ">") : "dx_sweep_diag";
    const bool mfma = c->far_mfma && c->lin_bs != 0 && c->lin_bs != 32;
    const std::string fsuf = c->far_pk ? "_pk" : "_mfma";
    names[3] = (c->has_mc && lin && c->n_far[0]) ? (mfma ? "lin_far_inside" + fsuf : "lin_far_inside<" + bs + ">") : "";
    names[4] = (c->has_mc && lin && c->n_far[1]) ? (mfma ? "lin_far_outside" + fsuf : "lin_far_outside<" + bs + ">") : "";
    names[5] = "";
    for (int k = 0; k < 3; k++) {
        if (fine) fine[k] = names[k].c_str();
        if (far) far[k] = names[3 + k].c_str();
        if (n_far) n_far[k] = c->n_far[k];
    }
    return RH_OK;
}

int rh_batch_fallbacks(rh_ctx* c, int which, int* out, int cap)
{
    if (!c || which < 0 || which > 3) return RH_ERR_ARG;
    if (!c->computed) return fail(c, RH_ERR_ARG, "no computed batch");
    const std::vector<int>& F = which == 3 ? c->rescaled_dx : (which == 2 ? c->rescaled_mc : (which ? c->fallback_dx : c->fallback_mc));
    if (cap > 0 && !out) return fail(c, RH_ERR_ARG, "rh_batch_fallbacks: out is NULL with cap > 0");
    for (int k = 0; k < (int)F.size() && k < cap; k++) out[k] = F[k];
    return (int)F.size();
}

int rh_set_scale_memory(rh_ctx* c, int on)
{
    if (!c) return RH_ERR_ARG;
    c->scale_memory = on != 0;
    if (!on) { c->lin_primary = -1; c->vlin_primary = -1; }
    return RH_OK;
}

int rh_set_kernel_timing(rh_ctx* c, int cls)
{
    if (!c || cls < -1 || cls > 4) return RH_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (cls >= 0 && c->tev.empty()) {
        c->tev.resize(8192);
        for (auto& e : c->tev) HIP_TRY(c, hipEventCreate(&e));
    }
    c->time_cls = cls;
    c->tev_n = 0;
    return RH_OK;
}

int rh_kernel_times(rh_ctx* c, int* n_launches, double* total_ms)
{
    if (!c || !n_launches || !total_ms) return RH_ERR_ARG;
    if (!c->computed) return fail(c, RH_ERR_ARG, "no computed batch");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->s_mc));
    HIP_TRY(c, hipStreamSynchronize(c->s_dx));
    double tot = 0.0;
    for (size_t k = 0; k + 1 < c->tev_n; k += 2) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->tev[k], c->tev[k + 1]));
        tot += ms;
    }
    *n_launches = (int)(c->tev_n / 2);
    *total_ms = tot;
    return RH_OK;
}

int rh_set_overlap(rh_ctx* c, int on)
{
    if (!c) return RH_ERR_ARG;
    c->overlap = on != 0;
    return RH_OK;
}

int rh_batch_timings(rh_ctx* c, double ms[4], int n_launch[3])
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed) return fail(c, RH_ERR_ARG, "no computed batch");
    if (ms) for (int k = 0; k < 4; k++) ms[k] = c->ms[k];
    if (n_launch) for (int k = 0; k < 3; k++) n_launch[k] = c->n_launch[k];
    return RH_OK;
}

int rh_batch_device_views(rh_ctx* c, const double** bp, size_t* tri_stride, const double** hp, size_t* hp_stride, int* hp_ld)
{
    if (!c) return RH_ERR_ARG;
    if (!c->computed) return fail(c, RH_ERR_ARG, "no computed batch");
    if (bp) *bp = (const double*)c->d_bp;
    if (tri_stride) *tri_stride = c->mc.tri_stride;
    if (hp) *hp = (const double*)c->d_hp;
    if (hp_stride) *hp_stride = c->dx.tab_stride;
    if (hp_ld) *hp_ld = c->dx.ldd;
    return RH_OK;
}

}  // extern "C"

// (file-scope alias of create_ctx for the per-pair helper context; C++ linkage, not part of the C ABI)
rh_ctx* make_ctx_for_helper(int device, int model, const char* param_file, const char* defaults_file, int use_bl, int semantics)
{
    return create_ctx(device, model, param_file, defaults_file, use_bl, semantics);
}
