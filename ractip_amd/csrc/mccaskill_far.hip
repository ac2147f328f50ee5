// mccaskill_far.hip -- block products for the O(n^3) terms of the linear-space McCaskill sweeps.
//
// The multibranch terms (reference: /root/reference/src/contrafold/InferenceEngine.ipp:3384-3411 inside,
// 4046-4064 outside) are triangular matrix products.  With positions cut into blocks of BS:
//   FM2 [i,j] = sum_k  FM1[i,k]  * FM[k,j]      k in blocks  I+2 .. J-2   ("far"),  rest "near"
//   FMo [i,j] = sum_i' FM1[i',i] * FM2o[i',j]   i' in blocks 0   .. I-2
//   FM1o[i,j] = sum_j' FM2o[i,j']* FM[j,j']     j' in blocks J+2 .. last
// Every operand block of a far term is COMPLETE (all its cells final) at least one fine diagonal before
// the first cell of tile (I,J) is due, so the far part of a whole tile is one dense BSxBS product per
// operand block pair: each operand element is read once per tile instead of once per cell (HBM traffic
// / BS), staged through LDS.  The per-diagonal kernels (mccaskill_lin.hip) add the <= 4*BS near terms.
// In scaled linear space the products need no exponent handling: (k-i)+(j-k) = j-i.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lin_model.h"

namespace rh {

enum LinTableFar { LF_FM1 = 3, LF_FM = 4, LF_FM2O = 7, LF_FM2F = 10, LF_FMOF = 11, LF_FM1OF = 12 };

namespace {

// cell (x,y) of a diagonal-major table, zero outside the interior 1 <= x <= y <= n-1
__device__ __forceinline__ double cellv(const double* __restrict__ T, int ld, int n, int x, int y)
{
    return (x >= 1 && x <= y && y <= n - 1) ? T[(size_t)(y - x) * ld + x] : 0.0;
}

// Load the BSxBS chunk {cell(T, x0+u, y0+v)} into LDS as dst[u][v] (TR = false) or dst[v][u] (TR = true).
// Threads walk the chunk along its diagonals v-u = const: those are contiguous runs of the table.
template <int BS, bool TR>
__device__ __forceinline__ void load_chunk(double (*dst)[BS + 1], const double* __restrict__ T, int ld, int n, int x0, int y0)
{
    const int u = threadIdx.x % BS;
    constexpr int DPP = 256 / BS;  // diagonals per pass
    for (int dd = threadIdx.x / BS; dd < 2 * BS - 1; dd += DPP) {
        const int v = u + dd - (BS - 1);
        if (v >= 0 && v < BS) {
            const double val = cellv(T, ld, n, x0 + u, y0 + v);
            if (TR) dst[v][u] = val; else dst[u][v] = val;
        }
    }
}

// C[r][c] += sum_kk A[r][kk] * Bm[kk][c] for this thread's outputs (1 for BS=16, 2x2 for BS=32)
template <int BS>
struct Acc {
    static constexpr int R = BS / 16;
    double c[R][R];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int a = 0; a < R; a++)
#pragma unroll
            for (int b = 0; b < R; b++) c[a][b] = 0.0;
    }
    __device__ __forceinline__ void mac(const double (*A)[BS + 1], const double (*Bm)[BS + 1])
    {
        const int r0 = (threadIdx.x / 16) * R, c0 = (threadIdx.x % 16) * R;
#pragma unroll 8
        for (int kk = 0; kk < BS; kk++) {
            double a[R], b[R];
#pragma unroll
            for (int q = 0; q < R; q++) { a[q] = A[r0 + q][kk]; b[q] = Bm[kk][c0 + q]; }
#pragma unroll
            for (int p = 0; p < R; p++)
#pragma unroll
                for (int q = 0; q < R; q++) c[p][q] = fma(a[p], b[q], c[p][q]);
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ T, int ld, int n, int i0, int j0) const
    {
        const int r0 = (threadIdx.x / 16) * R, c0 = (threadIdx.x % 16) * R;
#pragma unroll
        for (int p = 0; p < R; p++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = i0 + r0 + p, j = j0 + c0 + q;
                if (i >= 1 && i <= j && j <= n - 1) T[(size_t)(j - i) * ld + i] = c[p][q];
            }
    }
};

}  // namespace

// inside: FM2F[tile (I, I+D)] = sum_{K=I+2}^{J-2} FM1(I,K) x FM(K,J);  grid = (tiles I, sequences)
template <int BS>
__global__ __launch_bounds__(256) void lin_far_inside(McBatch B, int D)
{
    __shared__ double A[BS][BS + 1], Bm[BS][BS + 1];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int I = blockIdx.x, J = I + D;
    if (J * BS > n - 1) return;  // no interior column in this tile
    const int ld = B.ld;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ fm1 = tab + (size_t)LF_FM1 * B.tab_stride;
    const double* __restrict__ fm = tab + (size_t)LF_FM * B.tab_stride;
    Acc<BS> acc;
    acc.zero();
    for (int K = I + 2; K <= J - 2; K++) {
        load_chunk<BS, false>(A, fm1, ld, n, I * BS, K * BS);   // A[r][kk]  = FM1[i0+r, k0+kk]
        load_chunk<BS, false>(Bm, fm, ld, n, K * BS, J * BS);   // Bm[kk][c] = FM [k0+kk, j0+c]
        __syncthreads();
        acc.mac(A, Bm);
        __syncthreads();
    }
    acc.store(tab + (size_t)LF_FM2F * B.tab_stride, ld, n, I * BS, J * BS);
}

// outside: FMOF[tile]  = sum_{K<=I-2} FM1(K,I)^T x FM2o(K,J)
//          FM1OF[tile] = sum_{K>=J+2} FM2o(I,K)   x FM(J,K)^T ;  grid = (tiles I, sequences, 2)
template <int BS>
__global__ __launch_bounds__(256) void lin_far_outside(McBatch B, int D)
{
    __shared__ double A[BS][BS + 1], Bm[BS][BS + 1];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int I = blockIdx.x, J = I + D;
    if (J * BS > n - 1) return;
    const int ld = B.ld;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ fm1 = tab + (size_t)LF_FM1 * B.tab_stride;
    const double* __restrict__ fm = tab + (size_t)LF_FM * B.tab_stride;
    const double* __restrict__ fm2o = tab + (size_t)LF_FM2O * B.tab_stride;
    Acc<BS> acc;
    acc.zero();
    if (blockIdx.z == 0) {
        for (int K = 0; K <= I - 2; K++) {
            load_chunk<BS, true>(A, fm1, ld, n, K * BS, I * BS);     // A[r][kk]  = FM1 [k0+kk, i0+r]
            load_chunk<BS, false>(Bm, fm2o, ld, n, K * BS, J * BS);  // Bm[kk][c] = FM2o[k0+kk, j0+c]
            __syncthreads();
            acc.mac(A, Bm);
            __syncthreads();
        }
        acc.store(tab + (size_t)LF_FMOF * B.tab_stride, ld, n, I * BS, J * BS);
    } else {
        const int last = (n - 1) / BS;
        for (int K = J + 2; K <= last; K++) {
            load_chunk<BS, false>(A, fm2o, ld, n, I * BS, K * BS);   // A[r][kk]  = FM2o[i0+r, k0+kk]
            load_chunk<BS, true>(Bm, fm, ld, n, J * BS, K * BS);     // Bm[kk][c] = FM  [j0+c, k0+kk]
            __syncthreads();
            acc.mac(A, Bm);
            __syncthreads();
        }
        acc.store(tab + (size_t)LF_FM1OF * B.tab_stride, ld, n, I * BS, J * BS);
    }
}

// ---------------------------------------------------------------------------------
// f64-MFMA variant for BS = 16 (the dense-contraction recast of the multiloop term): one wavefront owns
// one 16x16x16 operand-chunk product = 4 x v_mfma_f64_16x16x4_f64; the 4 wavefronts of a workgroup take
// every 4th operand block of the tile (split-K), stage their chunks in private LDS (no workgroup barrier in
// the K loop) and meet once at the end.  Lane layout of the instruction (cdna_hip_programming.md section 3):
// A[l&15][l>>4], B[l>>4][l&15], D reg r -> row (l>>4)+4r, col l&15.
typedef double d4 __attribute__((ext_vector_type(4)));

template <bool TR>
__device__ __forceinline__ void load_chunk16_wave(double (*dst)[17], const double* __restrict__ T, int ld, int n, int x0, int y0)
{
    const int lane = threadIdx.x & 63;
    const int u = lane & 15;
#pragma unroll
    for (int pass = 0; pass < 8; pass++) {
        const int v = u + (lane >> 4) + 4 * pass - 15;
        if (v >= 0 && v < 16) {
            const double val = cellv(T, ld, n, x0 + u, y0 + v);
            if (TR) dst[v][u] = val; else dst[u][v] = val;
        }
    }
}

__device__ __forceinline__ d4 mfma_chunk(const double (*A)[17], const double (*Bm)[17], d4 acc)
{
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r][4 * s + q], Bm[4 * s + q][r], acc, 0, 0, 0);
    return acc;
}

// sum the 4 wavefronts' accumulators and store the tile (wave 0)
__device__ __forceinline__ void reduce_store(double (*red)[256], d4 acc, double* __restrict__ T, int ld, int n, int i0, int j0, bool rmw = false)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; r++) red[w][r * 64 + lane] = acc[r];
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double v = red[0][r * 64 + lane] + red[1][r * 64 + lane] + red[2][r * 64 + lane] + red[3][r * 64 + lane];
        const int i = i0 + (lane >> 4) + 4 * r, j = j0 + (lane & 15);
        if (i >= 1 && i <= j && j <= n - 1) { double* t = T + (size_t)(j - i) * ld + i; *t = rmw ? *t + v : v; }   // rmw: on top of the 64-block products
    }
}

__global__ __launch_bounds__(256) void lin_far_inside_mfma(McBatch B, int D)
{
    __shared__ double A[4][16][17], Bm[4][16][17], red[4][256];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int I = blockIdx.x, J = I + D;
    if (J * 16 > n - 1) return;
    const int w = threadIdx.x >> 6;
    const int ld = B.ld;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ fm1 = tab + (size_t)LF_FM1 * B.tab_stride;
    const double* __restrict__ fm = tab + (size_t)LF_FM * B.tab_stride;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int K = I + 2 + w; K <= J - 2; K += 4) {
        load_chunk16_wave<false>(A[w], fm1, ld, n, I * 16, K * 16);
        load_chunk16_wave<false>(Bm[w], fm, ld, n, K * 16, J * 16);
        acc = mfma_chunk(A[w], Bm[w], acc);
    }
    reduce_store(red, acc, tab + (size_t)LF_FM2F * B.tab_stride, ld, n, I * 16, J * 16);
}

__global__ __launch_bounds__(256) void lin_far_outside_mfma(McBatch B, int D)
{
    __shared__ double A[4][16][17], Bm[4][16][17], red[4][256];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int I = blockIdx.x, J = I + D;
    if (J * 16 > n - 1) return;
    const int w = threadIdx.x >> 6;
    const int ld = B.ld;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ fm1 = tab + (size_t)LF_FM1 * B.tab_stride;
    const double* __restrict__ fm = tab + (size_t)LF_FM * B.tab_stride;
    const double* __restrict__ fm2o = tab + (size_t)LF_FM2O * B.tab_stride;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    if (blockIdx.z == 0) {
        for (int K = w; K <= I - 2; K += 4) {
            load_chunk16_wave<true>(A[w], fm1, ld, n, K * 16, I * 16);     // A[r][kk]  = FM1 [k0+kk, i0+r]
            load_chunk16_wave<false>(Bm[w], fm2o, ld, n, K * 16, J * 16);  // Bm[kk][c] = FM2o[k0+kk, j0+c]
            acc = mfma_chunk(A[w], Bm[w], acc);
        }
        reduce_store(red, acc, tab + (size_t)LF_FMOF * B.tab_stride, ld, n, I * 16, J * 16);
    } else {
        const int last = (n - 1) / 16;
        for (int K = J + 2 + w; K <= last; K += 4) {
            load_chunk16_wave<false>(A[w], fm2o, ld, n, I * 16, K * 16);   // A[r][kk]  = FM2o[i0+r, k0+kk]
            load_chunk16_wave<true>(Bm[w], fm, ld, n, J * 16, K * 16);     // Bm[kk][c] = FM  [j0+c, k0+kk]
            acc = mfma_chunk(A[w], Bm[w], acc);
        }
        reduce_store(red, acc, tab + (size_t)LF_FM1OF * B.tab_stride, ld, n, I * 16, J * 16);
    }
}

// ---------------------------------------------------------------------------------
// Packed operand tiles.  Gathering a 16x16 chunk out of the diagonal-major tables costs 16 masked loads per wavefront
// with their full latency, and every chunk is gathered again by each of the ~n/32 products that use it.  Instead each
// tile is re-laid ONCE, when its block diagonal completes, into the register order of the MFMA operands:
//   layout LA: element (u,v) at (u + 16*(v%4))*4 + v/4  -> lane l reads 4 consecutive doubles = A[l&15][4s + (l>>4)], s = 0..3
//   layout LB: element (u,v) at (v + 16*(u%4))*4 + u/4  -> lane l reads B[4s + (l>>4)][l&15]
// (a transposed operand is the other layout of the same tile), so a product step is two 32-byte loads per lane from two
// contiguous 2 KB tiles and four v_mfma_f64_16x16x4_f64, with no LDS and every load of the K loop in flight at once.
enum PkCopy { PK_FM1_A = 0, PK_FM1_B, PK_FM_A, PK_FM_B, PK_FM2O_A, PK_FM2O_B };
__device__ __forceinline__ size_t pk_tile(int nb, int P, int Q) { return ((size_t)P * nb - (size_t)P * (P - 1) / 2 + (Q - P)) * 256; }

// grid = (tiles P of block diagonal Dblk, sequences, tables); outside = 0: FM1 and FM (blockIdx.z), outside = 1: FM2o
// banded != 0 (strip kernels, mccaskill_strip.hip): the far range of a cell is k in [i+32, j-32], not whole blocks; its two partial
// blocks are the tiles of block diagonal 2 with the entries (first index u, second index v), v < u, removed -- the same mask
// whichever product uses the tile (FM1 as the K = I+2 operand inside / the K = I-2 operand of FMOF, FM as K = J-2 / K = J+2),
// and such a tile is used by no other product, so it is simply packed masked.
// SWEEP only names the launch (0: issued by the inside sweep, 1: by the outside sweep), so that a kernel trace / counter pass attributes it
template <int SWEEP>
__global__ __launch_bounds__(256) void lin_pack_tiles(McBatch B, int Dblk, int outside, int banded)
{
    __shared__ double T[16][17];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int P = blockIdx.x, Q = P + Dblk;
    if (Q >= B.nb || Q * 16 > n - 1) return;   // no interior column: never read
    const int slot = outside ? LF_FM2O : (blockIdx.z == 0 ? LF_FM1 : LF_FM);
    const int copy = outside ? PK_FM2O_A : (blockIdx.z == 0 ? PK_FM1_A : PK_FM_A);
    const double* __restrict__ src = B.tab + (size_t)sq * B.seq_stride + (size_t)slot * B.tab_stride;
    {   // walk the chunk along its diagonals v-u = const (contiguous runs of the table)
        const int u = threadIdx.x & 15;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int v = u + (threadIdx.x >> 4) + 16 * pass - 15;
            if (v >= 0 && v < 16) T[u][v] = (banded && !outside && Dblk == 2 && v < u) ? 0.0 : cellv(src, B.ld, n, P * 16 + u, Q * 16 + v);
        }
    }
    __syncthreads();
    double* __restrict__ dst = B.pk + ((size_t)sq * kPkCopies + copy) * B.pk_stride + pk_tile(B.nb, P, Q);
    const int s = threadIdx.x & 3, l = threadIdx.x >> 2, r = l & 15, q = l >> 4;
    dst[threadIdx.x] = T[r][4 * s + q];                  // LA
    dst[B.pk_stride + threadIdx.x] = T[4 * s + q][r];    // LB (the next copy)
}

__device__ __forceinline__ d4 mfma4(d4 a, d4 b, d4 acc)
{
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// acc += sum over K = k_lo + w, k_lo + w + 4, ... <= k_hi of frag(pa + ta(K)) x frag(pb + tb(K)); two steps in flight
template <class TA, class TB>
__device__ __forceinline__ d4 pk_loop(const double* __restrict__ pa, const double* __restrict__ pb, int k_lo, int k_hi, TA ta, TB tb,
                                      d4 acc = d4{0.0, 0.0, 0.0, 0.0})
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int K = k_lo + w;
    for (; K + 4 <= k_hi; K += 8) {
        const d4 a0 = *(const d4*)(pa + ta(K) + lane * 4), b0 = *(const d4*)(pb + tb(K) + lane * 4);
        const d4 a1 = *(const d4*)(pa + ta(K + 4) + lane * 4), b1 = *(const d4*)(pb + tb(K + 4) + lane * 4);
        acc = mfma4(a0, b0, acc);
        acc = mfma4(a1, b1, acc);
    }
    if (K <= k_hi) {
        const d4 a0 = *(const d4*)(pa + ta(K) + lane * 4), b0 = *(const d4*)(pb + tb(K) + lane * 4);
        acc = mfma4(a0, b0, acc);
    }
    return acc;
}

// Two-level block products (l2 != 0, long sequences).  With 16x16 tiles every operand tile is read once per product: 4 KB per 8 kFLOP,
// which at n = 2000 is 250 GB per 32 pairs and the largest item of the step.  The bulk of every K range is therefore taken by
// 64x64 MACRO tiles (4x4 tiles, lin_far2_*): a macro tile (I2,J2) sums K2 = I2+2 .. J2-2 (in 64-blocks) with each wavefront
// owning a 32x32 quadrant -- 4 operand tiles feed 4 tile products, 16 KB per 64 tile products per workgroup, a quarter of the
// traffic -- and the 16-tile kernels add only the remaining <= 2 x 9 blocks next to the tile's row / column block ON TOP of it
// (read-modify-write).  All operand tiles of the macro range are complete before the first cell of the macro tile is due
// (spans <= 64(J2-I2)-65 < 64(J2-I2)-63), so far2(D2) is launched with the tile kernel of block diagonal 4*D2-3 (inside) /
// 4*D2+3 (outside), behind that launch's packing.

// two-molecule batch (B.cut): a target tile whose cells all lie on one strand (letters i..j+1 <= cut, or i > cut) is never read --
// outside: those spans do not enter the joint pair matrix; inside: only when the one-strand cells were copied from the
// single-molecule folds (B.seeded).  lo_i = first start position, hi_j = last end position of the tile.
__device__ __forceinline__ bool one_strand_tile(const McBatch& B, int sq, bool inside, int lo_i, int hi_j)
{
    if (!B.cut || (inside && !B.seeded)) return false;
    const int cut = B.cut[sq];
    return cut > 0 && (hi_j + 1 <= cut || lo_i > cut);
}

__global__ __launch_bounds__(256) void lin_far_inside_pk(McBatch B, int D, int l2)
{
    __shared__ double red[4][256];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int I = blockIdx.x, J = I + D;
    if (J * 16 > n - 1) return;
    if (one_strand_tile(B, sq, true, I * 16, J * 16 + 15)) return;
    const int nb = B.nb;
    const double* __restrict__ pk = B.pk + (size_t)sq * kPkCopies * B.pk_stride;
    // FM2F(I,J) = sum_K FM1(I,K) x FM(K,J)
    const auto ta = [=](int K) { return pk_tile(nb, I, K); };
    const auto tb = [=](int K) { return pk_tile(nb, K, J); };
    const int I2 = I >> 2, J2 = J >> 2;
    const bool two = l2 > 0 && n >= l2 && J2 - I2 >= 4;   // K = 4(I2+2) .. 4(J2-1)-1 came from the macro tile (l2 = shortest two-level sequence, 0 = none)
    d4 acc = pk_loop(pk + PK_FM1_A * B.pk_stride, pk + PK_FM_B * B.pk_stride, I + 2, two ? 4 * (I2 + 2) - 1 : J - 2, ta, tb);
    if (two) acc = pk_loop(pk + PK_FM1_A * B.pk_stride, pk + PK_FM_B * B.pk_stride, 4 * (J2 - 1), J - 2, ta, tb, acc);
    reduce_store(red, acc, B.tab + (size_t)sq * B.seq_stride + (size_t)LF_FM2F * B.tab_stride, B.ld, n, I * 16, J * 16, two);
}

__global__ __launch_bounds__(256) void lin_far_outside_pk(McBatch B, int D, int l2)
{
    __shared__ double red[4][256];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int I = blockIdx.x, J = I + D;
    if (J * 16 > n - 1) return;
    if (one_strand_tile(B, sq, false, I * 16, J * 16 + 15)) return;
    const int nb = B.nb;
    const double* __restrict__ pk = B.pk + (size_t)sq * kPkCopies * B.pk_stride;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const int I2 = I >> 2, J2 = J >> 2, last = (n - 1) / 16;
    if (blockIdx.z == 0) {   // FMOF(I,J) = sum_{K<=I-2} FM1(K,I)^T x FM2o(K,J); K <= 4(I2-1)-1 came from the macro tile
        const bool two = l2 > 0 && n >= l2 && I2 >= 2;
        const d4 acc = pk_loop(pk + PK_FM1_B * B.pk_stride, pk + PK_FM2O_B * B.pk_stride, two ? 4 * (I2 - 1) : 0, I - 2,
                               [=](int K) { return pk_tile(nb, K, I); }, [=](int K) { return pk_tile(nb, K, J); });
        reduce_store(red, acc, tab + (size_t)LF_FMOF * B.tab_stride, B.ld, n, I * 16, J * 16, two);
    } else {                 // FM1OF(I,J) = sum_{K>=J+2} FM2o(I,K) x FM(J,K)^T; K >= 4(J2+2) came from the macro tile
        const bool two = l2 > 0 && n >= l2 && 4 * (J2 + 2) <= last;
        const d4 acc = pk_loop(pk + PK_FM2O_A * B.pk_stride, pk + PK_FM_A * B.pk_stride, J + 2, two ? 4 * (J2 + 2) - 1 : last,
                               [=](int K) { return pk_tile(nb, I, K); }, [=](int K) { return pk_tile(nb, J, K); });
        reduce_store(red, acc, tab + (size_t)LF_FM1OF * B.tab_stride, B.ld, n, I * 16, J * 16, two);
    }
}

// ---- 64x64 macro tiles: wavefront w owns the 32x32 quadrant (rows 2(w>>1).., columns 2(w&1)..) of 16-tiles
struct Acc4 { d4 c[2][2]; };
__device__ __forceinline__ void store_quadrant(const Acc4& A, double* __restrict__ T, int ld, int n, int I2, int J2)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int i0 = (4 * I2 + 2 * (w >> 1) + a) * 16, j0 = (4 * J2 + 2 * (w & 1) + b) * 16;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + (lane >> 4) + 4 * r, j = j0 + (lane & 15);
                if (i >= 1 && i <= j && j <= n - 1) T[(size_t)(j - i) * ld + i] = A.c[a][b][r];
            }
        }
}
// acc(a,b) += sum_{K = k_lo}^{k_hi} frag(pa + ta(a,K)) x frag(pb + tb(K,b)); the loads of step K+1 are issued before the MFMAs of step K
template <class TA, class TB>
__device__ __forceinline__ void far2_loop(Acc4& A, const double* __restrict__ pa, const double* __restrict__ pb, int k_lo, int k_hi, TA ta, TB tb)
{
    const int lane = threadIdx.x & 63;
    if (k_lo > k_hi) return;
    d4 a0 = *(const d4*)(pa + ta(0, k_lo) + lane * 4), a1 = *(const d4*)(pa + ta(1, k_lo) + lane * 4);
    d4 b0 = *(const d4*)(pb + tb(k_lo, 0) + lane * 4), b1 = *(const d4*)(pb + tb(k_lo, 1) + lane * 4);
    for (int K = k_lo; K <= k_hi; K++) {
        const int Kn = K < k_hi ? K + 1 : K;
        const d4 na0 = *(const d4*)(pa + ta(0, Kn) + lane * 4), na1 = *(const d4*)(pa + ta(1, Kn) + lane * 4);
        const d4 nb0 = *(const d4*)(pb + tb(Kn, 0) + lane * 4), nb1 = *(const d4*)(pb + tb(Kn, 1) + lane * 4);
        A.c[0][0] = mfma4(a0, b0, A.c[0][0]);
        A.c[0][1] = mfma4(a0, b1, A.c[0][1]);
        A.c[1][0] = mfma4(a1, b0, A.c[1][0]);
        A.c[1][1] = mfma4(a1, b1, A.c[1][1]);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
}

// inside: FM2F(macro tile (I2, I2+D2)) = sum_{K = 4(I2+2)}^{4(J2-1)-1} FM1(I,K) x FM(K,J) over its 4x4 tiles; grid = (macro tiles, sequences)
__global__ __launch_bounds__(256) void lin_far2_inside(McBatch B, int D2, int l2)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    if (n < l2) return;   // whether a sequence takes the two-level form depends on its own length only: its bits do not depend on the batch
    const int I2 = blockIdx.x, J2 = I2 + D2;
    if (J2 * 64 > n - 1) return;
    if (one_strand_tile(B, sq, true, I2 * 64, J2 * 64 + 63)) return;
    const int nb = B.nb, w = threadIdx.x >> 6;
    const int Ia = 4 * I2 + 2 * (w >> 1), Jb = 4 * J2 + 2 * (w & 1);
    const double* __restrict__ pk = B.pk + (size_t)sq * kPkCopies * B.pk_stride;
    Acc4 A;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) A.c[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    far2_loop(A, pk + PK_FM1_A * B.pk_stride, pk + PK_FM_B * B.pk_stride, 4 * (I2 + 2), 4 * (J2 - 1) - 1,
              [=](int a, int K) { return pk_tile(nb, Ia + a, K); }, [=](int K, int b) { return pk_tile(nb, K, Jb + b < nb ? Jb + b : nb - 1); });
    store_quadrant(A, B.tab + (size_t)sq * B.seq_stride + (size_t)LF_FM2F * B.tab_stride, B.ld, n, I2, J2);
}

// outside: blockIdx.z = 0: FMOF(macro tile) = sum_{K <= 4(I2-1)-1} FM1(K,I)^T x FM2o(K,J);  1: FM1OF = sum_{K >= 4(J2+2)} FM2o(I,K) x FM(J,K)^T
__global__ __launch_bounds__(256) void lin_far2_outside(McBatch B, int D2, int l2)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    if (n < l2) return;
    const int I2 = blockIdx.x, J2 = I2 + D2;
    if (J2 * 64 > n - 1) return;
    if (one_strand_tile(B, sq, false, I2 * 64, J2 * 64 + 63)) return;
    const int nb = B.nb, w = threadIdx.x >> 6, last = (n - 1) / 16;
    const int Ia = 4 * I2 + 2 * (w >> 1), Jb = 4 * J2 + 2 * (w & 1);
    const double* __restrict__ pk = B.pk + (size_t)sq * kPkCopies * B.pk_stride;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    Acc4 A;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) A.c[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    const auto cl = [=](int Q) { return Q < nb ? Q : nb - 1; };   // a tile column past the sequence is never stored: any address will do
    if (blockIdx.z == 0) {
        if (I2 < 2) return;   // nothing 64 or more to the left: the tile kernel writes these cells
        far2_loop(A, pk + PK_FM1_B * B.pk_stride, pk + PK_FM2O_B * B.pk_stride, 0, 4 * (I2 - 1) - 1,
                  [=](int a, int K) { return pk_tile(nb, K, Ia + a); }, [=](int K, int b) { return pk_tile(nb, K, cl(Jb + b)); });
        store_quadrant(A, tab + (size_t)LF_FMOF * B.tab_stride, B.ld, n, I2, J2);
    } else {
        if (4 * (J2 + 2) > last) return;
        far2_loop(A, pk + PK_FM2O_A * B.pk_stride, pk + PK_FM_A * B.pk_stride, 4 * (J2 + 2), last,
                  [=](int a, int K) { return pk_tile(nb, Ia + a, K); }, [=](int K, int b) { return pk_tile(nb, cl(Jb + b), K); });
        store_quadrant(A, tab + (size_t)LF_FM1OF * B.tab_stride, B.ld, n, I2, J2);
    }
}

template __global__ void lin_pack_tiles<0>(McBatch, int, int, int);
template __global__ void lin_pack_tiles<1>(McBatch, int, int, int);
template __global__ void lin_far_inside<16>(McBatch, int);
template __global__ void lin_far_inside<32>(McBatch, int);
template __global__ void lin_far_outside<16>(McBatch, int);
template __global__ void lin_far_outside<32>(McBatch, int);

}  // namespace rh
