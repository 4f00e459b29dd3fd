// kernels.hip.h — gfx950 (CDNA4, wave64) device code of libroman_hip.so.
//
// Pipeline for a batch of B independent submap pairs (one "problem" each):
//   k_cos        normalised cosine matrix (+ descriptor norms), f64 MFMA 16x16x4   (cos_feature_dim > 0)
//   k_tables     intra-map distance tables with NaN sentinels      (n1^2 + n2^2 entries)
//   k_live       single scores + ordered compaction of live associations; picks the problem's KIND
//   k_rowbase / k_items   prefix of live counts over problems, work-item list
//   k_count      the O(L^2) pair tests -> candidate bit matrix (upper 64-bit words), k_mirror the lower ones
//   k_rowprefix  row degrees (+ per-word prefix counts for the fallback layout)
//   k_rowsort    rows by descending degree (+ kind 0: offsets of the candidate lists)
//   kind 0 ("stream", L <= STREAM_MAXL):
//     k_upper      every candidate pair kept once, in the row of its endpoint with the smaller POSITION (rank by degree):
//                  the row's kept candidates as a list of 16-bit live column indices
//     k_slicegeom  slice widths / bases of the quad layout
//     k_fill_list  candidate lists -> values, one quad (4 entries per lane = row) at a time, straight into the layout
//     k_solve_up   persistent per-problem CLIPPER solve on the upper triangle (pull + push SpMV)
//   kind 1 (fallback, any L): symmetric sorted SELL-64 in live numbering: k_fill, then k_solve (a workgroup per
//     problem) or k_solve_wide (cooperative launch: the whole device on one large problem at a time, or teams of compute
//     units — the workgroups of an XCD or of half an XCD — on one problem each)
//   kind 2: skipped for lack of workspace (k_skipped writes ROMAN_ST_WORKSPACE)
//   kind 3: finished by k_small — batch calls, <= 128 live associations (the reference's demo scale): pair tests, positions,
//     values, solve and pose in ONE kernel right behind k_live; the kernels above pass such a problem by
//   (k_cos_wave: the cosine matrices of maps of at most 48 objects, one wave per problem)
//
// Matrix layout of kind 0: only the strict upper triangle of M in position numbering is stored
// (entry (p,q), p < q, in row p) — 10 bytes per non-zero of the upper triangle.  Rows are cut into slices of 64
// consecutive positions, a slice is padded to its longest row (multiple of 4 entries) and stored as "quads":
// the 4 column indices of entries 4g..4g+3 of a lane are one 8-byte word, the values of entries 2h,2h+1 one
// 16-byte pair: 3 wide, fully coalesced loads per 4 entries.  Padding is inert (value 0, column L + lane, C-flag).
// Layout of kind 1 (sorted SELL-64, both triangles): entry e of the row in slot (slice s, lane l) sits at
//      sliceBase[s] + e*64 + l .
#pragma once
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/roman_hip.h"

namespace roman {

constexpr int WAVE = 64;
constexpr int STREAM_MAXL = 3072;        // live associations the stream layout / solver serve (48 slices of 64 rows)

struct DevParams {
    roman_params_t p;
    double sig2;        // sigma^2
    double sin_unc;     // sin(gravity_unc_ang_rad)
    double x_eps;       // smallest x with sqrt(x) >= epsilon   (c < eps  <=>  c*c-sum < x_eps)
    double x_mindist;   // smallest x with sqrt(x) >= mindist   (l < mindist <=> l^2 < x_mindist)
    int32_t single;     // invariant has per-association scores
    int32_t gravity;    // ROMAN invariant && gravity_guided
    int32_t F;          // features per object
    int32_t max_compact; // streaming solver: column compactions allowed per problem (speed only; set per launch)
    int32_t gmode;      // 0: no gravity; 1 + ROMAN_GRAV_* otherwise (1 combined, 2 separate gates, 3 z gate on full lengths)
    int32_t diag_one;   // single scores present but the diagonal is the identity (ROMAN_SINGLE_OFFDIAG)
    int32_t keep_all;   // single scores present but a zero one removes nothing (ROMAN_SINGLE_DIAG_KEEP): every association is live
    int32_t pruned;     // ROMAN_INV_EUCLIDEAN_PRUNED: the single score is the 0/1 prefilter on the RAW descriptor product; pair score alone in M,
                        // identity diagonal; a problem in which nothing survives keeps every association; rounding over survivors only
    int32_t allow_fallback;  // the fallback kernels are part of this launch; otherwise a problem that does not fit the stream layout is
                             // SKIPPED (kind 2, ROMAN_ST_WORKSPACE) and runs again with them (set per launch, from the sizing history)
    int32_t wide;            // fallback problems of this launch go to k_solve_wide (few, large) instead of k_solve (set per launch)
    int32_t idx16;           // ... and their column labels are 16 bits wide (no C flag; 0xffff = inert): 10 instead of 12 bytes per entry
    int32_t solve_flags;     // experiments: bit 0 = the one-wave solver keeps the quad stream (no coordinate list in registers); bit 1 = k_fill_list does NOT rotate
                             // the entries of a row (the rotation decorrelates the lanes' LDS pushes in the solver)
    int32_t small_only;      // the general kernels are NOT part of this launch (every problem of this parameter block has so far been finished by
                             // k_small): a problem k_small leaves behind is skipped (kind 2, ROMAN_ST_WORKSPACE) and takes them on its second run
    int32_t stream_maxL;     // problems of up to this many live associations take the stream layout (<= STREAM_MAXL; set per launch:
                             // it is also the column capacity of k_fill_slice's LDS tile and of the stream solver's LDS vectors)
    int32_t pre_K;           // k_count's integer prefilter: a pair goes to the exact gate iff its two quantised table entries differ by at most pre_K bins
    double  pre_invw;        // ... bins per metre (1 / bin width; bin width = epsilon / 32)
};

struct ProbDesc {
    int64_t off1, off2;    // first object of map 1 / map 2 in the feature pool
    int64_t assocOff;      // row offset into the explicit association list, -1 = all-to-all
    int64_t liveOff;       // offset into pools with one slot per input association
    int64_t cosOff;        // offset into the cosine pool (n1*n2)
    int64_t tabOff;        // offset into the table pool (n1*n1 then n2*n2)
    int64_t normOff;       // offset into the norm pool (n1 then n2)
    int32_t n1, n2, nA;
    int32_t qtabOff4;      // offset into the pool of 16-bit table bins (k_count's prefilter), in units of four entries: n1 rows of (n1 + 3 & ~3), then n2 rows of (n2 + 3 & ~3)
};

struct ProbState {
    int32_t  L;            // live associations
    int32_t  rowBase;      // prefix of L over the batch
    int64_t  nnzOff;       // offset of this problem's matrix segment
    int64_t  maskOff;      // offset (in 64-bit words) of this problem's candidate bit matrix
    uint32_t nnzCap;       // padded SELL slots allocated for this problem
    int32_t  itemBase;     // first work item (row block) of this problem
    int32_t  sgBase;       // first slice group (k_fill_slice work item) of this problem
    int32_t  kind;         // 0: stream layout + k_solve_up (L <= STREAM_MAXL), 1: symmetric SELL-64 fallback
    unsigned long long nnzUpper;   // stored strict-upper non-zeros (after the affinityeps filter)
    int64_t  listOff;      // stream layout: first element of this problem's candidate lists in the list pool
};

struct BatchTotals {
    int64_t nnzTotal;      // sum of nnzCap over the problems that fit
    int64_t maskWords;     // sum over the problems that fit of L * ceil(L/64)
    int32_t R;             // sum of L
    int32_t maxL;
    int32_t items;         // work items (row blocks) of the pair-test / fill kernels
    int32_t sliceGroups;   // work items (groups of SPI slices) of k_fill_slice
    // what the whole batch WOULD need (the pools are sized before the live counts are known: problems that do not fit
    // are skipped with ROMAN_ST_WORKSPACE and the host grows the pools from these numbers)
    int64_t needMaskWords;
    int64_t needNnz;
    int32_t overflow;      // problems skipped for lack of workspace
    int32_t maxStreamL;    // largest L among stream-layout problems
    int32_t minStreamL;    // smallest L among stream-layout problems (does the small-problem solver have work?)
    int32_t nGeneral;      // problems left to the general kernels (not finished by k_small, not skipped)
    unsigned long long listTop;    // bump pointer of the candidate-list pool = what the whole batch needs of it
    int32_t cosScreened;   // problems whose cosines went through k_cos_sel (0: the dense kernels took the batch) ...
    int32_t cosDense;      // ... and how many of them it left to the dense kernel (more candidates than its list holds)
};

struct ItemDesc { int32_t b, row0; };   // a block of consecutive live rows of problem b

// column-index word of a stored entry: live column index + a flag bit "C_pq == 0"
template <typename IdxT> struct IdxTraits;
template <> struct IdxTraits<uint16_t> { static constexpr uint32_t CZ = 0x8000u; static constexpr uint32_t MASK = 0x7fffu; };
template <> struct IdxTraits<uint32_t> { static constexpr uint32_t CZ = 0x80000000u; static constexpr uint32_t MASK = 0x7fffffffu; };

// Fallback layout, column word of an inert entry (padding / filtered): 32-bit words carry the row's own position with the C
// flag (a real vector element, excluded from C); 16-bit words (k_solve_wide only, no flag bit) are 0xffff = "no column".
template <typename IdxT> __device__ __forceinline__ IdxT fb_inert(uint32_t pos)
{
    return sizeof(IdxT) == 2 ? (IdxT)0xffffu : (IdxT)(pos | IdxTraits<IdxT>::CZ);
}

// Position of entry e of the row in lane-slot `slot` of a slice that starts at element `sbase` of the
// problem's matrix segment (sbase is a multiple of 256 in the quad layout, of 64 otherwise).
//  - SELL-64 (QUAD == false):  sbase + e*64 + slot  for both arrays.
//  - quad layout (QUAD == true, 16-bit indices, widths multiples of 4): the 4 column indices of entries
//    4g..4g+3 of a lane are contiguous (one 8-byte load), the values of entries 2h, 2h+1 are contiguous
//    (one 16-byte load): 3 wide loads per 4 entries instead of 8 narrow ones.
template <bool QUAD> __device__ __forceinline__ int64_t col_pos(int64_t sbase, uint32_t slot, uint32_t e)
{
    return QUAD ? sbase + (int64_t)(e >> 2) * 256 + slot * 4 + (e & 3u) : sbase + (int64_t)e * 64 + slot;
}
template <bool QUAD> __device__ __forceinline__ int64_t val_pos(int64_t sbase, uint32_t slot, uint32_t e)
{
    return QUAD ? sbase + (int64_t)(e >> 1) * 128 + slot * 2 + (e & 1u) : sbase + (int64_t)e * 64 + slot;
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ double d_nan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }    // wave-uniform value into an SGPR

// association index p of problem pd -> (map-1 object, map-2 object)
__device__ __forceinline__ void decode_assoc(const ProbDesc& pd, const int32_t* __restrict__ assoc,
                                             int p, int& i, int& j)
{
    if (pd.assocOff >= 0) { i = assoc[2 * (pd.assocOff + p)]; j = assoc[2 * (pd.assocOff + p) + 1]; }
    else { i = p / pd.n2; j = p - i * pd.n2; }
}

// exp and cbrt as FIXED sequences of correctly-rounded operations: the `score > affinityeps`
// gate is applied after them, so they must produce the same bits as the oracle's oracle_exp()/oracle_cbrt()
// (oracle/clipper_oracle.c states the same sequences; tests/test_gpu_parity.py compares the bits).  < 1 ulp.
__device__ __forceinline__ double bits_f64(unsigned long long b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ double fx_exp(double y)
{
    if (!(y > -700.0 && y < 700.0)) return exp(y);
    constexpr double LOG2E = 0x1.71547652b82fep+0, LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    const double k = rint(y * LOG2E);
    const double r1 = fma(-k, LN2_HI, y);
    const double rl = -k * LN2_LO;
    const double r = r1 + rl;
    const double r_err = (r1 - r) + rl;
    double q = 0x1.6124613a86d09p-33;                              // 1/13!
    q = fma(q, r, 0x1.1eed8eff8d898p-29); q = fma(q, r, 0x1.ae64567f544e4p-26); q = fma(q, r, 0x1.27e4fb7789f5cp-22);
    q = fma(q, r, 0x1.71de3a556c734p-19); q = fma(q, r, 0x1.a01a01a01a01ap-16); q = fma(q, r, 0x1.a01a01a01a01ap-13);
    q = fma(q, r, 0x1.6c16c16c16c17p-10); q = fma(q, r, 0x1.1111111111111p-7);  q = fma(q, r, 0x1.5555555555555p-5);
    q = fma(q, r, 0x1.5555555555555p-3);  q = fma(q, r, 0.5);     // ... 1/2!
    const double a = 1.0 + r;
    const double a_err = (r - (a - 1.0)) + r_err;
    const double p = a + fma(r * r, q, a_err);
    return p * bits_f64((unsigned long long)(1023 + (int)k) << 52);
}
__device__ __forceinline__ double fx_cbrt(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const int E = (int)((b >> 52) & 0x7ffull);
    if (!(x > 0.0) || E == 0 || E == 0x7ff) return cbrt(x);
    constexpr double C0 = 0x1.331e76e38c2bfp+0, C1 = -0x1.142641324f5d9p-2, C2 = 0x1.49dcf893faf42p-5, C3 = -0x1.2190c96665e59p-9;
    constexpr double THIRD = 0x1.5555555555555p-2;
    const int e = E - 1023;
    const int q = (e >= 0) ? e / 3 : -((2 - e) / 3);
    const int rem = e - 3 * q;
    const double m = bits_f64((b & 0x000fffffffffffffull) | ((unsigned long long)(1023 + rem) << 52));
    double y = fma(fma(fma(C3, m, C2), m, C1), m, C0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const double y2 = y * y, y3 = y2 * y;
        const double w = fma(-m, y3, 4.0);
        y = (y * w) * THIRD;
    }
    const double yy = y * y;
    double t = m * yy;
    const double t2 = t * t, e2 = fma(t, t, -t2);
    const double r = fma(-e2, t, fma(-t2, t, m));
    t = fma(r, yy * THIRD, t);
    return t * bits_f64((unsigned long long)(1023 + q) << 52);
}

__device__ __forceinline__ double root_w(double x, double w)
{
    if (w == 1.0) return x;
    if (w == 2.0) return sqrt(x);
    if (w == 3.0) return fx_cbrt(x);
    if (w == 4.0) return sqrt(sqrt(x));
    if (w == 6.0) return fx_cbrt(sqrt(x));
    return pow(x, 1.0 / w);
}
__device__ __forceinline__ double pow_w(double x, double w) { return (w == 1.0) ? x : pow(x, w); }

// Single score of one association; mirrors oracle single_score() operation for operation.
// ra(f), rb(f): ratio feature f of the two objects.
template <class RA, class RB>
__device__ __forceinline__ double single_score(const DevParams& D, RA ra, RB rb, double cosv)
{
    const roman_params_t& P = D.p;
    const int Fr = P.ratio_feature_dim, Fc = P.cos_feature_dim;
    if (D.pruned) {                         // the reference's NumPy prefilter ([REF roman/align/dist_reg_with_pruning.py:75-90]) as a 0/1 score;
        if (Fc > 0 && cosv < P.cosine_min) return 0.0;          // cosv is the RAW product here; a NaN compares false = keeps, as in NumPy
        for (int f = 0; f < Fr; ++f) {
            const double a = ra(f), b = rb(f);
            const double mn = a < b ? a : b, mx = a < b ? b : a;
            if (mn / mx < P.ratio_epsilon[f]) return 0.0;
        }
        return 1.0;
    }
    double wsum = 0.0, prod = 1.0, asum = 0.0;
    if (Fr > 0) {
        double rp = 1.0;
        for (int f = 0; f < Fr; ++f) {
            const double a = ra(f), b = rb(f);
            const double mn = a < b ? a : b, mx = a < b ? b : a;
            const double r = (mx > 0.0) ? mn / mx : 1.0;
            if (r < P.ratio_epsilon[f]) return 0.0;
            rp *= r;
        }
        const double R = root_w(rp, (double)Fr);
        prod *= pow_w(R, P.ratio_weight); asum += P.ratio_weight * R; wsum += P.ratio_weight;
    }
    if (Fc > 0) {
        double c = (cosv - P.cosine_min) / (P.cosine_max - P.cosine_min);
        if (!(c > 0.0)) return 0.0;
        if (c > 1.0) c = 1.0;
        prod *= pow_w(c, P.cosine_weight); asum += P.cosine_weight * c; wsum += P.cosine_weight;
    }
    if (wsum == 0.0) return 1.0;
    switch (P.fusion_method) {
    case ROMAN_FUSE_ARITHMETIC_MEAN: return asum / wsum;
    case ROMAN_FUSE_PRODUCT:         return prod;
    default:                         return root_w(prod, wsum);
    }
}

// Fusion of the pair score with the two single scores; mirrors oracle fuse_pair().
__device__ __forceinline__ double fuse_pair(const DevParams& D, double sa, double sp, double sq)
{
    if (!D.single || D.keep_all) return sa;
    if (sa == 0.0 || sp == 0.0 || sq == 0.0) return 0.0;
    if (D.pruned) return sa;
    if (D.p.single_mode == ROMAN_SINGLE_DIAG) return sa;
    const double ss = sp * sq, wd = D.p.distance_weight;
    switch (D.p.fusion_method) {
    case ROMAN_FUSE_ARITHMETIC_MEAN: return (wd * sa + (sp + sq)) / (wd + 2.0);
    case ROMAN_FUSE_PRODUCT:         return sa * ss;
    default:                         return root_w(pow_w(sa, wd) * ss, wd + 2.0);
    }
}

// ---------------------------------------------------------------------------------------------
// k_cos: cos[i][j] = <d1_i, d2_j> / (|d1_i| |d2_j|) on the f64 matrix core (v_mfma_f64_16x16x4_f64).
// One wave owns a 32x32 output tile (2x2 MFMA tiles).  Operand layout of one MFMA: lane l supplies
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; result reg r of lane l is C[row = (l>>4) + 4r][col = l&15].
// Per 16 descriptor elements a lane loads 4 CONSECUTIVE doubles of each of its 2+2 rows (32-byte
// loads) and feeds element t of every load to MFMA t: MFMA t contracts k = k0 + 4*(l>>4) + t — any
// bijection k <-> (MFMA, l>>4) is a valid contraction order as long as A and B use the same one.
// 16 MFMAs per 4 wide loads (the previous 16x16 tile issued 1 MFMA per 2 scalar loads).
// ---------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double dbl2_t __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) d4u_t { double v[4]; };     // 8-byte aligned 32-byte load

constexpr int COS_TILE = 32;

__global__ void __launch_bounds__(256) k_cos(DevParams D, int B, int G /* workgroups (4 tiles each) per problem */,
                                             const ProbDesc* __restrict__ probs,
                                             const double* __restrict__ feats,
                                             double* __restrict__ cosPool)
{
    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  All G workgroups of a problem get ids that are
    // congruent modulo 8, i.e. one XCD and one L2: the descriptors of a problem are read from HBM once instead
    // of once per XCD (8 problems are interleaved: id = 8 * (G * (b / 8) + g) + b % 8).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / G) * 8 + xcd;
    if (b >= B) return;
    const ProbDesc pd = probs[b];
    const int tj_n = (pd.n2 + COS_TILE - 1) / COS_TILE, ti_n = (pd.n1 + COS_TILE - 1) / COS_TILE;
    const int tile = (slot % G) * 4 + (threadIdx.x >> 6);
    if (tile >= ti_n * tj_n) return;
    const int lane = threadIdx.x & 63;
    const int i0 = (tile / tj_n) * COS_TILE, j0 = (tile % tj_n) * COS_TILE;
    const int Fc = D.p.cos_feature_dim, coff = D.p.point_dim + D.p.ratio_feature_dim;
    const int lr = lane & 15, kq = lane >> 4;
    const double* fa[2]; const double* fb[2]; bool va[2], vb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ia = i0 + 16 * h + lr, jb = j0 + 16 * h + lr;
        va[h] = ia < pd.n1; vb[h] = jb < pd.n2;
        fa[h] = feats + (pd.off1 + (va[h] ? ia : 0)) * D.F + coff + 4 * kq;
        fb[h] = feats + (pd.off2 + (vb[h] ? jb : 0)) * D.F + coff + 4 * kq;
    }
    // A map of 200 objects ends 8 rows into its last 32-row tile: the second 16-row block of that tile holds no object at
    // all.  Such blocks are skipped (wave-uniform: loads and MFMAs): 13 instead of 14 blocks per dimension at n = 200.
    const bool on[2][2] = {{true, uni_i(j0 + 16 < pd.n2) != 0}, {uni_i(i0 + 16 < pd.n1) != 0, uni_i(i0 + 16 < pd.n1) != 0 && uni_i(j0 + 16 < pd.n2) != 0}};
    const bool onA1 = on[1][0], onB1 = on[0][1];
    double4_t acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
    // squared norms of the tile's 32 + 32 descriptors ride along on the VALU (the operands are in registers
    // anyway): lane (lr, kq) accumulates the elements it holds, the four kq-lanes of a row are added at the end
    double sa[2] = {0.0, 0.0}, sb[2] = {0.0, 0.0};
    int k0 = 0;
    for (; k0 + 16 <= Fc; k0 += 16) {
        d4u_t a[2], b[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0 || onA1) a[h] = *reinterpret_cast<const d4u_t*>(fa[h] + k0); else a[h] = d4u_t{{0.0, 0.0, 0.0, 0.0}};
            if (h == 0 || onB1) b[h] = *reinterpret_cast<const d4u_t*>(fb[h] + k0); else b[h] = d4u_t{{0.0, 0.0, 0.0, 0.0}};
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { sa[h] = fma(a[h].v[t], a[h].v[t], sa[h]); sb[h] = fma(b[h].v[t], b[h].v[t], sb[h]); }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    if (on[x][y]) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[x] ? a[x].v[t] : 0.0, vb[y] ? b[y].v[t] : 0.0, acc[x][y], 0, 0, 0);
        }
    }
    if (k0 < Fc) {                                   // ragged tail of the descriptor (< 16 elements)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = k0 + 4 * kq + t;
            const bool vk = kk < Fc;
            double av[2], bv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                av[h] = (va[h] && vk) ? fa[h][k0 + t] : 0.0;
                bv[h] = (vb[h] && vk) ? fb[h][k0 + t] : 0.0;
                sa[h] = fma(av[h], av[h], sa[h]); sb[h] = fma(bv[h], bv[h], sb[h]);
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    if (on[x][y]) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                    // every lane (lr, *) ends with the norm of row 16h + lr
        sa[h] += __shfl_xor(sa[h], 16); sa[h] += __shfl_xor(sa[h], 32);
        sb[h] += __shfl_xor(sb[h], 16); sb[h] += __shfl_xor(sb[h], 32);
        sa[h] = sqrt(sa[h]); sb[h] = sqrt(sb[h]);
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int col = j0 + 16 * y + lr;
        const double nb = sb[y];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 16 * x + kq + 4 * r;
                const double na = __shfl(sa[x], kq + 4 * r);       // norm of row 16x + (kq + 4r): held by lanes with lr == kq + 4r
                if (row < pd.n1 && col < pd.n2)
                    cosPool[pd.cosOff + (int64_t)row * pd.n2 + col] = D.pruned ? acc[x][y][r] : ((na > 0.0 && nb > 0.0) ? acc[x][y][r] / (na * nb) : 0.0);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// k_cos_wave: the reference's DEMO scale (submaps of at most 48 objects): ONE WAVE computes a problem's whole cosine matrix —
// up to 3 x 3 MFMA blocks in registers, operands straight from global memory with 32-byte loads (the descriptors of a few
// hundred small submaps are L2-resident), no LDS, no barrier.  k_cos_tile gives such a problem a 4-wave workgroup that
// meets at a barrier 48 times (768-d descriptors) to multiply at most 9 blocks: 0.32 ms per 4096 problems, three times the
// matrix pipe's own time.  Per 16 descriptor elements a lane loads 4 consecutive doubles of each of its <= 3 + 3 rows and
// feeds element t of every load to the MFMAs of step t: the contraction order of every output element — chunks of 16
// ascending, step t of a chunk contracts k = k0 + 4 * (lane >> 4) + t — and the norm sums are k_cos's own: identical bits.
// Blocks that hold no object (n <= 32: the third, n <= 16: the second) are skipped, wave-uniformly.
// ---------------------------------------------------------------------------------------------
constexpr int COSW_NB = 3;               // MFMA blocks per dimension a wave can hold: maps of up to 48 objects

__global__ void __launch_bounds__(256) k_cos_wave(DevParams D, int B, const ProbDesc* __restrict__ probs,
                                                  const double* __restrict__ feats, double* __restrict__ cosPool)
{
    const int b = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (b >= B) return;
    const ProbDesc pd = probs[b];
    if (pd.n1 <= 0 || pd.n2 <= 0) return;
    const int lane = threadIdx.x & 63;
    const int Fc = D.p.cos_feature_dim, coff = D.p.point_dim + D.p.ratio_feature_dim;
    const int lr = lane & 15, kq = lane >> 4;
    const int nbi = uni_i((pd.n1 + 15) >> 4), nbj = uni_i((pd.n2 + 15) >> 4);      // blocks in use per dimension (<= COSW_NB)
    const double* fa[COSW_NB]; const double* fb[COSW_NB]; bool va[COSW_NB], vb[COSW_NB];
#pragma unroll
    for (int h = 0; h < COSW_NB; ++h) {
        const int ia = 16 * h + lr, jb = 16 * h + lr;
        va[h] = ia < pd.n1; vb[h] = jb < pd.n2;
        fa[h] = feats + (pd.off1 + (va[h] ? ia : 0)) * D.F + coff + 4 * kq;
        fb[h] = feats + (pd.off2 + (vb[h] ? jb : 0)) * D.F + coff + 4 * kq;
    }
    double4_t acc[COSW_NB][COSW_NB];
#pragma unroll
    for (int x = 0; x < COSW_NB; ++x)
#pragma unroll
        for (int y = 0; y < COSW_NB; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
    double sa[COSW_NB], sb[COSW_NB];
#pragma unroll
    for (int h = 0; h < COSW_NB; ++h) { sa[h] = 0.0; sb[h] = 0.0; }
    // the loads of chunk k0 + 16 are in flight while chunk k0 is multiplied (two register sets; all loads unconditional —
    // clamped to the last full chunk — so that the compiler waits with a count, not for an empty queue)
    int k0 = 0;
    const int nfull = Fc >> 4;                       // full chunks of 16
    d4u_t a[COSW_NB], bq[COSW_NB], a2[COSW_NB], b2[COSW_NB];
    auto fetch = [&](d4u_t (&ra)[COSW_NB], d4u_t (&rb)[COSW_NB], int kk) {
#pragma unroll
        for (int h = 0; h < COSW_NB; ++h) {
            if (h < nbi) ra[h] = *reinterpret_cast<const d4u_t*>(fa[h] + kk); else ra[h] = d4u_t{{0.0, 0.0, 0.0, 0.0}};
            if (h < nbj) rb[h] = *reinterpret_cast<const d4u_t*>(fb[h] + kk); else rb[h] = d4u_t{{0.0, 0.0, 0.0, 0.0}};
        }
    };
    auto multiply = [&](const d4u_t (&ra)[COSW_NB], const d4u_t (&rb)[COSW_NB]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int h = 0; h < COSW_NB; ++h) { sa[h] = fma(ra[h].v[t], ra[h].v[t], sa[h]); sb[h] = fma(rb[h].v[t], rb[h].v[t], sb[h]); }
#pragma unroll
            for (int x = 0; x < COSW_NB; ++x)
#pragma unroll
                for (int y = 0; y < COSW_NB; ++y)
                    if (x < nbi && y < nbj) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[x] ? ra[x].v[t] : 0.0, vb[y] ? rb[y].v[t] : 0.0, acc[x][y], 0, 0, 0);
        }
    };
    if (nfull > 0) {
        fetch(a, bq, 0);
        for (int c = 0; c < nfull; c += 2) {
            fetch(a2, b2, 16 * min(c + 1, nfull - 1));
            multiply(a, bq);
            if (c + 1 < nfull) {
                fetch(a, bq, 16 * min(c + 2, nfull - 1));
                multiply(a2, b2);
            }
        }
        k0 = 16 * nfull;
    }
    if (k0 < Fc) {                                   // ragged tail of the descriptor (< 16 elements)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = k0 + 4 * kq + t;
            const bool vk = kk < Fc;
            double av[COSW_NB], bv[COSW_NB];
#pragma unroll
            for (int h = 0; h < COSW_NB; ++h) {
                av[h] = (h < nbi && va[h] && vk) ? fa[h][k0 + t] : 0.0;
                bv[h] = (h < nbj && vb[h] && vk) ? fb[h][k0 + t] : 0.0;
                sa[h] = fma(av[h], av[h], sa[h]); sb[h] = fma(bv[h], bv[h], sb[h]);
            }
#pragma unroll
            for (int x = 0; x < COSW_NB; ++x)
#pragma unroll
                for (int y = 0; y < COSW_NB; ++y)
                    if (x < nbi && y < nbj) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
    }
#pragma unroll
    for (int h = 0; h < COSW_NB; ++h) {              // every lane (lr, *) ends with the norm of row 16h + lr
        sa[h] += __shfl_xor(sa[h], 16); sa[h] += __shfl_xor(sa[h], 32);
        sb[h] += __shfl_xor(sb[h], 16); sb[h] += __shfl_xor(sb[h], 32);
        sa[h] = sqrt(sa[h]); sb[h] = sqrt(sb[h]);
    }
#pragma unroll
    for (int y = 0; y < COSW_NB; ++y) {
        const int col = 16 * y + lr;
        const double nb = sb[y];
#pragma unroll
        for (int x = 0; x < COSW_NB; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * x + kq + 4 * r;
                const double na = __shfl(sa[x], kq + 4 * r);       // norm of row 16x + (kq + 4r): held by lanes with lr == kq + 4r
                if (row < pd.n1 && col < pd.n2)
                    cosPool[pd.cosOff + (int64_t)row * pd.n2 + col] = D.pruned ? acc[x][y][r] : ((na > 0.0 && nb > 0.0) ? acc[x][y][r] / (na * nb) : 0.0);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// k_cos_block: the cosine matrices of a FEW problems (a serial caller's one pair at a time) — maps of any size: a single pair of 200-object
// maps is 169 blocks on 169 waves, 6 us where the tile kernel's sixteen workgroups, two stages of loads in flight, take 29.  One wave alone walks a demo-size problem's
// 48 chunks x 36 matrix instructions in 68 us — the whole device idle beside it; here a wave takes ONE 16 x 16 block of the
// cosine matrix (up to nine waves per problem): four matrix instructions per chunk, eight chunks of loads in flight (a ring of
// 32-byte loads, refilled as it is consumed).  Every output element sees k_cos_wave's own contraction order — chunks of 16
// ascending, step t of a chunk contracts k = k0 + 4 * (lane >> 4) + t — and the norm sums are formed the same way: identical bits.
// ---------------------------------------------------------------------------------------------
constexpr int COSB_DEPTH = 8;

__global__ void __launch_bounds__(256) k_cos_block(DevParams D, int B, int nbx, int nby /* blocks per dimension of the largest problem of the call */,
                                                   const ProbDesc* __restrict__ probs,
                                                   const double* __restrict__ feats, double* __restrict__ cosPool)
{
    const int wid = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int per = nbx * nby;
    const int b = uni_i(wid / per);
    if (b >= B) return;
    const int blk = uni_i(wid - b * per), xs = blk / nby, ys = blk - xs * nby;
    const ProbDesc pd = probs[b];
    if (16 * xs >= pd.n1 || 16 * ys >= pd.n2) return;
    const int lane = threadIdx.x & 63;
    const int Fc = D.p.cos_feature_dim, coff = D.p.point_dim + D.p.ratio_feature_dim;
    const int lr = lane & 15, kq = lane >> 4;
    const int ia = 16 * xs + lr, jb = 16 * ys + lr;
    const bool va = ia < pd.n1, vb = jb < pd.n2;
    const double* fa = feats + (pd.off1 + (va ? ia : 0)) * D.F + coff + 4 * kq;
    const double* fb = feats + (pd.off2 + (vb ? jb : 0)) * D.F + coff + 4 * kq;
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
    double sa = 0.0, sb = 0.0;
    const int nfull = Fc >> 4;
    if (nfull > 0) {
        d4u_t ra[COSB_DEPTH], rb[COSB_DEPTH];
#pragma unroll
        for (int t = 0; t < COSB_DEPTH; ++t) {
            const int kk = 16 * min(t, nfull - 1);
            ra[t] = *reinterpret_cast<const d4u_t*>(fa + kk); rb[t] = *reinterpret_cast<const d4u_t*>(fb + kk);
        }
        for (int c0 = 0; c0 < nfull; c0 += COSB_DEPTH) {
#pragma unroll
            for (int t = 0; t < COSB_DEPTH; ++t) {
                const d4u_t a = ra[t], q = rb[t];
                const int kn = 16 * min(c0 + t + COSB_DEPTH, nfull - 1);
                ra[t] = *reinterpret_cast<const d4u_t*>(fa + kn); rb[t] = *reinterpret_cast<const d4u_t*>(fb + kn);
                if (c0 + t < nfull) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sa = fma(a.v[e], a.v[e], sa); sb = fma(q.v[e], q.v[e], sb);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va ? a.v[e] : 0.0, vb ? q.v[e] : 0.0, acc, 0, 0, 0);
                    }
                }
            }
        }
    }
    const int k0 = 16 * nfull;
    if (k0 < Fc) {                                   // ragged tail of the descriptor (< 16 elements)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = k0 + 4 * kq + t;
            const bool vk = kk < Fc;
            const double av = (va && vk) ? fa[k0 + t] : 0.0, bv = (vb && vk) ? fb[k0 + t] : 0.0;
            sa = fma(av, av, sa); sb = fma(bv, bv, sb);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
    }
    sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
    sb += __shfl_xor(sb, 16); sb += __shfl_xor(sb, 32);
    sa = sqrt(sa); sb = sqrt(sb);
    const int col = 16 * ys + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * xs + kq + 4 * r;
        const double na = __shfl(sa, kq + 4 * r);
        if (row < pd.n1 && col < pd.n2)
            cosPool[pd.cosOff + (int64_t)row * pd.n2 + col] = D.pruned ? acc[r] : ((na > 0.0 && sb > 0.0) ? acc[r] / (na * sb) : 0.0);
    }
}

// ---------------------------------------------------------------------------------------------
// k_cos_tile<KC>: the same products, operands staged through LDS.  A workgroup (4 waves) owns an output tile of up to 64x64,
// wave (wy, wx) the 2x2 MFMA blocks of its quarter.  Per stage of KC descriptor elements the 256 threads copy the tile's
// row pieces of KC doubles from global memory to LDS with 16-byte loads whose lanes run ALONG a row (SEGS lanes cover one
// row piece contiguously): every cache line that is touched is used completely, and a row is fetched once per workgroup
// instead of once per wave — 256 bytes of loads per MFMA instead of 512.  Why that matters (rocprofv3 round 3, config 3):
// the matrix pipe of k_cos is busy 42 % of the time; a load takes ~1.6 us under this load, so keeping the pipe fed takes
// (bytes per MFMA) x (MFMA rate) x 1.6 us ~ 100 KB in flight per compute unit, and a wave of k_cos has nothing in flight
// while it multiplies.  Here the loads of stages s + 1 and s + 2 are in flight while stage s is multiplied (two register
// sets, two LDS stages, one barrier per stage).
// Tiles are BALANCED: a dimension of nb 16-row blocks is cut into ceil(nb / 4) tiles of nb / tiles blocks, rounded either
// way (13 blocks: 3 + 3 + 3 + 4, not 4 + 4 + 4 + 1) — a workgroup with one block column would wait for its loads 32 times
// to issue a quarter of the MFMAs while it holds a quarter of the compute unit's registers.
// The contraction order of every output element is k_cos' own — chunks of 16 ascending, MFMA t of a chunk contracts
// k = k0 + 4 * (lane >> 4) + t —, so are the norm sums: identical bits.
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) d2u_t { double v[2]; };      // 8-byte aligned 16-byte load

__device__ __forceinline__ int cos_tiles(int n) { return (((n + 15) >> 4) + 3) >> 2; }            // tiles along a dimension of n rows

template <int KC>
__global__ void __launch_bounds__(256) k_cos_tile(DevParams D, int B, int G /* workgroups (tiles) per problem */,
                                                  const ProbDesc* __restrict__ probs,
                                                  const double* __restrict__ feats,
                                                  double* __restrict__ cosPool)
{
    constexpr int PITCH = KC * 8 + 16;           // bytes per staged row piece (the 16 spread the rows over the banks)
    constexpr int SEGS = KC / 2;                 // 16-byte segments per row piece
    constexpr int RPI = 256 / SEGS;              // rows per load instruction of the workgroup
    constexpr int NLD = 128 / RPI;               // load instructions per stage and thread (first half: A rows, second half: B rows)
    constexpr int STAGE = 128 * PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 stages x (64 A rows + 64 B rows) x PITCH
    // (the grid may be smaller than the number of tiles: the workgroups then loop)
    const int xcd = blockIdx.x & 7, nSlots = G * ((B + 7) >> 3);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int slot = blockIdx.x >> 3; slot < nSlots; slot += gridDim.x >> 3) {
    const int b = (slot / G) * 8 + xcd;          // all tiles of a problem on one XCD (k_cos)
    if (b >= B) continue;
    const ProbDesc pd = probs[b];
    const int ti_n = cos_tiles(pd.n1), tj_n = cos_tiles(pd.n2);
    const int tile = slot % G;
    if (tile >= ti_n * tj_n) continue;
    const int ti = tile / tj_n, tj = tile - ti * tj_n;
    const int nbi = (pd.n1 + 15) >> 4, nbj = (pd.n2 + 15) >> 4;
    const int bi0 = ti * nbi / ti_n, bj0 = tj * nbj / tj_n;                    // first block of the tile
    const int nbx = (ti + 1) * nbi / ti_n - bi0, nby = (tj + 1) * nbj / tj_n - bj0;   // blocks of the tile: 1..4
    const int i0 = bi0 * 16, j0 = bj0 * 16;
    const int iEnd = min(pd.n1, i0 + 16 * nbx), jEnd = min(pd.n2, j0 + 16 * nby);    // rows / columns of the tile
    const int Fc = D.p.cos_feature_dim, coff = D.p.point_dim + D.p.ratio_feature_dim;
    const int lr = lane & 15, kq = lane >> 4;
    // A wave takes ONE block row of the tile (all its nby blocks) or, when the tile has more block columns than rows, one block
    // column: a 3x3 tile keeps three waves busy with 3 blocks each, 3x4 and 4x3 tiles all four — dealt as 2x2 quarters they
    // were 4 + 2 + 2 + 1 and the workgroup as slow as a full tile (tools/ubench/mfma_tiles.hip: 213 us against the 169 of a
    // perfect deal for config 3).  Which wave takes which row rotates with the tile: a wave index is a SIMD.
    // A tile of at most 2x2 blocks (maps of up to 32 objects: the reference's demo scale) gives every wave ONE block.
    const bool oneBlock = nbx <= 2 && nby <= 2;
    const bool byRows = oneBlock || nbx >= nby;
    const int wsr = (w + tile) & 3;
    const int ws = oneBlock ? (wsr >> 1) : wsr;                  // the wave's block row (byRows) or block column
    const int m0 = oneBlock ? (wsr & 1) : 0;                     // first of its blocks across
    const int nMin = uni_i(oneBlock ? ((ws < nbx && m0 < nby) ? 1 : 0)
                                    : (ws < (byRows ? nbx : nby) ? (byRows ? nby : nbx) : 0));      // blocks of this wave: 0..4

    // the thread's share of a stage
    const int seg = tid % SEGS, row0 = tid / SEGS;
    const double* gp[NLD]; bool gv[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int r = (q * RPI + row0) & 63;
        const bool isA = q < NLD / 2;
        const int gr = (isA ? i0 : j0) + r;
        gv[q] = gr < (isA ? iEnd : jEnd);
        gp[q] = feats + ((isA ? pd.off1 : pd.off2) + (gv[q] ? gr : 0)) * D.F + coff + 2 * seg;
    }
    // (no control flow around the loads of the main loop: the compiler's wait-count pass then lets two stages fly)
    auto gload = [&](dbl2_t (&stg)[NLD], int k0) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const d2u_t t = *reinterpret_cast<const d2u_t*>(gp[q] + k0);
            stg[q] = dbl2_t{t.v[0], t.v[1]};
        }
    };
    auto gload_ragged = [&](dbl2_t (&stg)[NLD], int k0) {      // last stage of a descriptor that is no multiple of KC: element by element
        const int k = k0 + 2 * seg;
#pragma unroll
        for (int q = 0; q < NLD; ++q) stg[q] = dbl2_t{k < Fc ? gp[q][k0] : 0.0, k + 1 < Fc ? gp[q][k0 + 1] : 0.0};
    };
    auto lstore = [&](const dbl2_t (&stg)[NLD], int buf) {
#pragma unroll
        for (int q = 0; q < NLD; ++q)                // rows behind the tile: zeros
            *reinterpret_cast<dbl2_t*>(smem + buf * STAGE + (q * RPI + row0) * PITCH + seg * 16) = gv[q] ? stg[q] : dbl2_t{0.0, 0.0};
    };

    double4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = double4_t{0.0, 0.0, 0.0, 0.0};
    double sMaj = 0.0, sMin[4] = {0.0, 0.0, 0.0, 0.0};
    // LDS offsets of the lane's 32-byte pieces: A rows are stage rows 0..63, B rows 64..127
    const int offMaj = ((byRows ? 0 : 64) + 16 * ws + lr) * PITCH + 32 * kq;
    const int offMin = ((byRows ? 64 : 0) + 16 * m0 + lr) * PITCH + 32 * kq;

    auto compute = [&](int s) {
        const unsigned char* base = smem + (s & 1) * STAGE;
        if (nMin > 0) {
#pragma unroll
            for (int kc = 0; kc < KC; kc += 16) {
                if (s * KC + kc < Fc) {
                    // block by block (the four MFMAs of a block depend on each other through the accumulator: forwarded in the
                    // matrix pipe; every accumulator still sees t = 0..3 of chunk after chunk — its own order is unchanged):
                    // 16 operand registers instead of 40, and the reads of the next block fly behind the MFMAs of this one
                    double mj[4];
                    {
                        const dbl2_t v0 = *reinterpret_cast<const dbl2_t*>(base + offMaj + kc * 8);
                        const dbl2_t v1 = *reinterpret_cast<const dbl2_t*>(base + offMaj + kc * 8 + 16);
                        mj[0] = v0.x; mj[1] = v0.y; mj[2] = v1.x; mj[3] = v1.y;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) sMaj = fma(mj[t], mj[t], sMaj);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (m < nMin) {
                            const dbl2_t v0 = *reinterpret_cast<const dbl2_t*>(base + offMin + 16 * m * PITCH + kc * 8);
                            const dbl2_t v1 = *reinterpret_cast<const dbl2_t*>(base + offMin + 16 * m * PITCH + kc * 8 + 16);
                            const double mn[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                sMin[m] = fma(mn[t], mn[t], sMin[m]);
                                // a column wave multiplies B A^T: the TRANSPOSED block — the same products, summed in the same k order
                                acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(mj[t], mn[t], acc[m], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    };

    // stage s is multiplied while the loads of stages s + 1 (register set of the other parity) and s + 2 (this parity's) fly
    // (a third set — three stages in flight — measured 3 % slower at config 3 and at the demo scale)
    // (every load of the loop is UNCONDITIONAL — beyond the last stage it fetches that stage again: a load inside an `if`
    // makes the compiler's wait-count pass assume the shorter queue at the join and wait for everything in flight)
    const int SF = Fc / KC;                          // full stages; a ragged one may follow
    dbl2_t r0[NLD], r1[NLD];
    if (SF > 0) {
        gload(r0, 0);
        gload(r1, min(1, SF - 1) * KC);
        lstore(r0, 0);
        __syncthreads();
        for (int s = 0; s < SF; s += 2) {
            gload(r0, min(s + 2, SF - 1) * KC);
            compute(s);
            lstore(r1, 1);
            __syncthreads();
            gload(r1, min(s + 3, SF - 1) * KC);
            if (s + 1 < SF) compute(s + 1);
            lstore(r0, 0);
            __syncthreads();
        }
    }
    if (SF * KC < Fc) {                              // (every wave is behind the last barrier: both LDS stages are free)
        gload_ragged(r0, SF * KC);
        lstore(r0, SF & 1);
        __syncthreads();
        compute(SF);
    }
    // every lane (lr, *) ends with the norm of row lr of the block
    sMaj += __shfl_xor(sMaj, 16); sMaj += __shfl_xor(sMaj, 32); sMaj = sqrt(sMaj);
#pragma unroll
    for (int m = 0; m < 4; ++m) { sMin[m] += __shfl_xor(sMin[m], 16); sMin[m] += __shfl_xor(sMin[m], 32); sMin[m] = sqrt(sMin[m]); }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m < nMin) {                              // (the divisions run on the same f64 units as the MFMAs)
            // acc[m][r] of lane (lr, kq) is element (kq + 4r, lr) of (own block) x (block m across)^T
            const double nLr = sMin[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double nOwn = __shfl(sMaj, kq + 4 * r);      // norm of row (kq + 4r) of the own block: held by lanes with lr == kq + 4r
                const int row = byRows ? i0 + 16 * ws + kq + 4 * r : i0 + 16 * (m0 + m) + lr;
                const int col = byRows ? j0 + 16 * (m0 + m) + lr : j0 + 16 * ws + kq + 4 * r;
                const double na = byRows ? nOwn : nLr, nb = byRows ? nLr : nOwn;
                if (row < iEnd && col < jEnd)
                    cosPool[pd.cosOff + (int64_t)row * pd.n2 + col] = D.pruned ? acc[m][r] : ((na > 0.0 && nb > 0.0) ? acc[m][r] / (na * nb) : 0.0);
            }
        }
    }
    __syncthreads();                                 // the next tile's first stage overwrites LDS stage 0
    }
}

// ---------------------------------------------------------------------------------------------
// k_cos_deal<T>: the cosine matrices of a BATCH of mid-size maps (config 3: 256 pairs of 200 x 200 objects).  k_cos_tile's tiles
// are at most 4 x 4 blocks and a wave owns one block row of its tile: at 13 blocks per dimension (200 objects) the balanced
// tiling is 3 + 3 + 3 + 4, nine of the sixteen tiles are 3 x 3 and leave one wave of four without a block — the matrix pipe
// cannot be busier than 169 / 256 of the time (measured: 0.5).  Here a tile is up to T x T blocks (T = 5; 13 blocks: 5 + 4 + 4)
// and its BLOCKS are dealt to the four waves as equal runs of the row-major block list — 25, 20, 16 blocks: 6-7, 5, 4 per wave,
// 169 of 172 block slots used —; a wave holds up to 7 accumulators, reads the operands of its next block from LDS in front of
// the MFMAs of the current one, and a staged row feeds 4-5 blocks instead of 3-4.  LDS: two stages of (80 + 80) row pieces of
// 16 doubles = 46 KB, 153 registers: three workgroups per compute unit.
// Measured alone on config 3's shape (tools/ubench/cos_time.hip, one box): k_cos_tile 361 us, T = 7 (two workgroups per CU, 13
// accumulators) 344, T = 6 362, T = 5 321, T = 4 365; in the batch pipeline (rocprofv3) 322 -> 290 us.  The same tool strips parts
// of the kernel: without global loads, LDS stores, barriers and norms the T = 5 loop takes 208 us (the matrix pipe's own time for
// this deal), and every part that is put back costs 20-80 us, sub-additively — because the kernel is POWER-bound: a probe wave
// per XCD (clock64 against wall_clock64) sees the shader clock fall from 2.4 GHz (idle) to 2.1-2.2 GHz under the stripped loop
// and to 1.7-2.0 GHz under the complete kernel (k_cos_tile alike).  What makes this kernel faster is less work per MFMA.
// The launcher picks it when the batch has at least two workgroups per compute unit of such tiles; a single alignment keeps
// k_cos_tile's sixteen small tiles (latency).
// Bits: a block's MFMA sequence is k_cos's own (chunks of 16 ascending, MFMA t of a chunk contracts k = k0 + 4 (lane >> 4) + t);
// the norms are summed by a separate pass over the staged rows in the order of k_cos_tile's (lane (lr, kq): its k's ascending,
// then the two cross-quarter adds) and handed over through LDS.  Identical bits.
// ---------------------------------------------------------------------------------------------
constexpr int COSD_PITCH = 16 * 8 + 16;          // bytes per staged row piece (k_cos_tile<16>'s)
template <int T> struct CosDeal {                // T: blocks per tile and dimension
    static constexpr int ROWS = 2 * 16 * T;      // staged row pieces: A rows 0..16T-1, B rows 16T..32T-1
    static constexpr int STAGE = ROWS * COSD_PITCH;
    static constexpr int MAXB = (T * T + 3) / 4; // blocks of a wave
    static constexpr int LDS = 2 * STAGE + ROWS * 8;                    // two stages + the norms
    static constexpr int WAVES = T <= 5 ? 3 : 2; // waves per SIMD the registers are capped for (workgroups per compute unit)
    __device__ __host__ static int tiles(int n) { return (((n + 15) >> 4) + T - 1) / T; }
};

template <int T, int DBG /* 0; tools/ubench/cos_time.hip strips parts of the kernel to time the rest: 1 no loads in the loop, 2 no LDS stores / barriers, 4 no MFMAs, 8 no norms */>
__global__ void __launch_bounds__(256, CosDeal<T>::WAVES) k_cos_deal(DevParams D, int B, int G /* workgroups (tiles) per problem */,
                                                     const ProbDesc* __restrict__ probs,
                                                     const double* __restrict__ feats,
                                                     double* __restrict__ cosPool,
                                                     const int32_t* __restrict__ only /* NULL, or [B]: problems with a 0 are skipped (behind k_cos_sel) */)
{
    using CD = CosDeal<T>;
    constexpr int KC = 16, PITCH = COSD_PITCH, SEGS = KC / 2, RPI = 256 / SEGS, NLD = CD::ROWS / RPI, STAGE = CD::STAGE, MAXB = CD::MAXB;
    static_assert(NLD * RPI == CD::ROWS, "the workgroup's loads cover a stage exactly");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* nrm = reinterpret_cast<double*>(smem + 2 * STAGE);          // norm of staged row r
    // eight problems or more: all tiles of a problem on one XCD (they share its rows); fewer (grid = B * G): tiles over all XCDs
    const int xcd = blockIdx.x & 7, slot = B >= 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = uni_i(tid >> 6);
    const int b = B >= 8 ? (slot / G) * 8 + xcd : slot / G;
    if (b >= B) return;
    if (only && !only[b]) return;
    const ProbDesc pd = probs[b];
    const int ti_n = CD::tiles(pd.n1), tj_n = CD::tiles(pd.n2);
    const int tile = slot % G;
    if (tile >= ti_n * tj_n) return;
    const int ti = tile / tj_n, tj = tile - ti * tj_n;
    const int nbi = (pd.n1 + 15) >> 4, nbj = (pd.n2 + 15) >> 4;
    const int bi0 = ti * nbi / ti_n, bj0 = tj * nbj / tj_n;                    // first block of the tile (balanced cut)
    const int nbx = uni_i((ti + 1) * nbi / ti_n - bi0), nby = uni_i((tj + 1) * nbj / tj_n - bj0);   // blocks of the tile: 1..T
    const int i0 = bi0 * 16, j0 = bj0 * 16;
    const int iEnd = min(pd.n1, i0 + 16 * nbx), jEnd = min(pd.n2, j0 + 16 * nby);
    const int Fc = D.p.cos_feature_dim, coff = D.p.point_dim + D.p.ratio_feature_dim;
    const int lr = lane & 15, kq = lane >> 4;
    // the wave's run of the tile's row-major block list (which wave takes which run rotates with the tile: a wave index is a SIMD)
    const int wsr = (w + tile) & 3, nblk = nbx * nby;
    const int e0 = uni_i(wsr * nblk / 4), cnt = uni_i((wsr + 1) * nblk / 4 - e0);        // <= MAXB
    const int lanePart = lr * PITCH + 32 * kq;
    int offA[MAXB], offB[MAXB]; bool newRow[MAXB];
#pragma unroll
    for (int m = 0; m < MAXB; ++m) {
        const int e = min(e0 + m, nblk - 1), bx = e / nby, by = e - bx * nby;
        offA[m] = uni_i(16 * bx * PITCH); offB[m] = uni_i((16 * T + 16 * by) * PITCH);
        newRow[m] = uni_i((by == 0 && e0 + m < nblk) ? 1 : 0) != 0;
    }

    // the thread's share of a stage
    const int seg = tid % SEGS, row0 = tid / SEGS;
    const double* gp[NLD]; bool gv[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int r = q * RPI + row0;
        const bool isA = r < 16 * T;
        const int gr = isA ? i0 + r : j0 + r - 16 * T;
        gv[q] = gr < (isA ? iEnd : jEnd);
        gp[q] = feats + ((isA ? pd.off1 : pd.off2) + (gv[q] ? gr : 0)) * D.F + coff + 2 * seg;
    }
    // (no control flow around the loads of the main loop — k_cos_tile)
    auto gload = [&](dbl2_t (&stg)[NLD], int k0) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const d2u_t t = *reinterpret_cast<const d2u_t*>(gp[q] + k0);
            stg[q] = dbl2_t{t.v[0], t.v[1]};
        }
    };
    auto gload_ragged = [&](dbl2_t (&stg)[NLD], int k0) {      // last stage of a descriptor that is no multiple of 16: element by element
        const int k = k0 + 2 * seg;
#pragma unroll
        for (int q = 0; q < NLD; ++q) stg[q] = dbl2_t{k < Fc ? gp[q][k0] : 0.0, k + 1 < Fc ? gp[q][k0 + 1] : 0.0};
    };
    auto lstore = [&](const dbl2_t (&stg)[NLD], int buf) {
#pragma unroll
        for (int q = 0; q < NLD; ++q)                // rows behind the tile: zeros
            *reinterpret_cast<dbl2_t*>(smem + buf * STAGE + (q * RPI + row0) * PITCH + seg * 16) = gv[q] ? stg[q] : dbl2_t{0.0, 0.0};
    };

    double4_t acc[MAXB];
#pragma unroll
    for (int m = 0; m < MAXB; ++m) acc[m] = double4_t{0.0, 0.0, 0.0, 0.0};
    // norms: wave w sums the 16-row blocks w, w + 4, w + 8, ... of the stage's 2 T (those that hold rows of the tile)
    constexpr int NNB = (2 * T + 3) / 4;
    double sN[NNB]; bool nOn[NNB];
#pragma unroll
    for (int h = 0; h < NNB; ++h) {
        const int hb = w + 4 * h;
        sN[h] = 0.0;
        nOn[h] = uni_i((hb < T ? hb < nbx : (hb < 2 * T && hb - T < nby)) ? 1 : 0) != 0;
    }

    // Operands of block m + 1 are read from LDS BEFORE the four MFMAs of block m are issued (two register sets): the wave never
    // waits for LDS with an idle matrix pipe behind it.  The read behind the last block fetches that block again (offsets clamped).
    // The norms' f64 FMAs (the units the f64 MFMAs occupy) come as one burst behind the stage's MFMAs: measured equal to spreading
    // them between the blocks with their operands read ahead (tools/ubench/cos_time.hip).  A norm's own order — its lane's k
    // ascending — is k_cos_tile's.
    auto norm_sum = [&](int h, const dbl2_t& v0, const dbl2_t& v1) {
        sN[h] = fma(v0.x, v0.x, sN[h]); sN[h] = fma(v0.y, v0.y, sN[h]); sN[h] = fma(v1.x, v1.x, sN[h]); sN[h] = fma(v1.y, v1.y, sN[h]);
    };
    auto compute = [&](int s) {
        const unsigned char* base = smem + (s & 1) * STAGE + lanePart;
        dbl2_t a0, a1, b0, b1;
        a0 = *reinterpret_cast<const dbl2_t*>(base + offA[0]); a1 = *reinterpret_cast<const dbl2_t*>(base + offA[0] + 16);
        b0 = *reinterpret_cast<const dbl2_t*>(base + offB[0]); b1 = *reinterpret_cast<const dbl2_t*>(base + offB[0] + 16);
#pragma unroll
        for (int m = 0; m < MAXB; ++m) {
            if (m < cnt) {
                dbl2_t na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
                if (m + 1 < MAXB) {
                    nb0 = *reinterpret_cast<const dbl2_t*>(base + offB[m + 1]); nb1 = *reinterpret_cast<const dbl2_t*>(base + offB[m + 1] + 16);
                    if (newRow[m + 1]) {                 // the next block starts a block row: its A operand
                        na0 = *reinterpret_cast<const dbl2_t*>(base + offA[m + 1]); na1 = *reinterpret_cast<const dbl2_t*>(base + offA[m + 1] + 16);
                    }
                }
                if (!(DBG & 4)) {
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc[m], 0, 0, 0);
                } else { acc[m][0] += a0.x + b0.x; acc[m][1] += a0.y + b0.y; acc[m][2] += a1.x + b1.x; acc[m][3] += a1.y + b1.y; }
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        }
        if (!(DBG & 8)) {
#pragma unroll
            for (int h = 0; h < NNB; ++h)
                if (nOn[h]) {
                    const dbl2_t v0 = *reinterpret_cast<const dbl2_t*>(base + 16 * (w + 4 * h) * PITCH);
                    const dbl2_t v1 = *reinterpret_cast<const dbl2_t*>(base + 16 * (w + 4 * h) * PITCH + 16);
                    norm_sum(h, v0, v1);
                }
        }
    };

    // stage s is multiplied while the loads of stages s + 1 and s + 2 fly (k_cos_tile's loop)
    const int SF = Fc / KC;
    dbl2_t r0[NLD], r1[NLD];
    if (SF > 0) {
        gload(r0, 0);
        gload(r1, min(1, SF - 1) * KC);
        lstore(r0, 0);
        __syncthreads();
        for (int s = 0; s < SF; s += 2) {
            if (!(DBG & 1)) gload(r0, min(s + 2, SF - 1) * KC);
            compute(s);
            if (!(DBG & 2)) { lstore(r1, 1); __syncthreads(); }
            if (!(DBG & 1)) gload(r1, min(s + 3, SF - 1) * KC);
            if (s + 1 < SF) compute(s + 1);
            if (!(DBG & 2)) { lstore(r0, 0); __syncthreads(); }
        }
    }
    if (SF * KC < Fc) {                              // (every wave is behind the last barrier: both LDS stages are free)
        gload_ragged(r0, SF * KC);
        lstore(r0, SF & 1);
        __syncthreads();
        compute(SF);
    }
#pragma unroll
    for (int h = 0; h < NNB; ++h) {
        double v = sN[h];
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (nOn[h] && kq == 0) nrm[16 * (w + 4 * h) + lr] = sqrt(v);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MAXB; ++m) {
        if (m < cnt) {
            // acc[m][r] of lane (lr, kq) is element (kq + 4r, lr) of (block row bx) x (block column by)^T
            const int e = e0 + m, bx = e / nby, by = e - bx * nby;
            const int col = j0 + 16 * by + lr;
            const double nb = nrm[16 * T + 16 * by + lr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 16 * bx + kq + 4 * r;
                const double na = nrm[16 * bx + kq + 4 * r];
                if (row < iEnd && col < jEnd)
                    cosPool[pd.cosOff + (int64_t)row * pd.n2 + col] = D.pruned ? acc[m][r] : ((na > 0.0 && nb > 0.0) ? acc[m][r] / (na * nb) : 0.0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_cos_sel (round 6): the cosine matrices of a batch whose cosines are only used behind the gate cos > cosine_min
// (single_score(): an association with (cos - cosine_min) / (cosine_max - cosine_min) <= 0 scores 0 whatever its cosine is).
// k_cos_deal computes all n1 x n2 cosines in f64 on the matrix core and is POWER-bound (295 us at config 3); ~95 % of its
// products are thrown away by the gate.  Here one workgroup of 16 waves takes a problem:
//   pass 1  approximate cosines of all pairs: rows converted to bf16 on the way into LDS, v_mfma_f32_16x16x32_bf16 (1/32 of the
//           f64 MFMA's matrix-pipe time), divided by the EXACT f64 norms (summed here in the oracle's stated order: four chains,
//           (s0 + s1) + (s2 + s3));
//   select  pair (i, j) is a CANDIDATE unless approx < cosine_min - delta.  |approx - cos| <= 2^-8 (two bf16 roundings of 2^-9
//           each, Cauchy-Schwarz over the contraction) + 512 * 2^-22 (f32 accumulation, any order, truncating or not) < 0.0041
//           for rows whose norm lies in [2^-40, 2^40] (no element over- or underflows in bf16 / f32 beyond 2^-86 of the product of
//           the norms); rows with any other non-zero norm (huge, tiny, inf, NaN) make all their pairs candidates.  delta = 2^-6.
//   pass 2  the candidates' dot products in f64 with k_cos_deal's own contraction order (oracle dot_fixed(): per chunk of 16, t = 0..3
//           outer, g = 0..3 inner, k = k0 + 4 g + t, ONE fma chain), rows staged through LDS chunk by chunk, candidates sorted by i so
//           that the lanes of a wave read few distinct a rows (LDS broadcast); cos = dot / (na * nb) as everywhere.
// A non-candidate's entry of the pool holds the APPROXIMATE cosine (< cosine_min - delta + 0.0041 < cosine_min, and so is the exact one:
// both fail the gate alike).  Candidates' entries are bit-identical to k_cos_deal's.  A problem with more than CSEL_CAP candidates (or maps
// of more than 256 objects) is flagged in dense[] and left to k_cos_deal<., ., true>, which the launcher runs behind this kernel for the
// flagged problems only.  Never used for roman_debug_cosine / the pruned prefilter (raw products) / cosine_max <= cosine_min.
// ---------------------------------------------------------------------------------------------
constexpr int CSEL_ROWS = 512;                   // staged rows: A rows [0, 16 nb1), B rows [16 nb1, 16 nb1 + 16 nb2)
constexpr int CSEL_MAXN = 256;                   // objects per map
constexpr int CSEL_CAP = 4096;                   // candidates per problem (config 3: ~2200 of 40000); four per thread in pass 2
constexpr int CSEL_NPT = CSEL_CAP / 1024;        // candidates per thread
constexpr int CSEL_P1 = 256 + 16;                // bytes per staged row of pass 1: 128 bf16 + pad
constexpr int CSEL_P2 = 128 + 16;                // bytes per staged row of pass 2: 16 doubles + pad
constexpr int CSEL_MP = 11;                    // blocks of the product per wave: problems of up to 176 blocks of 16 x 16 (200 x 200 objects: 169)
constexpr int CSEL_STAGE = CSEL_ROWS * CSEL_P1;  // pass 1: the tile; behind it: pass 2's chunk and the candidate lists
static_assert(CSEL_ROWS * CSEL_P2 + 2 * CSEL_CAP * 4 <= CSEL_STAGE, "pass 2 fits where the tile was");
constexpr int CSEL_LDS = CSEL_STAGE + CSEL_ROWS * 8 + 2 * 260 * 4;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(8))) d8u_t { double v[8]; };     // 8-byte aligned 64-byte load

__device__ __forceinline__ bool csel_unsafe(double n) { return !(n >= 0x1p-40 && n <= 0x1p40); }

__global__ void __launch_bounds__(1024) k_cos_sel(DevParams D, int B, const ProbDesc* __restrict__ probs, const double* __restrict__ feats,
                                                  double* __restrict__ cosPool, int32_t* __restrict__ dense /* [B] out: 1 = left to the dense kernel */,
                                                  double thr /* cosine_min - delta; +inf: no candidates (the approximate matrix, tests) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* stage = smem;
    double* nrm = reinterpret_cast<double*>(smem + CSEL_STAGE);
    uint32_t* U = reinterpret_cast<uint32_t*>(smem + CSEL_ROWS * CSEL_P2);   // candidates as found: i | j << 16 (where the tile was: behind pass 2's chunk)
    uint32_t* S = U + CSEL_CAP;                                          // sorted by i
    int* hist = reinterpret_cast<int*>(nrm + CSEL_ROWS);                 // [0, 256): candidates of row i; [256]: all
    int* cur = hist + 260;
    const int b = blockIdx.x;
    if (b >= B) return;
    const ProbDesc pd = probs[b];
    const int tid = threadIdx.x, lane = tid & 63, w = uni_i(tid >> 6);
    const int n1 = pd.n1, n2 = pd.n2;
    if (n1 <= 0 || n2 <= 0) { if (tid == 0) dense[b] = 0; return; }
    const int nb1 = (n1 + 15) >> 4, nb2 = (n2 + 15) >> 4, RB = 16 * nb1, nblk = nb1 * nb2, rowsTot = RB + 16 * nb2;
    if (n1 > CSEL_MAXN || n2 > CSEL_MAXN || nblk > 16 * CSEL_MP) { if (tid == 0) dense[b] = 1; return; }
    const int Fc = D.p.cos_feature_dim, coff = D.p.point_dim + D.p.ratio_feature_dim;
    if (tid < 260) hist[tid] = 0;
    const double* base1 = feats + pd.off1 * D.F + coff;
    const double* base2 = feats + pd.off2 * D.F + coff;
    // staged row R holds object R of map 1 (R < RB) or object R - RB of map 2; rows behind a map's last object: zeros

    // ---- pass 1 -------------------------------------------------------------------------------------------
    // loads: one wave instruction reads 128 consecutive elements — ONE KILOBYTE — of one row: lane l the elements k0 + 2 l, k0 + 2 l + 1.
    // (Pieces of 128 / 256 B of eight / four rows per instruction — what a tile of 16 / 32 elements of every row needs — ran this pass
    // at 127 / 115 us whatever the prefetch depth: 256 workgroups x 416 rows are 10^5 streams of short bursts for the DRAM.)  A tile is
    // 128 elements of every row as bf16 (rows x 272 B); wave w owns the rows w, w + 16, ...; four rows per register set, two sets, each
    // re-loaded as soon as it is staged — also across the tile's barrier and its MFMAs.
    // The screen's norms are the bf16 rows' own (the diagonals of the products of every block of 16 rows with itself: wave w the
    // blocks w, w + 16 — no f64 arithmetic in this pass): cos(a^, b^) against cos(a, b) — two angles of at most asin(2^-9).
    constexpr int KT = 128, G1 = 4;
    const int nt = rowsTot >> 4;                                         // rows of a wave
    // No branch and no address arithmetic inside the stream of loads (a branch makes the compiler wait for ALL outstanding loads at its
    // join; the rows' addresses computed per load were 5800 scalar instructions per wave — the scalar unit busy 40 % of the pass): lane t
    // of the wave holds the address of the wave's row slot t — a slot behind the wave's last row, or a row behind its map's last object,
    // holds the last real one's (a cache hit; staged into a tile row whose products nobody reads: C[i][j] depends on rows i and j alone)
    // — and a slot's load takes it from there (v_readlane) as its scalar base.
    const int NG8 = ((nt + 2 * G1 - 1) / (2 * G1)) * 2;                  // groups of a tile, an even number
    const int NSF = Fc / KT;                                             // full tiles
    uint32_t slotLo, slotHi;
    {
        const int R = w + 16 * min(lane, nt - 1);
        const bool isA = R < RB;
        const int idx = min(isA ? R : R - RB, (isA ? n1 : n2) - 1);
        const uint64_t pa = reinterpret_cast<uint64_t>((isA ? base1 : base2) + (int64_t)idx * D.F);
        slotLo = (uint32_t)pa; slotHi = (uint32_t)(pa >> 32);
    }
    auto slot_ptr = [&](int t) -> const double* {                        // (wave-uniform)
        const uint64_t lo = (uint32_t)__builtin_amdgcn_readlane((int)slotLo, t), hi = (uint32_t)__builtin_amdgcn_readlane((int)slotHi, t);
        return reinterpret_cast<const double*>((hi << 32) | lo);
    };
    auto load_group = [&](dbl2_t (&x)[G1], int st, int g) {
        const int ko = st * KT + 2 * lane;
#pragma unroll
        for (int j = 0; j < G1; ++j) {
            const d2u_t tv = *reinterpret_cast<const d2u_t*>(slot_ptr(G1 * g + j) + ko);
            x[j] = dbl2_t{tv.v[0], tv.v[1]};
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_group_ragged = [&](dbl2_t (&x)[G1], int st, int g) {
        const int k = st * KT + 2 * lane;
#pragma unroll
        for (int j = 0; j < G1; ++j) {
            const double* rp = slot_ptr(G1 * g + j);
            x[j] = dbl2_t{k < Fc ? rp[k] : 0.0, k + 1 < Fc ? rp[k + 1] : 0.0};
        }
    };
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    auto stage_group = [&](const dbl2_t (&x)[G1], int g) {
#pragma unroll
        for (int j = 0; j < G1; ++j) {                                   // (slot t < 32: a row of the tile's 512 whatever nt is)
            const bf16x2_t h2 = {(__bf16)(float)x[j].x, (__bf16)(float)x[j].y};
            *reinterpret_cast<bf16x2_t*>(stage + (w + 16 * (G1 * g + j)) * CSEL_P1 + 4 * lane) = h2;
        }
    };
    // wave w multiplies a run of the product's row-major block list (the A operand is read again only where a block row starts)
    const int e0 = uni_i(w * nblk / 16), cntP = uni_i((w + 1) * nblk / 16 - e0);         // <= CSEL_MP
    f32x4_t acc[CSEL_MP], accD[2];
#pragma unroll
    for (int m = 0; m < CSEL_MP; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    accD[0] = accD[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int offA[CSEL_MP], offB[CSEL_MP], offD[2]; bool newRow[CSEL_MP];
#pragma unroll
    for (int m = 0; m < CSEL_MP; ++m) {
        const int e = min(e0 + m, e0 + cntP - 1), bx = e / nb2, by = e - bx * nb2;
        offA[m] = uni_i(16 * bx * CSEL_P1); offB[m] = uni_i((RB + 16 * by) * CSEL_P1);
        newRow[m] = uni_i((m == 0 || by == 0) ? 1 : 0) != 0;
    }
    offD[0] = uni_i(16 * min(w, nt - 1) * CSEL_P1); offD[1] = uni_i(16 * min(w + 16, nt - 1) * CSEL_P1);
    const int lanePart = (lane & 15) * CSEL_P1 + (lane >> 4) * 16;
    auto multiply = [&]() {
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const unsigned char* tb = stage + lanePart + 64 * sub;
            bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(tb + offA[0]);
#pragma unroll
            for (int m = 0; m < CSEL_MP; ++m)
                if (m < cntP) {
                    if (m > 0 && newRow[m]) a = *reinterpret_cast<const bf16x8_t*>(tb + offA[m]);
                    const bf16x8_t bq = *reinterpret_cast<const bf16x8_t*>(tb + offB[m]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[m], 0, 0, 0);
                }
#pragma unroll
            for (int m = 0; m < 2; ++m) {            // (staged rows: map 1's blocks, then map 2's)
                const bf16x8_t dg = *reinterpret_cast<const bf16x8_t*>(tb + offD[m]);
                accD[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dg, dg, accD[m], 0, 0, 0);
            }
        }
    };
    {
        dbl2_t xa[G1], xb[G1];
        if (NSF > 0) { load_group(xa, 0, 0); load_group(xb, 0, 1); }
        for (int st = 0; st < NSF; ++st) {
            for (int g = 0; g < NG8; g += 2) {
                // (the groups two ahead: the next tile's first two behind this tile's last — the last tile loads its own first two again)
                const int gn = g + 2 < NG8 ? g + 2 : 0, stn = g + 2 < NG8 ? st : min(st + 1, NSF - 1);
                stage_group(xa, g);
                load_group(xa, stn, gn);
                stage_group(xb, g + 1);
                load_group(xb, stn, gn + 1);
            }
            __syncthreads();
            multiply();
            __syncthreads();
        }
        if (NSF * KT < Fc) {                         // a ragged last tile: zeros behind the descriptor's end
            for (int g = 0; g < NG8; ++g) { load_group_ragged(xa, NSF, g); stage_group(xa, g); }
            __syncthreads();
            multiply();
            __syncthreads();
        }
    }
    // norms of the bf16 rows: element (r, r) of a diagonal block is register (lane & 3) of the lane with (lane & 15) >> 2 == lane >> 4
#pragma unroll
    for (int m = 0; m < 2; ++m)
        if (w + 16 * m < nt) {
            const int r4 = lane & 3;
            const float v = r4 == 0 ? accD[m][0] : r4 == 1 ? accD[m][1] : r4 == 2 ? accD[m][2] : accD[m][3];
            if (((lane & 15) >> 2) == (lane >> 4)) nrm[16 * (w + 16 * m) + (lane & 15)] = sqrt((double)v);
        }
    __syncthreads();

    // ---- select -------------------------------------------------------------------------------------------
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int m = 0; m < CSEL_MP; ++m)
        if (m < cntP) {
            const int e = e0 + m, bx = e / nb2, by = e - bx * nb2;
            const int j = 16 * by + (lane & 15);
            const double nbv = nrm[RB + j];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * bx + 4 * (lane >> 4) + rr;
                const double na = nrm[i];
                const bool ok = i < n1 && j < n2;
                // (a row whose f32 norm is zero, tiny, huge or no number is left to the exact pass altogether: its guard for zero norms too)
                const double cv = (double)acc[m][rr] / (na * nbv);
                const bool cand = ok && (csel_unsafe(na) || csel_unsafe(nbv) || !(cv < thr));
                if (ok && !cand) cosPool[pd.cosOff + (int64_t)i * n2 + j] = cv;
                const unsigned long long bm = __ballot(cand);
                if (bm) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&hist[256], __popcll(bm));
                    base = __shfl(base, 0);
                    if (cand) {
                        const int slot = base + __popcll(bm & lt);
                        if (slot < CSEL_CAP) U[slot] = (uint32_t)i | ((uint32_t)j << 16);
                        atomicAdd(&hist[i], 1);
                    }
                }
            }
        }
    __syncthreads();
    const int nc = hist[256];
    if (nc > CSEL_CAP) { if (tid == 0) dense[b] = 1; return; }
    if (tid == 0) dense[b] = 0;
    if (nc == 0) return;
    if (w == 0) {                                // cursors: exclusive prefix of the 256 row counts
        const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        int incl = c0 + c1 + c2 + c3;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const int ex = incl - (c0 + c1 + c2 + c3);
        cur[4 * lane] = ex; cur[4 * lane + 1] = ex + c0; cur[4 * lane + 2] = ex + c0 + c1; cur[4 * lane + 3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    for (int q = tid; q < nc; q += 1024) { const uint32_t v = U[q]; S[atomicAdd(&cur[v & 0xffffu], 1)] = v; }
    __syncthreads();

    // ---- pass 2 -------------------------------------------------------------------------------------------
    // The rows come again, chunk by chunk, as f64 (the same lanes, the same two register sets in flight).  The exact norms (the oracle's
    // stated order) are summed from the staged chunks: thread (r, h) takes chains 2 h and 2 h + 1 of row r.
    // (no branch inside the stream of loads, as in pass 1: every lane loads in each of the NQ rounds — behind the staged rows, or behind
    // its map's last object, the last real row again: a staged row that no candidate names)
    constexpr int NQ = CSEL_ROWS / 128;
    const int pp = lane & 7, laneRow = 8 * w + (lane >> 3);
    // (the launcher guarantees a feature pool of less than 4 GB: a lane's row is a 32-bit byte offset from the pool's start)
    uint32_t rofs[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int R = min(128 * q + laneRow, rowsTot - 1);
        const bool isA = R < RB;
        const int idx = min(isA ? R : R - RB, (isA ? n1 : n2) - 1);
        rofs[q] = (uint32_t)((((isA ? pd.off1 : pd.off2) + idx) * D.F + coff + 2 * pp) * 8);
    }
    const char* pool = reinterpret_cast<const char*>(feats);
    auto load16 = [&](dbl2_t (&x)[NQ], int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const d2u_t t = *reinterpret_cast<const d2u_t*>(pool + (rofs[q] + (uint32_t)(8 * k0))); x[q] = dbl2_t{t.v[0], t.v[1]}; }
    };
    auto load16_ragged = [&](dbl2_t (&x)[NQ], int k0) {
        const int k = k0 + 2 * pp;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const double* rb = reinterpret_cast<const double*>(pool + rofs[q]); x[q] = dbl2_t{k < Fc ? rb[k0] : 0.0, k + 1 < Fc ? rb[k0 + 1] : 0.0}; }
    };
    uint32_t my[CSEL_NPT]; double dot[CSEL_NPT];
    const int cnt2 = nc > tid ? (nc - tid + 1023) >> 10 : 0;
#pragma unroll
    for (int m = 0; m < CSEL_NPT; ++m) { my[m] = m < cnt2 ? S[m * 1024 + tid] : 0u; dot[m] = 0.0; }
    const int rN = tid >> 1, hN = tid & 1;
    double s0 = 0.0, s1 = 0.0;
    auto put16 = [&](const dbl2_t (&x)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            *reinterpret_cast<dbl2_t*>(stage + (128 * q + laneRow) * CSEL_P2 + 16 * pp) = x[q];
    };
    auto compute = [&](int s, auto fullChunk) {
        {   // (zeros behind a ragged descriptor's end: fma(0, 0, s) = s for a sum of squares)
            const dbl2_t* pr = reinterpret_cast<const dbl2_t*>(stage + rN * CSEL_P2 + 64 * hN);
            const dbl2_t e0_ = pr[0], e1_ = pr[1], e2_ = pr[2], e3_ = pr[3];
            s0 = fma(e0_.x, e0_.x, s0); s0 = fma(e0_.y, e0_.y, s0); s0 = fma(e1_.x, e1_.x, s0); s0 = fma(e1_.y, e1_.y, s0);
            s1 = fma(e2_.x, e2_.x, s1); s1 = fma(e2_.y, e2_.y, s1); s1 = fma(e3_.x, e3_.x, s1); s1 = fma(e3_.y, e3_.y, s1);
        }
        const int kleft = uni_i(Fc - 16 * s);    // elements of this chunk: >= 16 except the last of a ragged descriptor
#pragma unroll
        for (int m = 0; m < CSEL_NPT; ++m)
            if (m < cnt2) {
                const dbl2_t* pa = reinterpret_cast<const dbl2_t*>(stage + (my[m] & 0xffffu) * CSEL_P2);
                const dbl2_t* pb = reinterpret_cast<const dbl2_t*>(stage + (RB + (my[m] >> 16)) * CSEL_P2);
                double d = dot[m];
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {             // t = 2 tp, 2 tp + 1: the pairs (4 g + 2 tp, 4 g + 2 tp + 1), g = 0..3
                    dbl2_t av[4], bv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) { av[g] = pa[2 * g + tp]; bv[g] = pb[2 * g + tp]; }
                    if (decltype(fullChunk)::value) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) d = fma(av[g].x, bv[g].x, d);
#pragma unroll
                        for (int g = 0; g < 4; ++g) d = fma(av[g].y, bv[g].y, d);
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) if (4 * g + 2 * tp < kleft) d = fma(av[g].x, bv[g].x, d);
#pragma unroll
                        for (int g = 0; g < 4; ++g) if (4 * g + 2 * tp + 1 < kleft) d = fma(av[g].y, bv[g].y, d);
                    }
                }
                dot[m] = d;
            }
    };
    {
        const int NCF = Fc >> 4;                     // full chunks
        dbl2_t ya[NQ], yb[NQ];
        if (NCF > 0) { load16(ya, 0); load16(yb, 16 * min(1, NCF - 1)); }
        for (int s = 0; s < NCF; s += 2) {           // (two ahead; behind the last chunk that one again)
            put16(ya);
            __syncthreads();
            load16(ya, 16 * min(s + 2, NCF - 1));
            compute(s, std::true_type{});
            __syncthreads();
            if (s + 1 < NCF) {
                put16(yb);
                __syncthreads();
                load16(yb, 16 * min(s + 3, NCF - 1));
                compute(s + 1, std::true_type{});
                __syncthreads();
            }
        }
        if (NCF * 16 < Fc) {
            load16_ragged(ya, 16 * NCF);
            put16(ya);
            __syncthreads();
            compute(NCF, std::false_type{});
            __syncthreads();
        }
    }
    {
        const double p = s0 + s1, q = __shfl_xor(p, 1);
        if (hN == 0) nrm[rN] = sqrt(p + q);      // (s0 + s1) + (s2 + s3)
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < CSEL_NPT; ++m)
        if (m < cnt2) {
            const int i = (int)(my[m] & 0xffffu), j = (int)(my[m] >> 16);
            const double na = nrm[i], nbv = nrm[RB + j];
            cosPool[pd.cosOff + (int64_t)i * n2 + j] = (na > 0.0 && nbv > 0.0) ? dot[m] / (na * nbv) : 0.0;
        }
}

// ---------------------------------------------------------------------------------------------
// k_tables: TA[i][i'] (n1 x n1) and TB[j][j'] (n2 x n2).  Entry = horizontal distance (gravity)
// or full distance (otherwise) between two objects of the same map; NaN when the two objects
// coincide (distinctness) or are closer than mindist — every comparison against NaN is false, so
// one table lookup implements the distinctness skip, the mindist gate and the distance itself.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_tables(DevParams D, const ProbDesc* __restrict__ probs,
                                                 const double* __restrict__ feats,
                                                 double* __restrict__ tabPool, int RB /* rows per band */,
                                                 uint16_t* __restrict__ qtabPool /* the same entries as 15-bit bins of width epsilon / 32 (k_count's prefilter), or NULL */)
{
    // grid: (row bands of RB rows over max(n1,n2), 2 maps, B).  The map's points are staged in LDS once per
    // block (3 doubles per object: n gathers from rows 8 F bytes apart — the band is as tall as the batch allows, one band
    // per map when there are enough problems to fill the device); a wave writes table rows with lanes along the row (coalesced).
    extern __shared__ __attribute__((aligned(16))) double s_xyz[];
    const ProbDesc pd = probs[blockIdx.z];
    const int which = blockIdx.y;
    const int n = which == 0 ? pd.n1 : pd.n2;
    const int r0 = blockIdx.x * RB;
    if (r0 >= n) return;
    const int64_t base = which == 0 ? pd.off1 : pd.off2;
    const int pdim = D.p.point_dim;
    for (int t = threadIdx.x; t < n * 3; t += blockDim.x) {
        const int o = t / 3, c = t - o * 3;
        s_xyz[t] = (c < pdim) ? feats[(base + o) * D.F + c] : 0.0;
    }
    __syncthreads();
    double* tab = tabPool + pd.tabOff + (which == 0 ? 0 : (int64_t)pd.n1 * pd.n1);
    const int nq = (n + 3) & ~3;                                  // row stride of the bins (a row is whole 8-byte words)
    uint16_t* qtab = qtabPool ? qtabPool + 4 * (int64_t)pd.qtabOff4 + (which == 0 ? 0 : (int64_t)pd.n1 * ((pd.n1 + 3) & ~3)) : nullptr;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int rr = w; rr < RB; rr += nw) {
        const int a = r0 + rr;
        if (a >= n) break;
        const double ax = s_xyz[3 * a], ay = s_xyz[3 * a + 1], az = s_xyz[3 * a + 2];
        for (int b = lane; b < n; b += WAVE) {
            const double dx = ax - s_xyz[3 * b], dy = ay - s_xyz[3 * b + 1], dz = az - s_xyz[3 * b + 2];
            const double h2 = dx * dx + dy * dy;
            const double l2 = h2 + dz * dz;
            const bool bad = (a == b) || (D.p.mindist > 0.0 && l2 < D.x_mindist);
            const double v = (D.gmode == 1 || D.gmode == 2) ? sqrt(h2) : sqrt(l2);     // z-gate mode compares full lengths
            tab[(int64_t)a * n + b] = bad ? d_nan() : v;
            // NaN entries and everything beyond the bins' range share the last bin (fmin returns the number)
            if (qtab) qtab[(int64_t)a * nq + b] = (uint16_t)(uint32_t)__builtin_fmin((bad ? d_nan() : v) * D.pre_invw, 32767.0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_live<PHASE>: single score of every association and ordered (ascending association index)
// compaction of the live ones, in two launches over chunks of LIVE_CHUNK associations (grid:
// chunks x problems, so a single problem is spread over many CUs as well):
//   PHASE 0  every wave owns a contiguous quarter of the chunk (a segment) and leaves the segment's live associations
//            (index, score; in order) at the head of the segment's slice of a scratch pool, their number in segCnt
//   PHASE 1  segment base = sum of the counts in front of it (fixed order), copy into the live pools; the
//            workgroup of chunk 0 also publishes L.
// ---------------------------------------------------------------------------------------------
constexpr int LIVE_CHUNK = 4096;

template <int PHASE>
__global__ void __launch_bounds__(256) k_live(DevParams D, const ProbDesc* __restrict__ probs,
                                              ProbState* __restrict__ st,
                                              const double* __restrict__ feats,
                                              const int32_t* __restrict__ assoc,
                                              const double* __restrict__ cosPool,
                                              double* __restrict__ qS, int32_t* __restrict__ qP /* the segments' live associations: score, index */,
                                              int32_t* __restrict__ segCnt, int maxChunks,
                                              int32_t* __restrict__ lp, int32_t* __restrict__ li,
                                              int32_t* __restrict__ lj, double* __restrict__ ls, double* __restrict__ ld,
                                              double* __restrict__ lza, double* __restrict__ lzb)
{
    __shared__ int cbase[2];
    __shared__ uint16_t survQ[PHASE == 0 ? 4 : 1][PHASE == 0 ? LIVE_CHUNK / 4 : 1];     // phase 0: per wave, the associations behind the cosine gate
    const int b = blockIdx.y, c = blockIdx.x;
    const ProbDesc pd = probs[b];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int nA = pd.nA;
    const int nChunks = (nA + LIVE_CHUNK - 1) / LIVE_CHUNK;
    if (c >= nChunks && !(PHASE == 1 && c == 0)) return;
    constexpr int SEG = LIVE_CHUNK / 4;                          // associations per wave (blockDim.x == 256)
    constexpr int NIT = SEG / WAVE;
    const int p_beg = min(nA, c * LIVE_CHUNK + w * SEG), p_end = min(nA, p_beg + SEG);
    const int Fc = D.p.cos_feature_dim;
    const int64_t lo = pd.liveOff;
    int32_t* sc = segCnt + (int64_t)b * maxChunks * 4;           // live associations per wave segment, in association order
    const unsigned long long lt = (1ull << lane) - 1ull;

    if (PHASE == 0) {
        // The wave leaves its segment's LIVE associations — index and score, in order — at the head of the segment's slice
        // of (qP, qS), and their number in segCnt: nothing is written for the ~95 % that fail (a score array over all
        // associations, written here and read back by phase 1, was 2 x 84 MB per batch of 256 at config 3 — these two
        // kernels' whole time).  With a cosine term the cheap part of the score is a gate: the wave first sweeps its
        // segment with the gate alone (all cosine loads in flight together), queues the survivors in LDS, and then
        // evaluates the full score — ratios, roots, fusion: ~250 f64 instructions — for the survivors only, 64 at a time.
        // single_score() returns 0 for an association that fails the gate whatever its ratios are, so nothing changes.
        const int pdm = D.p.point_dim;
        auto full_score = [&](int p) {
            int i, j;
            decode_assoc(pd, assoc, p, i, j);
            const double cosv = (Fc > 0) ? cosPool[pd.cosOff + (int64_t)i * pd.n2 + j] : 0.0;
            const double* fi = feats + (pd.off1 + i) * D.F + pdm; const double* fj = feats + (pd.off2 + j) * D.F + pdm;
            return single_score(D, [&](int f) { return fi[f]; }, [&](int f) { return fj[f]; }, cosv);
        };
        int nlive = 0;
        auto emit = [&](bool valid, int p, double s) {          // ordered append of the lanes with a live association
            const bool live = valid && (s > 0.0 || D.keep_all);
            const unsigned long long m = __ballot(live);
            if (live) { const int k = p_beg + nlive + __popcll(m & lt); qP[lo + k] = p; qS[lo + k] = s; }
            nlive += __popcll(m);
        };
        if (D.single && Fc > 0 && !D.keep_all) {
            uint16_t* q = survQ[w];
            const double cden = D.p.cosine_max - D.p.cosine_min;
            int ns = 0;
            double cv[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int p = p_beg + it * WAVE + lane;
                cv[it] = 0.0;
                if (p < p_end) {
                    int i, j;
                    decode_assoc(pd, assoc, p, i, j);
                    cv[it] = cosPool[pd.cosOff + (int64_t)i * pd.n2 + j];
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int p = p_beg + it * WAVE + lane;
                const bool pass = p < p_end && (D.pruned ? !(cv[it] < D.p.cosine_min) : ((cv[it] - D.p.cosine_min) / cden > 0.0));
                const unsigned long long m = __ballot(pass);
                if (pass) q[ns + __popcll(m & lt)] = (uint16_t)(p - p_beg);
                ns += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k0 = 0; k0 < ns; k0 += WAVE) {
                const int k = k0 + lane;
                const int p = p_beg + (int)q[min(k, max(ns - 1, 0))];
                emit(k < ns, p, k < ns ? full_score(p) : 0.0);
            }
        } else {
            for (int p0 = p_beg; p0 < p_end; p0 += WAVE) {
                const int p = p0 + lane;
                emit(p < p_end, p, (p < p_end && D.single) ? full_score(p) : 1.0);
            }
        }
        if (lane == 0) sc[c * 4 + w] = nlive;
        return;
    }

    // PHASE 1
    if (w == 0) {                                               // live associations in front of this chunk, and the problem total
        int front = 0, total = 0;
        for (int k0 = 0; k0 < nChunks * 4; k0 += WAVE) {
            const int k = k0 + lane;
            const int v = k < nChunks * 4 ? sc[k] : 0;
            int f = k < c * 4 ? v : 0, t = v;
            for (int off = 32; off > 0; off >>= 1) { f += __shfl_xor(f, off); t += __shfl_xor(t, off); }
            front += f; total += t;
        }
        if (lane == 0) { cbase[0] = front; cbase[1] = total; }
    }
    __syncthreads();
    // ROMAN_INV_EUCLIDEAN_PRUNED with NO survivor: the reference hands clipperpy an empty list, i.e. the all-to-all one
    const bool all_live = D.pruned && cbase[1] == 0 && nA > 0;
    const int Ltot = all_live ? nA : cbase[1];
    if (c == 0 && tid == 0) {
        st[b].L = Ltot; st[b].nnzUpper = 0ull;
        st[b].kind = (Ltot <= D.stream_maxL && D.p.maxiniters >= 1 && D.p.maxlsiters >= 1) ? 0 : (D.allow_fallback ? 1 : 2);
    }
    if (c >= nChunks) return;
    int base = cbase[0];
    for (int k = 0; k < w; ++k) base += sc[c * 4 + k];
    const int n = all_live ? p_end - p_beg : sc[c * 4 + w];
    if (all_live) base = p_beg;                                 // live index == association index
    const bool has_z = D.p.point_dim == 3;
    for (int k0 = 0; k0 < n; k0 += WAVE) {
        const int k = k0 + lane;
        if (k < n) {
            const int p = all_live ? p_beg + k : qP[lo + p_beg + k];
            const double s = all_live ? 1.0 : qS[lo + p_beg + k];
            int i, j;
            decode_assoc(pd, assoc, p, i, j);
            const int pos = base + k;
            lp[lo + pos] = p; li[lo + pos] = i; lj[lo + pos] = j; ls[lo + pos] = s;
            ld[lo + pos] = D.diag_one ? 1.0 : s;               // M_pp: the single score, or the identity (ROMAN_SINGLE_OFFDIAG)
            lza[lo + pos] = has_z ? feats[(pd.off1 + i) * D.F + 2] : 0.0;
            lzb[lo + pos] = has_z ? feats[(pd.off2 + j) * D.F + 2] : 0.0;
        }
    }
}

// k_rowbase: prefix of the live counts over the problems of the batch; also the batch maxima, the offsets of
// the per-problem candidate bit matrices (L rows of ceil(L/64) words) and the work-item prefix
// (a work item = a block of up to RPB consecutive live rows of one problem).  The bit-matrix pools were sized
// before L was known: a problem whose matrix would end beyond `capMaskWords` becomes kind 2 (skipped).
// Block-wide exclusive prefix sum (blockDim.x a multiple of 64, <= 1024): every thread gets the sum of the values of the
// threads in front of it, `total` the sum over the block.  sh: 17 elements of scratch; two barriers.
template <class T>
__device__ __forceinline__ T block_excl_scan(T v, T* sh, T& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T inc = v;
    for (int off = 1; off < WAVE; off <<= 1) { const T t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    __syncthreads();                                             // (sh may still be read from the previous call)
    if (lane == WAVE - 1) sh[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int k = 0; k < nw; ++k) { const T x = sh[k]; if (k < w) base += x; tot += x; }
    total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(1024) k_rowbase(int B, int RPB, long long capMaskWords, ProbState* __restrict__ st, BatchTotals* __restrict__ tot,
                                                  int smallOnly /* the general kernels are not launched: what k_small left behind is skipped */,
                                                  const int32_t* __restrict__ cosDense /* NULL, or k_cos_sel's flags of the batch */)
{
    // one workgroup; a thread takes PER consecutive problems (a serial sweep of one wave over 4096 problems was 93 us —
    // 64 dependent round trips to memory — of the 1.4 ms the whole batch takes at the reference's demo scale)
    __shared__ long long shl[17];
    __shared__ int shi[17];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int PER = (B + nt - 1) / nt;
    const int b0 = tid * PER, b1 = min(B, b0 + PER);
    // the thread's problems are read ONCE (up to PC of them are kept in registers: the three sweeps below would otherwise be
    // three dependent round trips to memory each)
    constexpr int PC = 8;
    int cL[PC], cK[PC];
#pragma unroll
    for (int j = 0; j < PC; ++j) { const int b = b0 + j; const bool in = j < PER && b < b1; cL[j] = in ? st[b].L : 0; cK[j] = in ? st[b].kind : 2; }
    auto getL = [&](int b) { const int j = b - b0; int v = 0; if (j < PC) {
#pragma unroll
        for (int t = 0; t < PC; ++t) if (t == j) v = cL[t]; } else v = st[b].L; return v; };
    auto getK = [&](int b) { const int j = b - b0; int v = 2; if (j < PC) {
#pragma unroll
        for (int t = 0; t < PC; ++t) if (t == j) v = cK[t]; } else v = st[b].kind; return v; };
    long long sM = 0, sN = 0; int sR = 0;
    for (int b = b0; b < b1; ++b) {
        const int L = getL(b);
        const long long mwAll = (long long)L * ((L + 63) >> 6);
        sN += mwAll; sR += L;
        if (getK(b) < 2) sM += mwAll;                            // kind 2: already skipped (k_live: no fallback kernels in this launch); kind 3: finished by k_small
    }
    long long totM, totN; int totR, totI;
    const long long baseM = block_excl_scan(sM, shl, totM);
    (void)block_excl_scan(sN, shl, totN);
    const int baseR = block_excl_scan(sR, shi, totR);
    // the prefix is monotone: once a problem does not fit, none behind it does
    int sI = 0;
    {
        long long accM = baseM;
        for (int b = b0; b < b1; ++b) {
            const int L = getL(b);
            const bool skip = getK(b) >= 2;
            const long long mw = skip ? 0 : (long long)L * ((L + 63) >> 6);
            accM += mw;
            if (!skip && accM <= capMaskWords) sI += (L + RPB - 1) / RPB;
        }
    }
    const int baseI = block_excl_scan(sI, shi, totI);
    int mx = 0, mxs = 0, nover = 0, mns = 0x7fffffff, ngen = 0, ndense = 0;
    {
        long long accM = baseM; int accR = baseR, accI = baseI;
        for (int b = b0; b < b1; ++b) {
            if (cosDense) ndense += cosDense[b] != 0;
            const int L = getL(b);
            const int kind = getK(b);
            const bool skip = kind >= 2;
            const long long mw = skip ? 0 : (long long)L * ((L + 63) >> 6);
            const bool fits = !skip && !smallOnly && accM + mw <= capMaskWords;
            const int it = fits ? (L + RPB - 1) / RPB : 0;
            st[b].rowBase = accR; st[b].itemBase = accI; st[b].maskOff = fits ? accM : 0;
            if (kind < 2) ++ngen;                                // (also when it is skipped here: the history must learn that the general kernels are needed)
            if (kind == 3) mns = min(mns, L);                    // finished by k_small: still a small problem the history should know of
            else if (!fits) {                                    // (also a problem k_live already skipped: no fallback kernels in this launch)
                st[b].kind = 2; ++nover;
                // skipped only because the general kernels were left out (small_only): the history learns its size NOW, so that the
                // second attempt sizes the solver launch for it instead of spending one more attempt on that
                if (smallOnly && kind == 0) { mxs = max(mxs, L); mns = min(mns, L); }
            }
            else if (kind == 0) { mxs = max(mxs, L); mns = min(mns, L); }
            mx = max(mx, L);
            accM += mw; accR += L; accI += it;
        }
    }
    // batch maxima / counts: wave reduction, then one atomic per wave on LDS words
    __shared__ int red[6];
    if (tid == 0) { red[0] = 0; red[1] = 0; red[2] = 0; red[3] = 0x7fffffff; red[4] = 0; red[5] = 0; }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) { mx = max(mx, __shfl_xor(mx, off)); mxs = max(mxs, __shfl_xor(mxs, off)); nover += __shfl_xor(nover, off); mns = min(mns, __shfl_xor(mns, off)); ngen += __shfl_xor(ngen, off); ndense += __shfl_xor(ndense, off); }
    if ((tid & 63) == 0) { atomicMax(&red[0], mx); atomicMax(&red[1], mxs); atomicAdd(&red[2], nover); atomicMin(&red[3], mns); atomicAdd(&red[4], ngen); atomicAdd(&red[5], ndense); }
    __syncthreads();
    if (tid == 0) {
        tot->cosScreened = cosDense ? B : 0; tot->cosDense = red[5];
        tot->R = totR; tot->maxL = red[0]; tot->nnzTotal = 0; tot->maskWords = totM < capMaskWords ? totM : capMaskWords; tot->items = totI; tot->sliceGroups = 0;
        tot->needMaskWords = totN; tot->needNnz = 0; tot->overflow = red[2]; tot->maxStreamL = red[1]; tot->minStreamL = red[3]; tot->nGeneral = red[4]; tot->listTop = 0ull;
    }
}

// k_items: the work-item list of the pair-test and fill kernels.
__global__ void __launch_bounds__(256) k_items(int RPB, const ProbState* __restrict__ st, ItemDesc* __restrict__ items)
{
    const int b = blockIdx.x;
    const int L = st[b].L, ib = st[b].itemBase;
    const int n = (st[b].kind >= 2) ? 0 : (L + RPB - 1) / RPB;
    for (int t = threadIdx.x; t < n; t += blockDim.x) { ItemDesc d; d.b = b; d.row0 = t * RPB; items[ib + t] = d; }
}

// ---------------------------------------------------------------------------------------------
// k_count: the O(L^2) pair tests of the affinity build, done ONCE.  A workgroup takes a work item
// (RPB consecutive live rows of one problem) and stages the problem's COLUMN data (map-1/map-2
// object and the two z coordinates of every live association) in LDS once; each wave then owns a
// row k=(i,j) at a time: it stages the two table rows TA[i][:] and TB[j][:] in its private LDS
// slice and sweeps the live columns 64 at a time.  A test costs three coalesced LDS reads, two LDS
// gathers and ~12 f64 VALU ops, all exactly rounded (+,-,*,compare).  Output per row: the candidate
// bit mask (one ballot word per 64 columns), the running candidate count in front of every word
// (k_fill turns it into the entry index without another scan) and the row total.
// A live set that does not fit the LDS column tile is swept tile by tile.
// ---------------------------------------------------------------------------------------------
// exclusive prefix sum over the 64 lanes of a wave
// (DPP: six VALU instructions — row_shr:1,2,4,8 inside the rows of 16 lanes, row_bcast:15 / :31 across them; lanes without a
// source add the `old` operand, 0.  The __shfl_up form compiles to six DEPENDENT ds_bpermute round trips through the LDS
// crossbar, ~500 cycles of latency per scan: k_upper spends one scan per matrix row.)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane) { (void)lane; return wave_incl_scan(v) - v; }

// lane `sel` of (hi:lo) <- the wave-uniform 64-bit value m (two v_writelane_b32: uniform value, uniform lane select).  The
// instruction's constant bus takes one SGPR: the lane select travels in m0.  This compiler exposes no writelane builtin, and
// m0 is a reserved register that an asm statement may not list as clobbered: the statement saves and restores it itself.
__device__ __forceinline__ void writelane2(uint32_t& lo, uint32_t& hi, unsigned long long m, uint32_t sel)
{
    const uint32_t ml_ = (uint32_t)m, mh_ = (uint32_t)(m >> 32);
    uint32_t keep_;
    asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
        : "+v"(lo), "+v"(hi), "=&s"(keep_) : "s"(ml_), "s"(sel), "s"(mh_));
}

// GM: 0 no gravity prior, 1 ROMAN_GRAV_COMBINED, 2 ROMAN_GRAV_SEPARATE, 3 ROMAN_GRAV_ZGATE (tables then hold full lengths)
template <int GM>
__device__ __forceinline__ bool pair_gate(const DevParams& D, double a, double bb, double dz)
{
    if (GM == 1) {
        const double ch = fabs(a - bb);
        double hm;                              // max(a,bb) in ONE instruction (fmax() would first
        asm("v_max_f64 %0, %1, %2" : "=v"(hm) : "v"(a), "v"(bb));   // canonicalise both); NaN operands: x is NaN through ch anyway
        const double cv = __builtin_fmax(dz - D.sin_unc * hm, 0.0);
        const double xx = ch * ch + cv * cv;
        return xx < D.x_eps;                    // <=> sqrt(x) < epsilon ; NaN -> false
    } else if (GM == 2) {
        const double ch = fabs(a - bb);
        double hm;
        asm("v_max_f64 %0, %1, %2" : "=v"(hm) : "v"(a), "v"(bb));
        const double cv = __builtin_fmax(dz - D.sin_unc * hm, 0.0);
        return (ch < D.p.epsilon) & (cv < D.p.epsilon);             // a NaN table entry fails through ch
    } else if (GM == 3) {
        const double c = fabs(a - bb);
        double lm;
        asm("v_max_f64 %0, %1, %2" : "=v"(lm) : "v"(a), "v"(bb));
        return (c < D.p.epsilon) & (dz < D.p.epsilon + D.sin_unc * lm);
    } else {
        return fabs(a - bb) < D.p.epsilon;
    }
}

// Fast sweep: columns in LDS as the packed pair {i_q, n1+1+j_q} of indices into a table slice (+ the two z
// coordinates), padded to a multiple of 256 columns with a sentinel whose table entry is NaN (fails every
// test), so the inner loop has no bounds logic at all.  A wave sweeps NR ADJACENT rows at once (NR = 2 when
// the table slices fit): the column data is read once for both, the two rows' instruction streams are
// independent (the kernel is bound by LDS/VALU/scalar latency, not by any one pipe), and the loop overhead
// is shared.  Per 64 columns and row: 2 LDS gathers, 12 f64 VALU ops, 2 address adds, 2 v_writelane.
template <int GM, int NR, bool TILED>
__device__ __forceinline__ void count_rows_lds(const DevParams& D, const ProbDesc& pd, int L, int row0, int nrows,
                                               int w, int wpb, int lane,
                                               const uint32_t* cIJ /* i_q | (n1 + 1 + j_q) << 16: table-slice indices */, const double2* cZZ,
                                               const double* __restrict__ TA, const double* __restrict__ TB,
                                               double* tA /* NR slices of ldsPerRow doubles */, int ldsPerRow,
                                               unsigned long long* __restrict__ mbase,
                                               int c0_ /* first column of the LDS tile */, int clen /* its columns (multiple of 256) */,
                                               const int32_t* __restrict__ gI, const int32_t* __restrict__ gJ,
                                               const double* __restrict__ gZa, const double* __restrict__ gZb /* the rows' own data when the tile does not hold them (c0 > 0 or clen < Lpad) */)
{
    const int W = (L + 63) >> 6;
    // TILED == false: the tile holds the whole live set (c0 = 0 folds away: the sweep is the one-tile code of rounds 1-3)
    constexpr bool tiled = TILED;
    const int c0 = TILED ? c0_ : 0;
    const int cend = TILED ? c0 + clen : 0x7fffffff;
    const char* tbytes = reinterpret_cast<const char*>(tA);
    const int sliceBytes = ldsPerRow * 8;
    constexpr int U = 2;                                        // column chunks per step (with 2 rows per wave: 2 beats 1 and 4)
    const int Lpad = (L + U * WAVE - 1) & ~(U * WAVE - 1);
    // Rows are columns too: their objects and z come from the LDS column tile.  The table rows of the NEXT
    // rows are fetched into registers while the current ones are swept (maps of up to 256 objects; larger ones
    // load directly), so no global-memory latency sits between two sweeps.
    constexpr int TR = 4;
    const bool pre = pd.n1 <= TR * WAVE && pd.n2 <= TR * WAVE;
    double ra[NR][TR], rb[NR][TR];
    auto fetch_tab = [&](int r_) {
#pragma unroll
        for (int x = 0; x < NR; ++x) {
            const int k_ = row0 + min(r_ + x, nrows - 1);
            const int i_ = tiled ? gI[k_] : (int)(cIJ[k_] & 0xffffu), j_ = tiled ? gJ[k_] : (int)(cIJ[k_] >> 16) - (pd.n1 + 1);
            const double* gA_ = TA + (int64_t)i_ * pd.n1;
            const double* gB_ = TB + (int64_t)j_ * pd.n2;
#pragma unroll
            for (int m_ = 0; m_ < TR; ++m_) {
                ra[x][m_] = (lane + m_ * WAVE < pd.n1) ? gA_[lane + m_ * WAVE] : 0.0;
                rb[x][m_] = (lane + m_ * WAVE < pd.n2) ? gB_[lane + m_ * WAVE] : 0.0;
            }
        }
    };
    if (pre && NR * w < nrows) fetch_tab(__builtin_amdgcn_readfirstlane(NR * w));
    for (int r = NR * w; r < nrows; r += NR * wpb) {
        int k[NR]; double zi[NR], zj[NR];
#pragma unroll
        for (int x = 0; x < NR; ++x) {
            k[x] = __builtin_amdgcn_readfirstlane(row0 + min(r + x, nrows - 1));   // wave-uniform: loop bounds and lane selects in SGPRs
            zi[x] = GM ? (tiled ? gZa[k[x]] : cZZ[k[x]].x) : 0.0; zj[x] = GM ? (tiled ? gZb[k[x]] : cZZ[k[x]].y) : 0.0;
        }
        // stage the table rows (wave-private slices; LDS ops of one wave execute in order)
#pragma unroll
        for (int x = 0; x < NR; ++x) {
            double* sA = tA + x * ldsPerRow; double* sB = sA + pd.n1 + 1;
            if (pre) {
#pragma unroll
                for (int m_ = 0; m_ < TR; ++m_) {
                    if (lane + m_ * WAVE < pd.n1) sA[lane + m_ * WAVE] = ra[x][m_];
                    if (lane + m_ * WAVE < pd.n2) sB[lane + m_ * WAVE] = rb[x][m_];
                }
            } else {
                const int i = tiled ? gI[k[x]] : (int)(cIJ[k[x]] & 0xffffu), j = tiled ? gJ[k[x]] : (int)(cIJ[k[x]] >> 16) - (pd.n1 + 1);
                const double* gA = TA + (int64_t)i * pd.n1;
                const double* gB = TB + (int64_t)j * pd.n2;
                for (int t = lane; t < pd.n1; t += WAVE) sA[t] = gA[t];
                for (int t = lane; t < pd.n2; t += WAVE) sB[t] = gB[t];
            }
            if (lane == 0) sA[pd.n1] = d_nan();                 // sentinel entry of the padding columns
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (pre && r + NR * wpb < nrows) fetch_tab(__builtin_amdgcn_readfirstlane(r + NR * wpb));

        // The pair test is symmetric: row k computes only the 64-column words c >= R = k/64 (the diagonal word
        // completely, both of its triangles); the words c < R are the bit transposes of blocks computed by other
        // rows and are written by k_mirror.  Prefix counts and row totals are taken afterwards by k_rowprefix.
        // The NR rows of a wave are adjacent and start at a multiple of NR: they lie in the same 64-row block.
        const int R = k[0] >> 6;
        uint32_t mlo[NR], mhi[NR];                              // lane l: word (block*64 + l) of the current 64-word block
#pragma unroll
        for (int x = 0; x < NR; ++x) { mlo[x] = 0u; mhi[x] = 0u; }
        const int qEnd = TILED ? min(Lpad, cend) : Lpad;        // (a live set larger than the LDS tile is swept tile by tile: k_count)
        for (int q0 = max((R << 6) & ~(U * WAVE - 1), c0); q0 < qEnd; q0 += U * WAVE) {
            int2 ij[U]; double2 zz[U];
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const uint32_t pk = cIJ[q0 - c0 + t * WAVE + lane];
                ij[t] = make_int2((int)((pk & 0xffffu) << 3), (int)((pk >> 16) << 3));     // byte offsets into a table slice
                if (GM) zz[t] = cZZ[q0 - c0 + t * WAVE + lane];
            }
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const int widx = TILED ? uni_i((q0 >> 6) + t) : (q0 >> 6) + t;   // words >= W are all-zero (sentinel columns); (out-of-line tiled path: arguments arrive in vector registers)
#pragma unroll
                for (int x = 0; x < NR; ++x) {
                    const double a = *reinterpret_cast<const double*>(tbytes + x * sliceBytes + ij[t].x);
                    const double bb = *reinterpret_cast<const double*>(tbytes + x * sliceBytes + ij[t].y);
                    const double dz = GM ? fabs((zi[x] - zz[t].x) - (zj[x] - zz[t].y)) : 0.0;
                    const bool is = pair_gate<GM>(D, a, bb, dz);
                    const unsigned long long m = __ballot(is);
                    // lane (widx & 63) of (mhi:mlo) <- m
                    writelane2(mlo[x], mhi[x], m, (uint32_t)(widx & 63));
                }
            }
            const int wend = min(W, (q0 >> 6) + U);             // words [.., wend) are complete
            if ((wend & 63) == 0 || wend == W || (TILED && q0 + U * WAVE >= qEnd)) {   // flush the block of <= 64 words, coalesced (also at the end of a tile)
                const int wb = (wend - 1) & ~63;
                const int wlo = TILED ? max(R, c0 >> 6) : R;    // words in front of this tile were written when their tile was swept
#pragma unroll
                for (int x = 0; x < NR; ++x) {
                    const unsigned long long mreg = ((unsigned long long)mhi[x] << 32) | mlo[x];
                    if (r + x < nrows && wb + lane < wend && wb + lane >= wlo) mbase[(int64_t)k[x] * W + wb + lane] = mreg;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                        // table slices are rewritten by the next rows
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------
// The same pair tests with CANDIDATE GENERATION in front of the exact gate (round 6).  Every reading of the gate needs
// |TA[i][i'] - TB[j][j']| < epsilon (the horizontal / full-length difference is one of its terms and the other is >= 0), and only
// ~3 % of the live pairs pass it.  What bounds the plain sweep is its two random f64 LDS gathers per 64 tests and row (one LDS per
// compute unit serves sixteen waves); a first prefilter that kept two gathers per row — 16-bit bins instead of doubles — was
// bit-identical and SLOWER (650 against 570 us per batch: profiles/r06).  Here a wave sweeps FOUR adjacent rows at once and the
// four rows' bins of one table column are ONE 8-byte LDS word: two 8-byte gathers per 64 columns serve 256 pair tests.  Bins are
// epsilon / 32 wide (15 bits; NaN entries and everything beyond the range share the last bin), compared as packed 16-bit halves;
// the survivors (~6 % of the columns) are compacted into a per-wave LDS queue, and the EXISTING exact f64 gate runs on the queue,
// 64 candidates at a time with every lane busy, its operands fetched from the tables in memory (L2) for those few; a passing
// candidate sets its bit in the row's mask words in LDS (ds_or_b64), which go out coalesced at the end of the rows.
// The prefilter is a superset of the gate by construction: q(x) = trunc(min(x * invw, 32767)) is monotone, so |a - b| <= eps'
// gives |q(a) - q(b)| <= ceil(eps' * invw + rounding) <= 32 + 1; pre_K is 34.  A false positive (the last bin's far / NaN pairs
// included) costs an exact test, never a bit.  The exact gate sees the same operands in the same operation order as
// count_rows_lds: the mask words are bit-identical (tests/test_gpu_parity.py runs ladder cases, the ends of the bin range and
// the +-1-ulp plants through both kernels).
// ---------------------------------------------------------------------------------------------
constexpr int PRE_NR = 4;                 // rows a wave sweeps together (their bins share an LDS word)
constexpr int PRE_COLPAD = 128;           // sentinel columns behind the column tile (a sweep starts at the rows' own 64-column block and advances by 128)
constexpr int PRE_QCAP = 640;             // queue entries: the exact gate runs while 256 are waiting; a window's candidates go in at once when they fit (else 64 at a time)
#ifndef ROMAN_PRE_WAVES
#define ROMAN_PRE_WAVES 16                // waves per workgroup of the prefiltered sweep (118 registers per lane: no spills; 12 waves measured 7 % slower)
#endif
constexpr int PRE_WAVES = ROMAN_PRE_WAVES;
// per-wave LDS of the prefiltered sweep: PRE_NR mask rows (TC / 64 words), the packed bin table (n1 + 1 + n2 entries of 8 bytes, ldsPerRow
// rounded), the queue (16-bit entries), the rows' own data (4 x 32 bytes)
__host__ __device__ constexpr int count_pre_wave_bytes(int ldsPerRow, int TC)
{
    return PRE_NR * (TC >> 6) * 8 + ldsPerRow * 8 + PRE_QCAP * 2 + PRE_NR * 32;
}
struct PreRow { double zi, zj; uint32_t rowA, rowB, moff, krow; };   // heights of the row's objects; first entries of its two table rows; byte offset of its mask words; the live row
typedef short pre_s2 __attribute__((ext_vector_type(2)));

template <int GM>
__device__ __forceinline__ void count_rows_pre(const DevParams& D, const ProbDesc& pd, int L, int row0, int nrows,
                                               uint32_t* qctr /* LDS: next quad of rows to hand out (0 on entry) */, int lane,
                                               const uint32_t* cIJ /* i_q | (n1 + 1 + j_q) << 16: table-slice indices; sentinel columns up to ((L + 127) & ~127) + 128 */,
                                               const double* __restrict__ TA, const double* __restrict__ TB,
                                               const uint16_t* __restrict__ QA, const uint16_t* __restrict__ QB /* the tables as bins (k_tables): rows of (n + 3 & ~3) entries */,
                                               uint2* qT /* n1 + 1 + n2 packed bin entries */, uint16_t* queue,
                                               unsigned long long* rmask /* PRE_NR rows of Wcap words, zero on entry and on exit */, int Wcap,
                                               PreRow* rowinfo,
                                               unsigned long long* __restrict__ mbase,
                                               const double* __restrict__ gZa, const double* __restrict__ gZb,
                                               uint32_t* degS /* LDS, or NULL: the full degree of every live row, counted as the pairs pass (whole problems only) */,
                                               const dbl2_t* sO /* LDS: the objects' coordinates, (x, y) (z, -), map 1 at 0, map 2 at NO */, int NO /* 0: the exact gate reads the tables */)
{
    constexpr int NR = PRE_NR;
    const bool horiz = D.gmode == 1 || D.gmode == 2;            // the tables hold horizontal distances (k_tables)
    const int W = (L + 63) >> 6;
    const int Lpad = (L + 2 * WAVE - 1) & ~(2 * WAVE - 1);
    const uint32_t K2 = (uint32_t)D.pre_K * 0x00010001u;
    const int n1 = pd.n1, n2 = pd.n2;
    const int nq1 = (n1 + 3) & ~3, nq2 = (n2 + 3) & ~3;
    const bool pre = n1 <= 4 * WAVE && n2 <= 4 * WAVE;         // a lane holds four consecutive entries of a row: one 8-byte load per row
    uint2 ra[NR], rb[NR];
    int pi[NR], pj[NR];                                         // objects of the rows whose bin rows are in ra / rb
    double pzi = 0.0, pzj = 0.0;                                // lane x < NR: the heights of row x's two objects
    auto fetch_tab = [&](int r_) {
        if (GM) { const int kz_ = row0 + min(r_ + min(lane, NR - 1), nrows - 1); pzi = gZa[kz_]; pzj = gZb[kz_]; }
#pragma unroll
        for (int x = 0; x < NR; ++x) {
            const int k_ = row0 + min(r_ + x, nrows - 1);
            const uint32_t pk_ = cIJ[k_];
            pi[x] = __builtin_amdgcn_readfirstlane((int)(pk_ & 0xffffu)); pj[x] = __builtin_amdgcn_readfirstlane((int)(pk_ >> 16) - (n1 + 1));
            ra[x] = *reinterpret_cast<const uint2*>(QA + (int64_t)pi[x] * nq1 + min(4 * lane, nq1 - 4));
            rb[x] = *reinterpret_cast<const uint2*>(QB + (int64_t)pj[x] * nq2 + min(4 * lane, nq2 - 4));
        }
    };
    // Quads of rows are handed out in order (an LDS counter): the rows in front sweep the most columns, so the longest quads start
    // first and the waves of a workgroup finish within one quad of each other whatever their number
    auto grab = [&]() -> int {
        uint32_t g = 0u;
        if (lane == 0) g = __hip_atomic_fetch_add(qctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return NR * __builtin_amdgcn_readfirstlane((int)g);
    };
    int r = grab(), rnext = nrows;
    if (pre && r < nrows) fetch_tab(r);
    uint32_t qn = 0;                                            // queue fill (wave-uniform)
#ifdef ROMAN_COUNT_TIMING
    unsigned long long tc_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tm_; unsigned nq_ = 0, ncons_ = 0, npush_ = 0;   // stage, sweep, push, consume, store, (unused)
#define TCK(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); tc_[k_] += n_ - tm_; tm_ = n_; } while (0)
#else
#define TCK(k_) do { } while (0)
#endif
    auto lds_order = [&]() {                                    // LDS operations of one wave execute in order: this is for the compiler
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // The exact gate over up to 256 queue entries starting at `base`: four candidates per lane, the operands of all four requested
    // (tables and heights from memory: L2) before the first is used — a single chain's round trip would otherwise be all a wave does
    auto consume = [&](uint32_t base, uint32_t n) {
        constexpr int CH = 4;
        uint32_t e[CH]; double a[CH], bb[CH], za[CH], zb[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t idx = (uint32_t)(c * WAVE + lane);
            e[c] = idx < n ? (uint32_t)queue[base + idx] : 0xffffu;     // (0x3fff: a column beyond every live set: inactive below)
            const uint32_t qq = e[c] & 0x3fffu;
            const bool act = (int)qq < L;                       // (a sentinel column can only get here through the last bin)
            const uint32_t pk = cIJ[act ? qq : 0u];
            const PreRow* ri = rowinfo + (e[c] >> 14);
            if (NO > 0) {
                // the two distances RECOMPUTED from the coordinates in LDS with k_tables' own operation sequence (dx*dx + dy*dy (+ dz*dz), one
                // correctly rounded sqrt: the same bits, as in k_fill_list), the heights from there too: no request leaves the compute unit
                // (four L2 gathers per candidate before: 136 M per batch of 256 — the exact gate was a third of a quad's cycles)
                const dbl2_t oa0 = sO[2 * ri->rowA], oa1 = sO[2 * ri->rowA + 1], ob0 = sO[2 * (NO + ri->rowB)], ob1 = sO[2 * (NO + ri->rowB) + 1];   // (rowA, rowB: the row's OBJECTS here)
                const uint32_t iq = pk & 0xffffu, jq = (pk >> 16) - (uint32_t)(n1 + 1);
                const dbl2_t pa0 = sO[2 * iq], pa1 = sO[2 * iq + 1], pb0 = sO[2 * (NO + jq)], pb1 = sO[2 * (NO + jq) + 1];
                const double dxa = oa0.x - pa0.x, dya = oa0.y - pa0.y, dxb = ob0.x - pb0.x, dyb = ob0.y - pb0.y;
                const double dza = oa1.x - pa1.x, dzb = ob1.x - pb1.x;
                const double h2a = dxa * dxa + dya * dya, h2b = dxb * dxb + dyb * dyb;
                a[c] = horiz ? sqrt(h2a) : sqrt(h2a + dza * dza);
                bb[c] = horiz ? sqrt(h2b) : sqrt(h2b + dzb * dzb);
                if (GM) { za[c] = pa1.x; zb[c] = pb1.x; }
            } else {
            a[c] = TA[ri->rowA + (pk & 0xffffu)];               // (32-bit element offsets: maps of at most 32767 objects take this sweep)
            bb[c] = TB[ri->rowB + (pk >> 16)];                  // (rowB is short of the row's start by n1 + 1: the packed index carries it)
            if (GM) { za[c] = gZa[act ? qq : 0u]; zb[c] = gZb[act ? qq : 0u]; }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t qq = e[c] & 0x3fffu, x = e[c] >> 14;
            double dz = 0.0;
            if (GM) { const PreRow* ri = rowinfo + x; dz = fabs((ri->zi - za[c]) - (ri->zj - zb[c])); }
            const bool is = (int)qq < L && pair_gate<GM>(D, a[c], bb[c], dz);
            if (is) {
                const PreRow* ri = rowinfo + x;
                (void)__hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(rmask) + ri->moff + ((qq >> 6) << 3)),
                                            1ull << (qq & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // a pair of the strict upper triangle counts for both of its rows (k_lists' degrees: it then skips its own sweep)
                if (degS != nullptr && qq > ri->krow) {
                    (void)__hip_atomic_fetch_add(degS + ri->krow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_add(degS + qq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    // candidate bits of a window (up to four steps of two 64-column chunks: bit 8 s + 4 t + x of a lane = column qwin + (2 s + t) 64 + lane,
    // row x) -> queue entries; the exact gate runs whenever 256 entries are waiting
    auto flush = [&](uint32_t acc, int qwin) {
        const uint32_t cnt = (uint32_t)__builtin_popcount(acc);
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t qb = (uint32_t)(qwin + lane);
        if (qn + total <= (uint32_t)PRE_QCAP) {
            // the usual case: every lane writes its own candidates behind those of the lanes in front of it (one scan, no ballots)
            uint32_t at = qn + incl - cnt;
            while (__ballot(acc != 0u) != 0ull) {
                if (acc != 0u) {
                    const uint32_t b = (uint32_t)__builtin_ctz(acc);
                    acc &= acc - 1u;
                    queue[at++] = (uint16_t)((qb + ((b >> 2) << 6)) | ((b & 3u) << 14));
                }
#ifdef ROMAN_COUNT_TIMING
                ++npush_;
#endif
            }
            qn += total;
        } else {
            // a window with more candidates than the queue has room for: 64 at a time, the exact gate in between
            while (true) {
                const unsigned long long m = __ballot(acc != 0u);
                if (m == 0ull) break;
                const uint32_t b = (uint32_t)__builtin_ctz(acc | 0x80000000u);
                const bool has = acc != 0u;
                acc &= acc - 1u;
                const uint32_t at = qn + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (has) queue[at] = (uint16_t)((qb + ((b >> 2) << 6)) | ((b & 3u) << 14));
                qn += (uint32_t)__popcll(m);
                if (qn >= 256u) { TCK(2); lds_order(); qn -= 256u; consume(qn, 256u); lds_order(); TCK(3); }
            }
        }
        if (qn >= 256u) {
            TCK(2); lds_order();
            while (qn >= 256u) { qn -= 256u; consume(qn, 256u);
#ifdef ROMAN_COUNT_TIMING
                ++ncons_;
#endif
            }
            lds_order(); TCK(3);
        }
    };
    for (; r < nrows; r = rnext) {
#ifdef ROMAN_COUNT_TIMING
        tm_ = __builtin_readcyclecounter(); ++nq_;
#endif
        int k[NR];
#pragma unroll
        for (int x = 0; x < NR; ++x) k[x] = __builtin_amdgcn_readfirstlane(row0 + min(r + x, nrows - 1));
        if (!pre) {
#pragma unroll
            for (int x = 0; x < NR; ++x) {
                const uint32_t pk_ = cIJ[k[x]];
                pi[x] = __builtin_amdgcn_readfirstlane((int)(pk_ & 0xffffu)); pj[x] = __builtin_amdgcn_readfirstlane((int)(pk_ >> 16) - (n1 + 1));
            }
        }
        if (!pre && GM) { const int kz_ = row0 + min(r + min(lane, NR - 1), nrows - 1); pzi = gZa[kz_]; pzj = gZb[kz_]; }
        if (lane < NR) {                                        // the rows' own objects and heights, for the exact gate
            int i_ = pi[0], j_ = pj[0];
#pragma unroll
            for (int x = 1; x < NR; ++x) if (lane == x) { i_ = pi[x]; j_ = pj[x]; }
            PreRow ri; ri.rowA = (uint32_t)i_ * (uint32_t)n1; ri.rowB = (uint32_t)j_ * (uint32_t)n2 - (uint32_t)(n1 + 1);
            if (NO > 0) { ri.rowA = (uint32_t)i_; ri.rowB = (uint32_t)j_; }
            ri.moff = (uint32_t)lane * (uint32_t)Wcap * 8u; ri.zi = pzi; ri.zj = pzj;
            int kr_ = k[0];
#pragma unroll
            for (int x = 1; x < NR; ++x) if (lane == x) kr_ = (r + x < nrows) ? k[x] : 0x7fffffff;   // (a duplicated last row counts nothing)
            ri.krow = (uint32_t)kr_;
            rowinfo[lane] = ri;
        }
        // the packed bin table: entry t = the NR rows' bins of table column t (map-1 columns, the sentinel, map-2 columns)
        if (pre) {
            // lane l holds entries 4 l .. 4 l + 3 of every row: the low / high halves of its two words, zipped over the rows
            const int t0 = 4 * lane;
            if (t0 < n1) qT[t0] = make_uint2(__builtin_amdgcn_perm(ra[1].x, ra[0].x, 0x05040100u), __builtin_amdgcn_perm(ra[3].x, ra[2].x, 0x05040100u));
            if (t0 + 1 < n1) qT[t0 + 1] = make_uint2(__builtin_amdgcn_perm(ra[1].x, ra[0].x, 0x07060302u), __builtin_amdgcn_perm(ra[3].x, ra[2].x, 0x07060302u));
            if (t0 + 2 < n1) qT[t0 + 2] = make_uint2(__builtin_amdgcn_perm(ra[1].y, ra[0].y, 0x05040100u), __builtin_amdgcn_perm(ra[3].y, ra[2].y, 0x05040100u));
            if (t0 + 3 < n1) qT[t0 + 3] = make_uint2(__builtin_amdgcn_perm(ra[1].y, ra[0].y, 0x07060302u), __builtin_amdgcn_perm(ra[3].y, ra[2].y, 0x07060302u));
            uint2* qB = qT + n1 + 1;
            if (t0 < n2) qB[t0] = make_uint2(__builtin_amdgcn_perm(rb[1].x, rb[0].x, 0x05040100u), __builtin_amdgcn_perm(rb[3].x, rb[2].x, 0x05040100u));
            if (t0 + 1 < n2) qB[t0 + 1] = make_uint2(__builtin_amdgcn_perm(rb[1].x, rb[0].x, 0x07060302u), __builtin_amdgcn_perm(rb[3].x, rb[2].x, 0x07060302u));
            if (t0 + 2 < n2) qB[t0 + 2] = make_uint2(__builtin_amdgcn_perm(rb[1].y, rb[0].y, 0x05040100u), __builtin_amdgcn_perm(rb[3].y, rb[2].y, 0x05040100u));
            if (t0 + 3 < n2) qB[t0 + 3] = make_uint2(__builtin_amdgcn_perm(rb[1].y, rb[0].y, 0x07060302u), __builtin_amdgcn_perm(rb[3].y, rb[2].y, 0x07060302u));
        } else {
            for (int t = lane; t < n1; t += WAVE) {
                uint32_t b_[NR];
#pragma unroll
                for (int x = 0; x < NR; ++x) b_[x] = QA[(int64_t)pi[x] * nq1 + t];
                qT[t] = make_uint2(b_[0] | (b_[1] << 16), b_[2] | (b_[3] << 16));
            }
            for (int t = lane; t < n2; t += WAVE) {
                uint32_t b_[NR];
#pragma unroll
                for (int x = 0; x < NR; ++x) b_[x] = QB[(int64_t)pj[x] * nq2 + t];
                qT[n1 + 1 + t] = make_uint2(b_[0] | (b_[1] << 16), b_[2] | (b_[3] << 16));
            }
        }
        if (lane == 0) qT[n1] = make_uint2(0x7fff7fffu, 0x7fff7fffu);   // sentinel entry of the padding columns
        lds_order();
        rnext = grab();
        if (pre && rnext < nrows) fetch_tab(rnext);

        TCK(0);
        const int R = k[0] >> 6;                                // the NR rows of a wave lie in the same 64-row block
        const char* qbytes = reinterpret_cast<const char*>(qT);
        uint32_t nacc = 0u;                                     // NOT-candidate bits of the current window
        int step = 0, qwin = R << 6;
        for (int q0 = R << 6; q0 < Lpad; q0 += 2 * WAVE, ++step) {
            uint32_t pk[2]; uint2 ea[2], eb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) pk[t] = cIJ[q0 + t * WAVE + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ea[t] = *reinterpret_cast<const uint2*>(qbytes + ((pk[t] & 0xffffu) << 3));
                eb[t] = *reinterpret_cast<const uint2*>(qbytes + ((pk[t] >> 16) << 3));
            }
            uint32_t n8 = 0u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // per 16-bit half: K - |a - b| is negative (sign bit set) where the pair is NOT a candidate
                const pre_s2 c0 = __builtin_bit_cast(pre_s2, K2) - __builtin_elementwise_abs(__builtin_bit_cast(pre_s2, ea[t].x) - __builtin_bit_cast(pre_s2, eb[t].x));
                const pre_s2 c1 = __builtin_bit_cast(pre_s2, K2) - __builtin_elementwise_abs(__builtin_bit_cast(pre_s2, ea[t].y) - __builtin_bit_cast(pre_s2, eb[t].y));
                const uint32_t u = ((__builtin_bit_cast(uint32_t, c0) & 0x80008000u) >> 15) | ((__builtin_bit_cast(uint32_t, c1) & 0x80008000u) >> 13);
                n8 |= ((u | (u >> 15)) & 0xfu) << (4 * t);      // row 0 -> bit 0, row 1 (bit 16) -> bit 1, row 2 -> bit 2, row 3 (bit 18) -> bit 3
            }
            nacc |= n8 << ((step & 3) << 3);
            if ((step & 3) == 3) {
                TCK(1);
                flush(~nacc, qwin);
                TCK(2);
                nacc = 0u; qwin = q0 + 2 * WAVE;
            }
        }
        TCK(1);
        if (step & 3) flush(~nacc & ((1u << ((step & 3) << 3)) - 1u), qwin);
        TCK(2);
        lds_order();
        if (qn) { consume(0u, qn); qn = 0u; }
        lds_order();
        TCK(3);
        // the rows' words [R, W): out, coalesced, and cleared for the next rows
#pragma unroll
        for (int x = 0; x < NR; ++x) {
            for (int wv = R + lane; wv < W; wv += WAVE) {
                const unsigned long long mreg = rmask[x * Wcap + wv];
                rmask[x * Wcap + wv] = 0ull;
                // (non-temporal: the mask rows are read next by another kernel, from whatever XCD; measured -5 us here, -3 us in k_lists)
                if (r + x < nrows) __builtin_nontemporal_store(mreg, &mbase[(int64_t)k[x] * W + wv]);
            }
        }
        lds_order();                                            // the bin table and the rows' data are rewritten by the next rows
        TCK(4);
    }
#ifdef ROMAN_COUNT_TIMING
    if (lane == 0 && (threadIdx.x >> 6) == 0 && (blockIdx.x & 63) == 0 && row0 == 0 && nq_ > 0)
        printf("[k_count pre] wg %d L %d quads %u: stage %llu sweep %llu push %llu (iterations %u) consume %llu (+%u inside the sweep) store %llu cycles per quad\n", (int)blockIdx.x, L, nq_,
               tc_[0] / nq_, tc_[1] / nq_, tc_[2] / nq_, npush_ / nq_, tc_[3] / nq_, ncons_, tc_[4] / nq_);
#endif
#undef TCK
}

// A work item of a live set that does not fit the LDS column tile (no semantic gate: L = n1 * n2): the item's rows are swept
// against one tile of columns after the other (a row needs only the words at and behind its own 64-row block: the tiles in
// front of the item's first row are skipped).  Its own kernel instantiation (k_count<..., true>): the one-tile kernel keeps the code it had.
template <int GM, int NR>
__device__ __forceinline__ void count_item_tiled(const DevParams& D, const ProbDesc& pd, int L, int row0, int nrows,
                                              uint32_t* cIJ, double2* cZZ, const double* __restrict__ TA, const double* __restrict__ TB,
                                              double* tA, int ldsPerRow, unsigned long long* __restrict__ mbase, int TC,
                                              const int32_t* __restrict__ gI, const int32_t* __restrict__ gJ,
                                              const double* __restrict__ gZa, const double* __restrict__ gZb)
{
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = uni_i(tid >> 6), wpb = nt >> 6;
    L = uni_i(L); row0 = uni_i(row0); nrows = uni_i(nrows); TC = uni_i(TC); ldsPerRow = uni_i(ldsPerRow);   // (wave-uniform: back into scalar registers)
    const int Lpad = (L + 255) & ~255;
    for (int c0 = (((row0 >> 6) << 6) / TC) * TC; c0 < Lpad; c0 += TC) {
        const int clen = min(TC, Lpad - c0);
        __syncthreads();                        // every wave is done with the previous tile's columns
        for (int q = tid; q < clen; q += nt) {
            const int qq = c0 + q;
            const bool v = qq < L;
            cIJ[q] = v ? ((uint32_t)gI[qq] | ((uint32_t)(pd.n1 + 1 + gJ[qq]) << 16)) : ((uint32_t)pd.n1 | ((uint32_t)(pd.n1 + 1) << 16));
            if (GM) cZZ[q] = v ? make_double2(gZa[qq], gZb[qq]) : make_double2(0.0, 0.0);
        }
        __syncthreads();
        count_rows_lds<GM, NR, true>(D, pd, L, row0, nrows, w, wpb, lane, cIJ, cZZ, TA, TB, tA, ldsPerRow, mbase, c0, clen, gI, gJ, gZa, gZb);
    }
}

template <int GM, int NR, bool TILED, bool PRE = false /* candidate generation in front of the exact gate (one-tile sweep only): count_rows_pre */>
__global__ void __launch_bounds__(PRE ? PRE_WAVES * 64 : 1024) k_count(DevParams D, const ProbDesc* __restrict__ probs,
                                                const ProbState* __restrict__ st,
                                                const BatchTotals* __restrict__ tot,
                                                const ItemDesc* __restrict__ items,
                                                const double* __restrict__ tabPool,
                                                const int32_t* __restrict__ li, const int32_t* __restrict__ lj,
                                                const double* __restrict__ lza, const double* __restrict__ lzb,
                                                uint32_t* __restrict__ rowCnt,
                                                unsigned long long* __restrict__ maskPool,
                                                uint32_t* __restrict__ prefPool,
                                                int TC /* LDS column tile (multiple of 256) */, int ldsPerWave /* doubles: NR table slices */, int RPB,
                                                const uint16_t* __restrict__ qtabPool /* PRE: the tables as bins (k_tables) */,
                                                const double* __restrict__ feats, int NO /* PRE: object capacity per map of the coordinate tile in LDS (0: none — the exact gate reads the tables) */)
{
    // LDS: [GM: cZZ[TC]] cIJ[TC] | per wave NR table slices (n1 + 1 + n2 doubles each)
    // PRE (NR is ignored: PRE_NR rows per wave; ldsPerWave = entries of the packed bin table): cIJ[TC] | per wave count_pre_wave_bytes()
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2* cZZ = reinterpret_cast<double2*>(smem);
    uint32_t* cIJ = reinterpret_cast<uint32_t*>(cZZ + ((GM && !PRE) ? TC : 0));
    double* tabs = reinterpret_cast<double*>(cIJ + TC);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, w = tid >> 6, wpb = nt >> 6;
    double* tA = tabs + (size_t)w * ldsPerWave;
    // (PRE) no f64 table slices at all: per wave the mask rows, the packed bin table, the queue, the rows' data
    const int preBytes = PRE ? count_pre_wave_bytes(ldsPerWave, TC) : 0;
    uint32_t* qctr = cIJ + TC + PRE_COLPAD;                      // (PRE) the rows' hand-out counter, 16 bytes
    uint32_t* degL = cIJ + TC + PRE_COLPAD + 4;                  // (PRE, whole problems) TC row degrees
    unsigned char* pw = reinterpret_cast<unsigned char*>(cIJ + TC + PRE_COLPAD + 4 + ((PRE && RPB < 0) ? TC : 0)) + (size_t)w * preBytes;
    unsigned long long* rmask = reinterpret_cast<unsigned long long*>(pw);
    uint2* qT = reinterpret_cast<uint2*>(rmask + PRE_NR * (TC >> 6));
    uint16_t* queue = reinterpret_cast<uint16_t*>(qT + ldsPerWave);
    PreRow* rowinfo = reinterpret_cast<PreRow*>(queue + PRE_QCAP);
    // (PRE) behind the waves' regions: the objects' coordinates of the item's problem, 32 bytes per object
    const size_t sOoff = ((size_t)(reinterpret_cast<unsigned char*>(cIJ + TC + PRE_COLPAD + 4 + ((PRE && RPB < 0) ? TC : 0)) - smem) + (size_t)wpb * (size_t)preBytes + 15) & ~(size_t)15;
    dbl2_t* sO = reinterpret_cast<dbl2_t*>(smem + sOoff);
    if (PRE) { for (int x = lane; x < PRE_NR * (TC >> 6); x += WAVE) rmask[x] = 0ull; }
    // PRE with RPB < 0: a work item is a whole PROBLEM (B = -RPB problems; batches with at least a problem per compute unit): the
    // column tile is staged once per problem instead of once per 128 rows, and the rows go to the waves quad by quad
    const bool whole = PRE && RPB < 0;
    const int nItems = whole ? -RPB : tot->items;
    // XCD-aware order (workgroup ids are dealt to the 8 XCDs round-robin): XCD x takes the CONTIGUOUS range [x Gx, (x + 1) Gx) of
    // work items — the items of a problem, which share its tables, columns and mask rows, run on one L2
    const int Gx_ = (nItems + 7) >> 3;
    for (int sIdx_ = blockIdx.x; (sIdx_ >> 3) < Gx_; sIdx_ += gridDim.x) {
        // (whole problems: problem t on the XCD t mod 8, where k_lists — one workgroup per problem, b = blockIdx — reads its mask rows)
        const int t = whole ? sIdx_ : (sIdx_ & 7) * Gx_ + (sIdx_ >> 3);
        if (t >= nItems) continue;
        ItemDesc it;
        if (whole) { it.b = t; it.row0 = 0; if (st[t].kind >= 2 || st[t].L <= 0) continue; } else it = items[t];
        const int b = it.b;
        const ProbDesc pd = probs[b];
        const int L = st[b].L;
        const int64_t lo = pd.liveOff, mo = st[b].maskOff;
        const int nrows = whole ? L : min(RPB, L - it.row0);
        const double* TA = tabPool + pd.tabOff;
        const double* TB = TA + (int64_t)pd.n1 * pd.n1;
        const int Lpad = (L + 255) & ~255;
        // The column data of the whole live set in LDS when it fits (the usual case); a larger live set (no semantic gate:
        // L = n1 * n2) is swept TILE BY TILE — the item's rows are staged once per tile (a row needs only the words at and
        // behind its own 64-row block: tiles in front of the item's first row are skipped).
        // TILED == false: the items whose whole live set fits the LDS column tile (the usual case: one sweep per row);
        // TILED == true (a second launch, only when such problems can exist): the others, tile by tile.
        if ((Lpad <= TC) == TILED) continue;
        if (!TILED) {
            __syncthreads();                    // every wave is done with the previous item's columns
            for (int q = tid; q < Lpad + (PRE ? PRE_COLPAD : 0); q += nt) {     // (PRE: sentinel columns up to the end of a sweep's last step)
                const bool v = q < L;
                cIJ[q] = v ? ((uint32_t)li[lo + q] | ((uint32_t)(pd.n1 + 1 + lj[lo + q]) << 16)) : ((uint32_t)pd.n1 | ((uint32_t)(pd.n1 + 1) << 16));
                if (GM && !PRE) cZZ[q] = v ? make_double2(lza[lo + q], lzb[lo + q]) : make_double2(0.0, 0.0);
            }
            if (PRE && NO > 0) {
                const int pdim = D.p.point_dim;
                for (int o = tid; o < pd.n1 + pd.n2; o += nt) {
                    const double* f = feats + (o < pd.n1 ? pd.off1 + o : pd.off2 + (o - pd.n1)) * D.F;
                    const int slot = o < pd.n1 ? o : NO + (o - pd.n1);
                    sO[2 * slot] = dbl2_t{f[0], pdim > 1 ? f[1] : 0.0};
                    sO[2 * slot + 1] = dbl2_t{pdim > 2 ? f[2] : 0.0, 0.0};
                }
            }
            if (PRE && tid == 0) *qctr = 0u;
            if (PRE && whole) for (int q = tid; q < L; q += nt) degL[q] = 0u;
            __syncthreads();
            if (PRE) {
                count_rows_pre<GM>(D, pd, L, it.row0, nrows, qctr, lane, cIJ, TA, TB,
                                   qtabPool + 4 * (int64_t)pd.qtabOff4, qtabPool + 4 * (int64_t)pd.qtabOff4 + (int64_t)pd.n1 * ((pd.n1 + 3) & ~3),
                                   qT, queue, rmask, TC >> 6, rowinfo, maskPool + mo, lza + lo, lzb + lo, whole ? degL : nullptr, sO, NO);
                if (whole) {                                    // the problem's degrees, for k_lists (live order)
                    __syncthreads();
                    for (int q = tid; q < L; q += nt) rowCnt[lo + q] = degL[q];
                }
            }
            else
            count_rows_lds<GM, NR, false>(D, pd, L, it.row0, nrows, w, wpb, lane, cIJ, cZZ, TA, TB, tA, ldsPerWave / NR, maskPool + mo,
                                          0, Lpad, li + lo, lj + lo, lza + lo, lzb + lo);
        } else {
            count_item_tiled<GM, NR>(D, pd, L, it.row0, nrows, cIJ, cZZ, TA, TB, tA, ldsPerWave / NR, maskPool + mo, TC,
                                     li + lo, lj + lo, lza + lo, lzb + lo);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_mirror: lower triangle of the candidate bit matrix.  Block (c, R) of 64x64 bits, c > R, is the
// transpose of block (R, c) that k_count produced: lane l loads word c of row R*64+l, the wave
// transposes the 64x64 bit block in registers (6 butterfly stages) and lane l stores word R of row
// c*64+l.  (The earlier scheme, one 8-byte atomic OR per hit from inside k_count, cost 0.25 ms per
// batch of 256; this kernel takes 0.13 ms and needs no zeroed mask pool.)
// A wave owns 1 x 8 blocks: it loads the words c..c+7 of the 64 rows of source row block R (one cache line per
// row for all 8 loads, all in flight together).  The 8 waves of a workgroup take 8 consecutive R for the same
// c range: their stores to words R..R+7 of a target row fall into one line and are merged by the write-back L2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, int lane)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned long long m = (s == 32) ? 0x00000000ffffffffull : (s == 16) ? 0x0000ffff0000ffffull : (s == 8) ? 0x00ff00ff00ff00ffull
                                   : (s == 4) ? 0x0f0f0f0f0f0f0f0full : (s == 2) ? 0x3333333333333333ull : 0x5555555555555555ull;
        const unsigned long long o = __shfl_xor(x, s);
        x = (lane & s) ? ((x & ~m) | ((o & ~m) >> s)) : ((x & m) | ((o & m) << s));
    }
    return x;
}

__global__ void __launch_bounds__(512) k_mirror(int B, int T /* workgroups per problem */, const ProbState* __restrict__ st, unsigned long long* __restrict__ maskPool, int skip0)
{
    // The bit matrices are SPARSE (1.8 set bits per 64-bit word at config 3): a block is transposed by scattering its set
    // bits — lane l (source row) ORs bit l into word j of the wave's LDS tile for every set bit j of its word, one LDS atomic
    // per set bit — instead of the dense 6-stage butterfly (transpose64: ~130 instructions per block whatever it holds;
    // this kernel was bound by them, 153 us per batch of 256).  LDS operations of one wave execute in order: no barriers.
    __shared__ unsigned long long tile[8][8][64];                 // per wave: 8 blocks x 64 target words
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // All T workgroups of a problem carry ids that are congruent modulo 8: one XCD, one L2 (k_cos).  A 128-byte line of the
    // lower triangle collects its 16 words from 16 different tasks; dealt round-robin over the 8 non-coherent L2s every one
    // of them wrote its part of the line back separately (2.2 x the bytes, read-modify-write in HBM: half this kernel's time).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / T) * 8 + xcd, tb = slot % T;
    if (b >= B) return;
    if (st[b].kind >= 2 || (skip0 && st[b].kind == 0)) return;    // (skip0: the stream-layout problems take k_lists — no symmetric matrix needed)
    const int L = st[b].L;
    const int W = (L + 63) >> 6;
    const int nR = ((W - 1 + nw - 1) / nw) * nw;                  // source row blocks 0..W-2, padded to whole workgroups
    const int nTasks = ((W + 7) / 8) * max(nR, 1);                // (column strip of 8 words) x (source row block)
    unsigned long long* mb = maskPool + st[b].maskOff;
    // the grid was sized from an ESTIMATE of the largest live set: a larger problem takes several rounds
    for (int task = tb * nw + w; task < nTasks; task += T * nw) {
        const int ct = task / max(nR, 1), R = task - ct * nR;
        const int cb = ct * 8;
        if (R >= W - 1 || cb >= W || cb + 7 <= R) continue;       // nothing below the diagonal in this strip
        const unsigned long long* src = mb + (int64_t)(R * 64 + lane) * W;         // rows R*64+lane < (W-1)*64 < L exist
        unsigned long long x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (cb + i > R && cb + i < W) ? src[cb + i] : 0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[w][i][lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long me = 1ull << lane;
        for (;;) {
            unsigned long long left = 0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (x[i]) {
                    const int j = __builtin_ctzll(x[i]);
                    (void)__hip_atomic_fetch_or(&tile[w][i][j], me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    x[i] &= x[i] - 1ull;
                }
                left |= x[i];
            }
            if (__ballot(left != 0ull) == 0ull) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cb + i;
            if (c > R && c < W) {                                // wave-uniform
                const unsigned long long y = tile[w][i][lane];
                if (c * 64 + lane < L) mb[(int64_t)(c * 64 + lane) * W + R] = y;
            }
        }
        __builtin_amdgcn_wave_barrier();                         // (the next task zeroes the tile behind these reads: in order)
    }
}

// ---------------------------------------------------------------------------------------------
// k_rowprefix: per live row, the number of candidates in front of every 64-column mask word (the entry
// index of the word's first candidate in k_fill) and the row total.  One wave per row, lanes = words.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_rowprefix(const ProbDesc* __restrict__ probs, const ProbState* __restrict__ st,
                                                    const BatchTotals* __restrict__ tot, const ItemDesc* __restrict__ items,
                                                    const unsigned long long* __restrict__ maskPool, uint32_t* __restrict__ prefPool,
                                                    uint32_t* __restrict__ rowCnt, int RPB, int skip0 /* stream-layout problems take k_lists */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int nItems = tot->items;
    // XCD-aware order (workgroup ids are dealt to the 8 XCDs round-robin): XCD x takes the CONTIGUOUS range [x Gx, (x + 1) Gx) of
    // work items — the items of a problem, which share its tables, columns and mask rows, run on one L2
    const int Gx_ = (nItems + 7) >> 3;
    for (int sIdx_ = blockIdx.x; (sIdx_ >> 3) < Gx_; sIdx_ += gridDim.x) {
        const int t = (sIdx_ & 7) * Gx_ + (sIdx_ >> 3);
        if (t >= nItems) continue;
        const ItemDesc it = items[t];
        const int b = it.b;
        const int L = st[b].L;
        const int W = (L + 63) >> 6;
        const int64_t lo = probs[b].liveOff, mo = st[b].maskOff;
        const int nrows = min(RPB, L - it.row0);
        const bool wantPrefix = st[b].kind != 0;               // the stream layout takes its prefix counts in k_upper
        if (skip0 && !wantPrefix) continue;
        if (W <= WAVE) {                                        // one word per lane: four rows per step, their loads in flight together
            constexpr int U = 4;
            for (int r = w; r < nrows; r += U * wpb) {
                uint32_t c[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = it.row0 + min(r + u * wpb, nrows - 1);
                    c[u] = (lane < W) ? (uint32_t)__popcll(maskPool[mo + (int64_t)k * W + lane]) : 0u;
                }
                uint32_t inc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) inc[u] = wave_incl_scan(c[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r + u * wpb < nrows) {
                        const int k = it.row0 + r + u * wpb;
                        if (wantPrefix && lane < W) prefPool[mo + (int64_t)k * W + lane] = inc[u] - c[u];
                        if (lane == WAVE - 1) rowCnt[lo + k] = inc[u];
                    }
                }
            }
            continue;
        }
        for (int r = w; r < nrows; r += wpb) {
            const int k = it.row0 + r;
            const unsigned long long* mrow = maskPool + mo + (int64_t)k * W;
            uint32_t* prow = prefPool + mo + (int64_t)k * W;
            uint32_t carry = 0;
            for (int wb = 0; wb < W; wb += WAVE) {
                const bool v = wb + lane < W;
                const uint32_t c = v ? (uint32_t)__popcll(mrow[wb + lane]) : 0u;
                const uint32_t ex = wave_excl_scan(c, lane);
                if (v && wantPrefix) prow[wb + lane] = carry + ex;
                carry += __shfl(ex + c, WAVE - 1);
            }
            if (lane == 0) rowCnt[lo + k] = carry;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_rowsort: per problem, the live rows ordered by descending candidate count (degree).
//  kind 0 (stream layout): a STABLE, deterministic order — rank(k) = #{rows with a larger degree} + #{rows k' < k with
//    the same degree} — because the rank becomes the association's POSITION, the numbering of the stored matrix
//    (which entries a row keeps).  Bitonic sort of unique (degree, row) keys in LDS.  Slice geometry follows in
//    k_slicegeom, once k_upper knows the upper degrees.
//  kind 1 (fallback): counting sort with atomic ranks (the order among rows of equal degree is arbitrary — it
//    changes where a row is stored, never what is computed for it), then the sorted SELL-64 geometry:
//    rowPos[k] = sorted position of row k, perm[pos] = row, per slice its width (longest row) and base offset.
// ---------------------------------------------------------------------------------------------
// Positions of the stream layout WITHOUT a sort.  The position of row k is its rank by (degree descending, row ascending):
//   #(rows of larger degree) + #(rows of the same degree and smaller index).
// Degrees are small integers: a histogram (LDS atomics), its scan in descending degree order (= where every degree's range of
// positions starts), a scatter of the rows into their degree's range in whatever order the atomics grant, and the rank of a row inside
// its range by counting the smaller row indices there (a dozen rows share a degree on average).  keys[0 .. L) receives what the
// bitonic sort of the unique keys ((degree + 1) << 12) | (4095 - row) left there — the same array, bit for bit — and zeros up to N.
// The bitonic sort it stands for was 67 k cycles of k_lists' 390 k per problem and 30 us of the single-pair call (k_rowsort).
// Returns false (nothing written that matters) when more than eqMax rows share one degree — the ranks inside a range cost its
// square —: the caller sorts.  deg(k): degree of live row k (< L).  wsum: nt / 64 + 2 words.
template <class DegF>
__device__ __forceinline__ bool place_keys(int L, int N, int eqMax, DegF deg, uint32_t* keys, uint32_t* hcnt /* [L] */, uint32_t* hcur /* [L] */,
                                           uint16_t* tmp /* [L] */, uint32_t* wsum)
{
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6, nw = nt >> 6;
    for (int d = tid; d < L; d += nt) hcnt[d] = 0u;
    if (tid == 0) { wsum[nw] = 0u; wsum[nw + 1] = 0u; }           // carry of the scan, largest group of equal degrees
    __syncthreads();
    for (int k = tid; k < L; k += nt) atomicAdd(&hcnt[deg(k)], 1u);
    __syncthreads();
    uint32_t mx = 0u;
    for (int r0 = 0; r0 < L; r0 += nt) {                           // exclusive scan over the bins, LARGEST degree first
        const int r = r0 + tid;
        const uint32_t v = r < L ? hcnt[L - 1 - r] : 0u;
        mx = max(mx, v);
        const uint32_t inc = wave_incl_scan(v);
        if (lane == WAVE - 1) wsum[w] = inc;
        __syncthreads();
        uint32_t wbase = 0u, tot = 0u;
        for (int t = 0; t < nw; ++t) { if (t < w) wbase += wsum[t]; tot += wsum[t]; }
        const uint32_t carry = wsum[nw];
        if (r < L) hcur[L - 1 - r] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 0) wsum[nw] = carry + tot;
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    if (lane == 0) atomicMax(&wsum[nw + 1], mx);
    __syncthreads();
    if ((int)wsum[nw + 1] > eqMax) return false;                   // (the same word in every thread)
    for (int k = tid; k < L; k += nt) { const uint32_t slot = atomicAdd(&hcur[deg(k)], 1u); tmp[slot] = (uint16_t)k; }   // hcur[d] ends as the END of d's range
    for (int t = L + tid; t < N; t += nt) keys[t] = 0u;
    __syncthreads();
    for (int i = tid; i < L; i += nt) {
        const uint32_t k = tmp[i], d = (uint32_t)deg((int)k);
        const uint32_t end = hcur[d], start = end - hcnt[d];
        uint32_t r = 0u;
        for (uint32_t j = start; j < end; j += 4u) {
            uint32_t o[4];
#pragma unroll
            for (uint32_t x = 0; x < 4u; ++x) o[x] = tmp[min(j + x, end - 1u)];
#pragma unroll
            for (uint32_t x = 0; x < 4u; ++x) r += (j + x < end && o[x] < k) ? 1u : 0u;
        }
        keys[start + r] = ((d + 1u) << 12) | (4095u - k);
    }
    __syncthreads();
    return true;
}

constexpr int SORT_KEYS = 8192;

__global__ void __launch_bounds__(1024) k_rowsort(const ProbDesc* __restrict__ probs, ProbState* __restrict__ st,
                                                  BatchTotals* __restrict__ tot,
                                                  const uint32_t* __restrict__ rowCnt,
                                                  uint32_t* __restrict__ rowPos, uint32_t* __restrict__ perm,
                                                  uint32_t* __restrict__ sliceWidth, uint32_t* __restrict__ sliceBase,
                                                  uint32_t* __restrict__ listOff, long long capList, int skip0 /* stream-layout problems take k_lists */,
                                                  int eqMax /* place_keys(): rows of one degree at most; 0: always the bitonic sort */)
{
    // (static LDS of this kernel: ~70 KB — hist 32 KB, place_keys()' tables 24 KB, rows and degrees 12 KB.  gfx950 has 160 KB per
    //  workgroup; the 64 KB parts before it are NOT a target of this library: see the static_assert below and the Makefile)
    static_assert(sizeof(uint32_t) * (SORT_KEYS + 2 * STREAM_MAXL) + sizeof(uint16_t) * 2 * STREAM_MAXL <= 160 * 1024, "k_rowsort's tables must fit the LDS of a gfx950 workgroup");
    __shared__ uint32_t hist[SORT_KEYS];         // indexed by SORT_KEYS-1-key: ascending index = descending count
    __shared__ uint32_t wsum[20];
    __shared__ uint32_t hcntS[STREAM_MAXL], hcurS[STREAM_MAXL];     // place_keys(): rows per degree, cursor / end of the degree's range
    __shared__ uint16_t tmpS[STREAM_MAXL], dgS[STREAM_MAXL];        // ... the rows in their ranges (any order), the degrees
    __shared__ uint32_t carry_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nt = blockDim.x, nw = nt >> 6;
    const int L = st[b].L;
    const int64_t lo = probs[b].liveOff;
    if (st[b].kind >= 2) return;
    if (st[b].kind == 0 && skip0) return;
    if (st[b].kind == 0) {
        // stream layout: bitonic sort (descending) of the UNIQUE keys ((degree + 1) << 12) | (4095 - row): larger degree
        // first, equal degrees in row order — the rank is the row's position.  N = next power of two >= L keys in LDS
        // (padding keys 0 sort last), log2(N)(log2(N)+1)/2 <= 78 compare-exchange stages of N/2 pairs.
        static_assert(STREAM_MAXL <= 4096 && SORT_KEYS >= 4096, "bitonic sort capacity");
        int N = 64; while (N < L) N <<= 1;
        for (int t = tid; t < L; t += nt) dgS[t] = (uint16_t)rowCnt[lo + t];
        __syncthreads();
        // (round 5: the sorted key array without the sort — place_keys() above; the sort remains for live sets in which very many rows share a degree)
        if (eqMax <= 0 || !place_keys(L, N, eqMax, [&](int k) { return (uint32_t)dgS[k]; }, hist, hcntS, hcurS, tmpS, wsum)) {
        for (int t = tid; t < N; t += nt) hist[t] = (t < L) ? ((((uint32_t)dgS[t] + 1u) << 12) | (uint32_t)(4095 - t)) : 0u;
        __syncthreads();
        for (int k2 = 2; k2 <= N; k2 <<= 1)
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (int t = tid; t < (N >> 1); t += nt) {
                    const int i1 = ((t & ~(j2 - 1)) << 1) | (t & (j2 - 1)), i2 = i1 | j2;     // the pair (i1, i1 ^ j2)
                    const uint32_t a = hist[i1], c2 = hist[i2];
                    const bool desc = (i1 & k2) == 0;                                         // direction of this bitonic block
                    if ((a < c2) == desc) { hist[i1] = c2; hist[i2] = a; }
                }
                __syncthreads();
            }
        }
        for (int q = tid; q < L; q += nt) {
            const uint32_t k = 4095u - (hist[q] & 4095u);
            perm[lo + q] = k; rowPos[lo + k] = (uint32_t)q;
        }
        // Candidate lists (k_upper writes them, k_fill_list reads them): position q keeps at most
        // min(degree, L - 1 - q) of its candidates — room for that many 16-bit column indices, rounded up to whole
        // quads, at listOff[q] behind the problem's base.  The base comes from a bump pointer shared by the batch (where
        // a problem's lists lie is immaterial); a problem whose lists would end beyond the pool becomes kind 2.
        {
            const int PER = (L + nt - 1) / nt;                  // consecutive positions per thread (<= 3)
            uint32_t sum = 0;
            for (int t = 0; t < PER; ++t) {
                const int q = tid * PER + t;
                if (q < L) sum += (min((hist[q] >> 12) - 1u, (uint32_t)(L - 1 - q)) + 3u) & ~3u;
            }
            const uint32_t inc = wave_incl_scan(sum);
            if (lane == WAVE - 1) wsum[w] = inc;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
            for (int t = 0; t < nw; ++t) { if (t < w) wbase += wsum[t]; total += wsum[t]; }
            if (tid == 0) {
                const unsigned long long base = atomicAdd(&tot->listTop, (unsigned long long)total);
                st[b].listOff = (int64_t)base;
                if ((long long)(base + total) > capList) { st[b].kind = 2; atomicAdd(&tot->overflow, 1); }
            }
            uint32_t run = wbase + inc - sum;
            for (int t = 0; t < PER; ++t) {
                const int q = tid * PER + t;
                if (q < L) { listOff[lo + q] = run; run += (min((hist[q] >> 12) - 1u, (uint32_t)(L - 1 - q)) + 3u) & ~3u; }
            }
        }
        return;                                                 // slice geometry of the stream layout: k_slicegeom
    }
    for (int t = tid; t < SORT_KEYS; t += nt) hist[t] = 0;
    __syncthreads();
    for (int k = tid; k < L; k += nt) atomicAdd(&hist[SORT_KEYS - 1 - min(rowCnt[lo + k], (uint32_t)(SORT_KEYS - 1))], 1u);
    __syncthreads();
    {   // exclusive scan of hist (8 consecutive bins per thread)
        constexpr int PER = SORT_KEYS / 1024;
        uint32_t loc[PER]; uint32_t sum = 0;
        for (int t = 0; t < PER; ++t) { loc[t] = hist[tid * PER + t]; sum += loc[t]; }
        const uint32_t inc = wave_incl_scan(sum);
        if (lane == WAVE - 1) wsum[w] = inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (int t = 0; t < w; ++t) wbase += wsum[t];
        uint32_t run = wbase + inc - sum;
        for (int t = 0; t < PER; ++t) { hist[tid * PER + t] = run; run += loc[t]; }
    }
    __syncthreads();
    for (int k = tid; k < L; k += nt) {
        const uint32_t pos = atomicAdd(&hist[SORT_KEYS - 1 - min(rowCnt[lo + k], (uint32_t)(SORT_KEYS - 1))], 1u);
        rowPos[lo + k] = pos; perm[lo + pos] = (uint32_t)k;
    }
    __syncthreads();       // block-scope: perm visible to the whole workgroup
    const int nsl = (L + 63) >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int s0 = 0; s0 < nsl; s0 += nt) {
        const int s = s0 + tid;
        uint32_t width = 0;
        if (s < nsl) {
            const int hi = min(L, (s << 6) + 64);
            for (int p = s << 6; p < hi; ++p) width = max(width, rowCnt[lo + perm[lo + p]]);
            width = (width + 3u) & ~3u;                         // quad layout: 4 entries per lane and step
            sliceWidth[lo + s] = width;
        }
        const uint32_t v = width * 64u;
        const uint32_t inc = wave_incl_scan(v);
        if (lane == WAVE - 1) wsum[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int t = 0; t < nw; ++t) { if (t < w) wbase += wsum[t]; tot += wsum[t]; }
        const uint32_t carry = carry_s;
        if (s < nsl) sliceBase[lo + s] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0) st[b].nnzCap = carry_s;
}

// ---------------------------------------------------------------------------------------------
// k_upper (kind 0): which candidates row p of the stored matrix keeps.  The matrix is stored once per unordered pair:
// candidate (k, q) of the symmetric bit matrix stays in the row of the endpoint with the SMALLER position.  Row p of the
// result is row k = perm[p] of the live-order matrix with the bits q dropped whose position is not larger than p —
// the COLUMNS stay in live order (the fill kernel labels an entry with its column's position; the order of the entries
// inside a row is immaterial: the solver's sums are exact).  One wave per row: the row's words sit in its lanes (lane =
// word); for word w the lanes compare the positions of columns 64w..64w+63 (LDS table) with p, the ballot is ANDed
// onto the word (consecutive rows update the masks instead of rebuilding them).  The per-word prefix counts (entry index of a word's first kept candidate in k_fill_slice), the
// row's upper degree and the position-ordered copies of the pools the solver reads come out of the same pass.
// ---------------------------------------------------------------------------------------------
struct LivePools { int32_t* lp; int32_t* li; int32_t* lj; double* ls; double* ld; double* lza; double* lzb; };

__global__ void __launch_bounds__(1024) k_upper(const ProbDesc* __restrict__ probs, const ProbState* __restrict__ st,
                                                const BatchTotals* __restrict__ tot, const ItemDesc* __restrict__ items,
                                                const unsigned long long* __restrict__ maskPool,
                                                uint16_t* __restrict__ listPool, const uint32_t* __restrict__ listOff,
                                                uint32_t* __restrict__ rowCnt, const uint32_t* __restrict__ perm,
                                                const uint32_t* __restrict__ rowPos, LivePools src, LivePools dst, int RPB)
{
    __shared__ uint16_t posS[STREAM_MAXL + 64];                  // position of every live column (padding: 0 = never kept)
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6, wpb = nt >> 6;
    const int nItems = tot->items;
    int staged = -1;
    // XCD-aware order (workgroup ids are dealt to the 8 XCDs round-robin): XCD x takes the CONTIGUOUS range [x Gx, (x + 1) Gx) of
    // work items — the items of a problem, which share its tables, columns and mask rows, run on one L2
    const int Gx_ = (nItems + 7) >> 3;
    for (int sIdx_ = blockIdx.x; (sIdx_ >> 3) < Gx_; sIdx_ += gridDim.x) {
        const int t = (sIdx_ & 7) * Gx_ + (sIdx_ >> 3);
        if (t >= nItems) continue;
        const ItemDesc it = items[t];
        const int b = it.b;
        if (uni_i(st[b].kind) == 0) {                           // (no `continue` around the barriers below)
        const int L = st[b].L;
        const int W = (L + 63) >> 6;
        const int64_t lo = probs[b].liveOff, mo = st[b].maskOff;
        uint16_t* const lists = listPool + st[b].listOff;
        const int nrows = min(RPB, L - it.row0);
        if (staged != b) {                                       // consecutive items of a workgroup often share the problem
            __syncthreads();
            for (int q = tid; q < W * 64; q += nt) posS[q] = (q < L) ? (uint16_t)rowPos[lo + q] : (uint16_t)0;
            __syncthreads();
            staged = b;
        }
        // A wave takes CONSECUTIVE rows: the set of columns behind p loses exactly one member, column perm[p + 1], when p
        // moves on by one — the "behind" masks of all words are built once (lane = word) and then only updated.
        const int RW = (nrows + wpb - 1) / wpb;
        const int r0 = w * RW, r1 = min(nrows, r0 + RW);
        if (r0 < r1) {
            const int p0 = it.row0 + r0;
            uint32_t glo = 0u, ghi = 0u;                         // lane wd: columns of word wd with a position > p
            for (int wd = 0; wd < W; ++wd) {
                const unsigned long long gt = __ballot((int)posS[(wd << 6) + lane] > p0);
                writelane2(glo, ghi, gt, (uint32_t)__builtin_amdgcn_readfirstlane(wd));
            }
            unsigned long long behind = ((unsigned long long)ghi << 32) | glo;
            // the wave's rows: live index and list offset of row r0 + l in lane l (read back with v_readlane), and the
            // position-ordered copies of the pools the solver reads; the mask row of the NEXT row is requested before
            // the current one is worked on, so no memory round trip sits between two rows
            const int nr = r1 - r0;                              // <= RPB / waves per block <= 64
            const int kv = (lane < nr) ? (int)perm[lo + p0 + lane] : 0;
            const uint32_t ov = (lane < nr) ? listOff[lo + p0 + lane] : 0u;
            if (lane < nr) { dst.lp[lo + p0 + lane] = src.lp[lo + kv]; dst.ld[lo + p0 + lane] = src.ld[lo + kv]; }
            unsigned long long m_n = (lane < W) ? maskPool[mo + (int64_t)__builtin_amdgcn_readlane(kv, 0) * W + lane] : 0ull;
            for (int i = 0; i < nr; ++i) {
                const int p = p0 + i;                            // position of the row being written
                const unsigned long long raw = m_n;
                int kn = 0;
                if (i + 1 < nr) {
                    kn = __builtin_amdgcn_readlane(kv, i + 1);
                    m_n = (lane < W) ? maskPool[mo + (int64_t)kn * W + lane] : 0ull;
                }
                const unsigned long long mine = raw & behind;
                const uint32_t c = (uint32_t)__popcll(mine);
                const uint32_t ex = wave_excl_scan(c, lane);
                const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)(ex + c), WAVE - 1);
                {   // the row's candidate list: live column indices in ascending order, padded to a whole quad
                    uint16_t* lst = lists + (uint32_t)__builtin_amdgcn_readlane((int)ov, i);
                    unsigned long long m = mine; uint32_t e = ex;
                    while (m) { lst[e++] = (uint16_t)((lane << 6) + __builtin_ctzll(m)); m &= m - 1ull; }
                    if (lane < 4 && cnt + (uint32_t)lane < ((cnt + 3u) & ~3u)) lst[cnt + lane] = (uint16_t)0xffffu;
                }
                if (lane == 0) rowCnt[lo + p] = cnt;             // upper degree of position p (the live-order degrees are spent)
                if (i + 1 < nr && lane == (kn >> 6)) behind &= ~(1ull << (kn & 63));   // the column at position p + 1 is no longer behind
            }
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_lists (kind 0; launches in which no fallback problem can occur): from the UPPER blocks of the candidate bit matrix — what
// k_count writes — straight to positions and kept-candidate lists, one workgroup per problem, everything that is looked up by
// index in LDS.  It stands for k_mirror + k_rowprefix + k_rowsort + k_upper, which reach the same lists through the symmetric
// matrix (0.55 MB per problem written and read again, four dependent launches: 0.31 ms of a 2.0 ms step at config 3):
//   degrees    a stored bit (k, q), k < q, counts for row k (popcount of the row's words) and for row q (one LDS atomic per
//              set bit): the full degree of the symmetric matrix without ever forming it;
//   positions  rank by (degree descending, row ascending): k_rowsort's bitonic sort of unique keys, in LDS;
//   lists      room for position p as k_rowsort sizes it (min(degree, L - 1 - p) entries, whole quads, bump pointer of the
//              batch); a second sweep over the bits hands every pair to its endpoint of SMALLER position (slot by an LDS
//              cursor, the other endpoint's live column index as the entry), then the rows are padded to whole quads and their
//              upper degrees written in position order with the position-ordered copies of the pools the solver reads.
// The ENTRIES of a list arrive in the order the cursors were taken, not ascending: where an entry sits in its row is immaterial
// (the fill labels it with its column's position, the solver's sums are exact integers) — roman_get_upper_csr sorts its rows.
// Sweep: 16 lanes per row, four rows per wave step, lane = one 64-column word of the row's upper part (the diagonal word keeps
// its bits behind the row's own); a row's words come in one or two coalesced 128-byte pieces.
// ---------------------------------------------------------------------------------------------
constexpr int LISTS_NT = 1024;

__global__ void __launch_bounds__(LISTS_NT) k_lists(int B, const ProbDesc* __restrict__ probs, ProbState* __restrict__ st, BatchTotals* __restrict__ tot,
                                                   const unsigned long long* __restrict__ maskPool,
                                                   uint16_t* __restrict__ listPool, uint32_t* __restrict__ listOff,
                                                   uint32_t* __restrict__ rowCnt, uint32_t* __restrict__ perm, uint32_t* __restrict__ rowPos,
                                                   LivePools src, LivePools dst, long long capList,
                                                   int eqMax /* place_keys(): rows of one degree at most; 0: always the bitonic sort */,
                                                   int degGiven /* rowCnt holds every live row's full degree (k_count counted the pairs as they passed: whole problems): no degree sweep */)
{
    __shared__ uint32_t degS[STREAM_MAXL + 64];                 // full degree of a live row; after the sort: the row's list cursor
    __shared__ uint32_t keyS[4096];                             // sort keys
    __shared__ unsigned long long tabS[STREAM_MAXL + 64];       // per live row: list offset << 16 | position (one LDS read serves both); before: place_keys()' two tables
    __shared__ uint16_t tmpS[STREAM_MAXL + 64];                 // place_keys(): the rows in their degree's range
    __shared__ uint32_t wsum[LISTS_NT / 64 + 4];
    __shared__ int okS;
    static_assert(STREAM_MAXL <= 4096, "sort capacity");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NW = LISTS_NT / 64;
    const int g = lane >> 4, sl = lane & 15;                    // row of the wave step, word of the row's piece
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                        // (the previous problem's LDS is no longer read)
        if (st[b].kind != 0) continue;
        const int L = st[b].L;
        if (L <= 0) continue;
        const int W = (L + 63) >> 6;
        const int64_t lo = probs[b].liveOff, mo = st[b].maskOff;
        const unsigned long long* mrow = maskPool + mo;
        // f(k, m, c): word c (strictly-upper bits only) of live row k, for every row and every word at or behind its diagonal one.
        // A wave step = four rows (one 64-row block: k0 is a multiple of four) x sixteen words from c0 on; the words of the next
        // six steps are in flight while one is worked on (unconditional loads at clamped addresses: the wait counters stay exact)
        auto fetch = [&](int k0_, int c0_) -> unsigned long long {
            const int k = k0_ + g, c = c0_ + sl;
            const bool valid = k < L && c < W;                  // (c >= k >> 6 by construction: c0 starts at the rows' diagonal word)
            unsigned long long m = mrow[(int64_t)min(k, L - 1) * W + min(c, W - 1)];
            if (!valid) m = 0ull;
            if (c == (k >> 6)) m &= ((k & 63) == 63) ? 0ull : (~0ull << ((k & 63) + 1));   // the diagonal block holds both triangles: columns behind k only
            return m;
        };
        auto advance = [&](int& k0_, int& c0_) { c0_ += 16; if (c0_ >= W) { k0_ += 4 * NW; c0_ = k0_ >> 6; } };
        auto sweep = [&](auto f) {
            constexpr int PF = 6;                               // steps in flight: a wave walks ~50 steps, each a dependent round trip otherwise
            int kr[PF], cr[PF]; unsigned long long mr[PF];
            int kn = 4 * w, cn = kn >> 6;
#pragma unroll
            for (int i = 0; i < PF; ++i) { kr[i] = kn; cr[i] = cn; mr[i] = fetch(kn, cn); advance(kn, cn); }
            while (kr[0] < L) {                                 // (steps are in ascending row order; a step behind the last row carries no bits)
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int kc = kr[i], cc = cr[i]; const unsigned long long mc = mr[i];
                    kr[i] = kn; cr[i] = cn; mr[i] = fetch(kn, cn); advance(kn, cn);
                    f(kc + g, mc, cc + sl);
                }
            }
        };
#ifdef ROMAN_LISTS_TIMING
        unsigned long long tl_[6]; tl_[0] = __builtin_readcyclecounter();
#define LMARK(i_) tl_[i_] = __builtin_readcyclecounter()
#else
#define LMARK(i_) do { } while (0)
#endif
        for (int q = tid; q < L; q += LISTS_NT) degS[q] = degGiven ? rowCnt[lo + q] : 0u;
        __syncthreads();
        // ---- degrees ----
        if (!degGiven) {
        sweep([&](int k, unsigned long long m, int c) {
            if (m) {
                atomicAdd(&degS[k], (uint32_t)__popcll(m));
                while (m) { atomicAdd(&degS[(c << 6) + __builtin_ctzll(m)], 1u); m &= m - 1ull; }
            }
        });
        __syncthreads();
        }
        LMARK(1);
        // ---- positions: bitonic sort (descending) of the unique keys ((degree + 1) << 12) | (4095 - row) (k_rowsort's order) ----
        int N = 64; while (N < L) N <<= 1;
        // (the sorted key array without the sort: place_keys(); its two tables of L words each lie in tabS, which is written behind this phase)
        if (eqMax <= 0 || !place_keys(L, N, eqMax, [&](int k) { return degS[k]; }, keyS, reinterpret_cast<uint32_t*>(tabS), reinterpret_cast<uint32_t*>(tabS) + (STREAM_MAXL + 64), tmpS, wsum)) {
        for (int t = tid; t < N; t += LISTS_NT) keyS[t] = (t < L) ? (((degS[t] + 1u) << 12) | (uint32_t)(4095 - t)) : 0u;
        __syncthreads();
        for (int k2 = 2; k2 <= N; k2 <<= 1)
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (int t = tid; t < (N >> 1); t += LISTS_NT) {
                    const int i1 = ((t & ~(j2 - 1)) << 1) | (t & (j2 - 1)), i2 = i1 | j2;
                    const uint32_t a = keyS[i1], c2 = keyS[i2];
                    const bool desc = (i1 & k2) == 0;
                    if ((a < c2) == desc) { keyS[i1] = c2; keyS[i2] = a; }
                }
                __syncthreads();                                // (stages inside a wave's own 128 keys without the barrier: measured, no faster)
            }
        }
        LMARK(2);
        // ---- list room per position (whole quads), the problem's base from the batch's bump pointer ----
        {
            const int PER = (L + LISTS_NT - 1) / LISTS_NT;      // consecutive positions per thread (<= 3)
            uint32_t sum = 0;
            for (int t = 0; t < PER; ++t) {
                const int q = tid * PER + t;
                if (q < L) sum += (min((keyS[q] >> 12) - 1u, (uint32_t)(L - 1 - q)) + 3u) & ~3u;
            }
            const uint32_t inc = wave_incl_scan(sum);
            if (lane == WAVE - 1) wsum[w] = inc;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
            for (int t = 0; t < NW; ++t) { if (t < w) wbase += wsum[t]; total += wsum[t]; }
            if (tid == 0) {
                const unsigned long long base = atomicAdd(&tot->listTop, (unsigned long long)total);
                st[b].listOff = (int64_t)base;
                const bool fits = (long long)(base + total) <= capList;
                if (!fits) { st[b].kind = 2; atomicAdd(&tot->overflow, 1); }
                okS = fits ? 1 : 0;
            }
            uint32_t run = wbase + inc - sum;
            for (int t = 0; t < PER; ++t) {
                const int q = tid * PER + t;
                if (q < L) {
                    const uint32_t k = 4095u - (keyS[q] & 4095u);
                    perm[lo + q] = k; rowPos[lo + k] = (uint32_t)q;
                    tabS[k] = ((unsigned long long)run << 16) | (unsigned long long)q; listOff[lo + q] = run; degS[k] = 0u;     // (degS: now the row's list cursor)
                    run += (min((keyS[q] >> 12) - 1u, (uint32_t)(L - 1 - q)) + 3u) & ~3u;
                }
            }
        }
        __syncthreads();
        if (!okS) continue;                                     // lists beyond the pool: skipped like any workspace overflow (the history now knows the need)
        uint16_t* const lists = listPool + st[b].listOff;
        LMARK(3);
        // ---- every stored pair to its endpoint of smaller position ----
        sweep([&](int k, unsigned long long m, int c) {
            if (!m) return;
            const unsigned long long tk = tabS[k];
            const uint32_t pk = (uint32_t)tk & 0xffffu, ok = (uint32_t)(tk >> 16);
            while (m) {
                const uint32_t q = (uint32_t)((c << 6) + __builtin_ctzll(m));
                m &= m - 1ull;
                const unsigned long long tq = tabS[q];
                const bool mine = pk < ((uint32_t)tq & 0xffffu);                       // the pair belongs to its endpoint of smaller position
                const uint32_t slot = atomicAdd(&degS[mine ? (uint32_t)k : q], 1u);
                lists[(mine ? ok : (uint32_t)(tq >> 16)) + slot] = (uint16_t)(mine ? q : (uint32_t)k);
            }
        });
        __syncthreads();
        LMARK(4);
        // ---- padding, upper degrees and the position-ordered pools ----
        for (int p = tid; p < L; p += LISTS_NT) {
            const uint32_t k = 4095u - (keyS[p] & 4095u);
            const uint32_t cnt = degS[k];
            rowCnt[lo + p] = cnt;
            uint16_t* lst = lists + (uint32_t)(tabS[k] >> 16);
            for (uint32_t e = cnt; e < ((cnt + 3u) & ~3u); ++e) lst[e] = (uint16_t)0xffffu;
            dst.lp[lo + p] = src.lp[lo + k]; dst.ld[lo + p] = src.ld[lo + k];
        }
#ifdef ROMAN_LISTS_TIMING
        LMARK(5);
        if (tid == 0 && (b & 63) == 0) printf("[k_lists] b=%d L=%d cycles: degrees %llu sort %llu offsets %llu scatter %llu tail %llu\n", b, L,
                                              tl_[1] - tl_[0], tl_[2] - tl_[1], tl_[3] - tl_[2], tl_[4] - tl_[3], tl_[5] - tl_[4]);
#endif
#undef LMARK
    }
}

// k_slicegeom (kind 0): slice widths (longest upper row of the slice, rounded up to whole quads), slice bases and the
// problem's slot total.  One wave per problem, lane = slice.
__global__ void __launch_bounds__(64) k_slicegeom(const ProbDesc* __restrict__ probs, ProbState* __restrict__ st,
                                                  const uint32_t* __restrict__ rowCnt,
                                                  uint32_t* __restrict__ sliceWidth, uint32_t* __restrict__ sliceBase)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (st[b].kind != 0) return;
    const int L = st[b].L;
    const int64_t lo = probs[b].liveOff;
    const int nsl = (L + 63) >> 6;                                // <= STREAM_MAXL / 64 <= 64
    uint32_t width = 0;
    if (lane < nsl) {
        const int hi = min(L, (lane << 6) + 64);
        for (int p = lane << 6; p < hi; ++p) width = max(width, rowCnt[lo + p]);
        width = (width + 3u) & ~3u;
        sliceWidth[lo + lane] = width;
    }
    const uint32_t v = width * 64u;
    const uint32_t ex = wave_excl_scan(v, lane);
    if (lane < nsl) sliceBase[lo + lane] = ex;
    if (lane == WAVE - 1) st[b].nnzCap = ex + v;
}

// k_probscan: prefix over the problems of the per-problem slot totals and of the slice-group counts (k_fill_slice work items:
// min(NG, slices) groups of consecutive slices per stream-layout problem).  A problem whose matrix segment would end beyond
// `capNnz` slots becomes kind 2 (skipped; nothing has been written for it yet).
__global__ void __launch_bounds__(1024) k_probscan(int B, int NG /* fill groups per problem */, long long capNnz, ProbState* __restrict__ st, BatchTotals* __restrict__ tot)
{
    // one workgroup, PER consecutive problems per thread (k_rowbase)
    __shared__ long long shl[17];
    __shared__ int shi[17];
    __shared__ int red[1];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int PER = (B + nt - 1) / nt;
    const int b0 = tid * PER, b1 = min(B, b0 + PER);
    constexpr int PC = 8;                                        // (k_rowbase: the thread's problems are read once)
    int cK[PC], cL[PC]; uint32_t cC[PC];
#pragma unroll
    for (int j = 0; j < PC; ++j) { const int b = b0 + j; const bool in = j < PER && b < b1; cK[j] = in ? st[b].kind : 2; cL[j] = in ? st[b].L : 0; cC[j] = in ? st[b].nnzCap : 0u; }
    auto get = [&](int b, int& kind, int& L, long long& cap) {
        const int j = b - b0;
        if (j < PC) {
#pragma unroll
            for (int t = 0; t < PC; ++t) if (t == j) { kind = cK[t]; L = cL[t]; cap = (long long)cC[t]; }
        } else { kind = st[b].kind; L = st[b].L; cap = (long long)st[b].nnzCap; }
        if (kind >= 2) cap = 0;
    };
    long long sC = 0;
    for (int b = b0; b < b1; ++b) { int kind, L; long long cap; get(b, kind, L, cap); sC += cap; }
    long long totC; int totG;
    const long long baseC = block_excl_scan(sC, shl, totC);
    int sG = 0;
    {
        long long acc = baseC;
        for (int b = b0; b < b1; ++b) {
            int kind, L; long long cap; get(b, kind, L, cap);
            acc += cap;
            if (kind == 0 && acc <= capNnz) sG += min(NG, (L + 63) >> 6);
        }
    }
    const int baseG = block_excl_scan(sG, shi, totG);
    if (tid == 0) red[0] = 0;
    __syncthreads();
    int nover = 0;
    {
        long long acc = baseC; int gacc = baseG;
        for (int b = b0; b < b1; ++b) {
            int kind, L; long long cap; get(b, kind, L, cap);
            const bool fits = acc + cap <= capNnz;
            const int ng = (kind == 0 && fits) ? min(NG, (L + 63) >> 6) : 0;
            st[b].nnzOff = fits ? acc : 0; st[b].sgBase = gacc;
            if (!fits && kind < 2) { st[b].kind = 2; ++nover; }
            acc += cap; gacc += ng;
        }
    }
    for (int off = 32; off > 0; off >>= 1) nover += __shfl_xor(nover, off);
    if ((tid & 63) == 0 && nover) atomicAdd(&red[0], nover);
    __syncthreads();
    if (tid == 0) { tot->nnzTotal = totC < capNnz ? totC : capNnz; tot->sliceGroups = totG; tot->needNnz = totC; tot->overflow += red[0]; }
}

// ---------------------------------------------------------------------------------------------
// k_fill: candidates -> values, as ONE flat stream over the mask words of a work item (RPB rows x
// ceil(L/64) words, contiguous in HBM).  Each lane takes a mask word together with the candidate
// count in front of it; in every "bit step" each lane that still has set bits emits one candidate
// (row, column, entry index) into the wave's LDS ring.  Whenever 64 candidates are queued the wave
// evaluates them densely, whatever rows they belong to: x (recomputed bit-identically from the
// tables), sqrt/exp/cbrt, fusion with the two single scores, the affinityeps filter, and the store
// into the row's SELL slot column.  The problem's per-association data (objects, z, single score,
// SELL slot base) is staged in LDS once per work item.  Finally every row's slot column is padded
// to the slice width with inert entries.
// ---------------------------------------------------------------------------------------------
constexpr int FILL_E = 1;            // candidates a lane evaluates at a time (2: measured without gain, 212 B of scratch)
constexpr int FILL_Q = 256;          // ring capacity per wave (>= 64 FILL_E queued + 64 emitted per bit step)

// The rows of a k_fill work item (at most FILL_RPB consecutive positions) in LDS: live row, objects, single score, z, slice base + lane slot
constexpr int FILL_RPB = 128;
struct FillRows { int32_t* k; int32_t* i; int32_t* j; uint32_t* base; double* s; double* za; double* zb; };
constexpr int FILL_ROWBYTES = 4 * 4 + 3 * 8;    // per row

// CM: where an entry's per-association data comes from — 1: the problem's whole column tile in LDS (rows and columns); 2: a WINDOW of
// columns [wq0, wq0 + 64 Ww) in LDS, the item's rows from their own small tile (FillRows)
template <bool GRAV, typename IdxT, int CM, bool QUAD>
__device__ __forceinline__ uint32_t fill_item(const DevParams& D, const ProbDesc& pd, int L, int row0, int nrows, int wq0, int Ww,
                                              int w, int wpb, int lane,
                                              const int32_t* cI, const int32_t* cJ, const double* cS,
                                              const double* cZa, const double* cZb, const uint32_t* cBase, const uint32_t* cPos,
                                              const double* __restrict__ TA, const double* __restrict__ TB,
                                              const unsigned long long* __restrict__ mbase, const uint32_t* __restrict__ pbase,
                                              const FillRows& R /* the item's rows (R.k: position row0 + r -> live row: an item is a block of consecutive POSITIONS) */,
                                              uint32_t* qK, uint32_t* qQ, uint32_t* qE,
                                              IdxT* __restrict__ cols, double* __restrict__ vals)
{
    constexpr bool LDSCOL = CM == 1;
    const int W = (L + 63) >> 6;
    const int64_t nwords = (int64_t)nrows * Ww;
    const int w0 = wq0 >> 6;
    // The item's rows are the rows at positions row0 .. row0 + nrows - 1: what it writes is a few consecutive slices of the matrix (a
    // 128-byte line of the quad layout holds 16 bytes of each of eight NEIGHBOURING positions — with items of consecutive live rows its
    // eight parts arrived from eight workgroups at eight different times: eight partial writes per line)
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t head = 0, queued = 0, upper = 0;

    // up to FILL_E x 64 queued candidates at a time, FILL_E per lane: their record reads, then their table reads, are in flight together
    // (one candidate per lane was a chain of two dependent L2 round trips per 64 entries and wave: the kernel was bound by that latency,
    //  not by its gathers' number — packing seven arrays into one record changed nothing, 6.0 ms per 64 x L = 10 000 either way)
    auto evaluate = [&](uint32_t take) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bool on[FILL_E]; int rl[FILL_E], k[FILL_E], q[FILL_E]; uint32_t e[FILL_E];
        int i[FILL_E], j[FILL_E], iq[FILL_E], jq[FILL_E]; double sk[FILL_E], sq[FILL_E], dz[FILL_E];
        uint32_t base[FILL_E], pk[FILL_E], pq[FILL_E];       // slice base + lane slot (base is a multiple of 64); positions of row and column
#pragma unroll
        for (int h = 0; h < FILL_E; ++h) {
            const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)h;
            on[h] = idx < take;
            const uint32_t s = (head + (on[h] ? idx : 0u)) & (FILL_Q - 1);    // (an idle lane repeats the first candidate's reads; it stores nothing)
            rl[h] = (int)qK[s]; k[h] = R.k[rl[h]]; q[h] = (int)qQ[s]; e[h] = qE[s];
        }
#pragma unroll
        for (int h = 0; h < FILL_E; ++h) {
            if (LDSCOL) {
                i[h] = cI[k[h]]; j[h] = cJ[k[h]]; iq[h] = cI[q[h]]; jq[h] = cJ[q[h]]; sk[h] = cS[k[h]]; sq[h] = cS[q[h]];
                dz[h] = GRAV ? (cZa[k[h]] - cZa[q[h]]) - (cZb[k[h]] - cZb[q[h]]) : 0.0;
                base[h] = cBase[k[h]]; pk[h] = cPos[k[h]]; pq[h] = cPos[q[h]];
            } else {
                const int r_ = rl[h], ql = q[h] - wq0;
                i[h] = R.i[r_]; j[h] = R.j[r_]; iq[h] = cI[ql]; jq[h] = cJ[ql]; sk[h] = R.s[r_]; sq[h] = cS[ql];
                dz[h] = GRAV ? (R.za[r_] - cZa[ql]) - (R.zb[r_] - cZb[ql]) : 0.0;
                base[h] = R.base[r_]; pk[h] = (uint32_t)(row0 + r_); pq[h] = cPos[ql];
            }
        }
        double a[FILL_E], bb[FILL_E];
#pragma unroll
        for (int h = 0; h < FILL_E; ++h) { a[h] = TA[(int64_t)i[h] * pd.n1 + iq[h]]; bb[h] = TB[(int64_t)j[h] * pd.n2 + jq[h]]; }
#pragma unroll
        for (int h = 0; h < FILL_E; ++h) {
            double c;
            if (GRAV && D.gmode != 3) {
                const double ch = fabs(a[h] - bb[h]);
                const double hm = a[h] > bb[h] ? a[h] : bb[h];
                double cv = fabs(dz[h]) - D.sin_unc * hm;
                if (cv < 0.0) cv = 0.0;
                c = sqrt(ch * ch + cv * cv);
            } else {
                c = fabs(a[h] - bb[h]);                         // no gravity prior, or the z-gate reading (full lengths)
            }
            const double sa = fx_exp(((-0.5 * c) * c) / D.sig2);
            const double v = fuse_pair(D, sa, sk[h], sq[h]);
            const bool keep = v > D.p.affinityeps;
            // an entry at or below affinityeps belongs neither to M nor to C: inert slot
            // column labels are POSITIONS (sorted row order): the solvers keep their vectors in that order
            if (on[h]) {
                cols[col_pos<QUAD>(base[h] & ~63u, base[h] & 63u, e[h])] = keep ? (IdxT)pq[h] : fb_inert<IdxT>(pk[h]);
                vals[val_pos<QUAD>(base[h] & ~63u, base[h] & 63u, e[h])] = keep ? v : 0.0;
                upper += (keep && q[h] > k[h]) ? 1u : 0u;
            }
        }
        head = (head + take) & (FILL_Q - 1); queued -= take;
    };

    const int64_t nblk = (nwords + 63) >> 6;
    unsigned long long m_next = 0ull; uint32_t p_next = 0u, k_next = 0u;
    auto fetch = [&](int64_t x_) {                              // word x_ of the item's flat stream: live row, mask word, entries in front of it
        const int r_ = (int)x_ / Ww;                            // (RPB * W < 2^31)
        const int64_t a_ = (int64_t)R.k[r_] * W + (w0 + ((int)x_ - r_ * Ww));
        k_next = (uint32_t)r_; m_next = mbase[a_]; p_next = pbase[a_];
    };
    {   // prefetch the wave's first block
        const int64_t x = (int64_t)w * 64 + lane;
        if (w < nblk && x < nwords) fetch(x);
    }
    for (int64_t blk = w; blk < nblk; blk += wpb) {
        unsigned long long m = m_next; uint32_t e = p_next; const uint32_t kl = k_next;     // (kl: the word's row of the item)
        const int64_t x = blk * 64 + lane;
        {   // prefetch the next block while this one is expanded / evaluated
            const int64_t xn = x + (int64_t)wpb * 64;
            m_next = 0ull; p_next = 0u;
            if (blk + wpb < nblk && xn < nwords) fetch(xn);
        }
        const uint32_t qb = (uint32_t)(w0 + ((int)x - ((int)x / Ww) * Ww)) << 6;  // first column of this word
        for (;;) {                                              // bit steps
            const bool has = m != 0ull;
            const unsigned long long act = __ballot(has);
            if (act == 0ull) break;
            if (has) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t s = (head + queued + (uint32_t)__popcll(act & lt)) & (FILL_Q - 1);
                qK[s] = kl; qQ[s] = qb + (uint32_t)bit; qE[s] = e;
                ++e;
            }
            queued += (uint32_t)__popcll(act);
            while (queued >= 64u * FILL_E) evaluate(64u * FILL_E);
        }
    }
    while (queued > 0u) evaluate(min(queued, 64u * (uint32_t)FILL_E));
    return upper;
}

template <bool GRAV, typename IdxT, bool QUAD>
__global__ void __launch_bounds__(1024) k_fill(DevParams D, const ProbDesc* __restrict__ probs,
                                               ProbState* __restrict__ st,
                                               const BatchTotals* __restrict__ tot,
                                               const ItemDesc* __restrict__ items,
                                               const double* __restrict__ tabPool,
                                               const int32_t* __restrict__ li, const int32_t* __restrict__ lj,
                                               const double* __restrict__ ls,
                                               const double* __restrict__ lza, const double* __restrict__ lzb,
                                               const uint32_t* __restrict__ rowCnt,
                                               const unsigned long long* __restrict__ maskPool,
                                               const uint32_t* __restrict__ prefPool,
                                               const uint32_t* __restrict__ rowPos,
                                               const uint32_t* __restrict__ sliceWidth,
                                               const uint32_t* __restrict__ sliceBase,
                                               IdxT* __restrict__ cols, double* __restrict__ vals, int TC, int RPB /* <= FILL_RPB */,
                                               const uint32_t* __restrict__ permPool /* position -> live row */)
{
    // LDS: cS[TC] [GRAV: cZa[TC] cZb[TC]] cI[TC] cJ[TC] cBase[TC] cPos[TC] | per-wave rings qK qQ qE
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* cS = reinterpret_cast<double*>(smem);
    double* cZa = cS + TC;
    double* cZb = cZa + (GRAV ? TC : 0);
    int32_t* cI = reinterpret_cast<int32_t*>(cZb + (GRAV ? TC : 0));
    int32_t* cJ = cI + TC;
    uint32_t* cBase = reinterpret_cast<uint32_t*>(cJ + TC);
    uint32_t* cPos = cBase + TC;
    uint32_t* rings = cPos + TC;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, w = tid >> 6, wpb = nt >> 6;
    FillRows R;                                                  // behind the rings: the item's rows
    R.s = reinterpret_cast<double*>(rings + (size_t)wpb * 3 * FILL_Q); R.za = R.s + FILL_RPB; R.zb = R.za + FILL_RPB;
    R.k = reinterpret_cast<int32_t*>(R.zb + FILL_RPB); R.i = R.k + FILL_RPB; R.j = R.i + FILL_RPB; R.base = reinterpret_cast<uint32_t*>(R.j + FILL_RPB);
    uint32_t* qK = rings + (size_t)w * 3 * FILL_Q;
    uint32_t* qQ = qK + FILL_Q;
    uint32_t* qE = qQ + FILL_Q;
    const int nItems = tot->items;
    // XCD-aware order (workgroup ids are dealt to the 8 XCDs round-robin): XCD x takes the CONTIGUOUS range [x Gx, (x + 1) Gx) of
    // work items — the items of a problem, which share its tables, columns and mask rows, run on one L2
    // A workgroup takes a UNIT of FILL_RPB / RPB consecutive items and fills the runs of items of one problem in it as ONE item of up to
    // FILL_RPB rows: what an item stages — a window of columns per ~2 800 live associations — does not depend on its rows (64 x L = 10 000:
    // 6.0 / 3.9 / 2.8 ms with items of 32 / 64 / 128 rows; the other kernels of the build keep the item size that suits them)
    const int gpu_ = max(1, FILL_RPB / RPB);
    const int nUnits = (nItems + gpu_ - 1) / gpu_;
    const int Gx_ = (nUnits + 7) >> 3;
    for (int sIdx_ = blockIdx.x; (sIdx_ >> 3) < Gx_; sIdx_ += gridDim.x) {
        const int u_ = (sIdx_ & 7) * Gx_ + (sIdx_ >> 3);
        if (u_ >= nUnits) continue;
        const int tEnd_ = min(nItems, (u_ + 1) * gpu_);
        for (int t = u_ * gpu_; t < tEnd_; ) {
        ItemDesc it = items[t];
        const int b = it.b;
        int te_ = t + 1;
        while (te_ < tEnd_ && items[te_].b == b) ++te_;         // (the items of a problem follow each other, in row order)
        const int nIt_ = te_ - t;
        t = te_;
        if (uni_i(st[b].kind) == 1) {           // stream-layout problems are filled by k_fill_slice, skipped ones not at all
        const ProbDesc pd = probs[b];
        const int L = st[b].L;
        const int64_t lo = pd.liveOff, mo = st[b].maskOff, no = st[b].nnzOff;
        const int nrows = min(nIt_ * RPB, L - it.row0);
        const double* TA = tabPool + pd.tabOff;
        const double* TB = TA + (int64_t)pd.n1 * pd.n1;
        const bool ldscol = L <= TC;
        __syncthreads();                        // every wave is done with the previous item's columns and rows
        for (int r = tid; r < nrows; r += nt) {
            const uint32_t pos = (uint32_t)(it.row0 + r);
            const int k = (int)permPool[lo + pos];
            R.k[r] = k; R.i[r] = li[lo + k]; R.j[r] = lj[lo + k]; R.s[r] = ls[lo + k];
            if (GRAV) { R.za[r] = lza[lo + k]; R.zb[r] = lzb[lo + k]; }
            R.base[r] = sliceBase[lo + (pos >> 6)] + (pos & 63u);
        }
        if (ldscol) {
            for (int q = tid; q < L; q += nt) {
                cI[q] = li[lo + q]; cJ[q] = lj[lo + q]; cS[q] = ls[lo + q];
                if (GRAV) { cZa[q] = lza[lo + q]; cZb[q] = lzb[lo + q]; }
                const uint32_t pos = rowPos[lo + q];
                cBase[q] = sliceBase[lo + (pos >> 6)] + (pos & 63u); cPos[q] = pos;
            }
        }
        __syncthreads();
        uint32_t upper = 0u;
        if (ldscol)
            upper = fill_item<GRAV, IdxT, 1, QUAD>(D, pd, L, it.row0, nrows, 0, (L + 63) >> 6, w, wpb, lane, cI, cJ, cS, cZa, cZb, cBase, cPos, TA, TB,
                                                maskPool + mo, prefPool + mo, R, qK, qQ, qE, cols + no, vals + no);
        else {
            // live sets beyond the tile: WINDOWS of TC columns (a multiple of 64), one after the other — the window's columns into LDS
            // (coalesced), the item's word stream restricted to the window's words; an entry then costs its two table reads and its two
            // stores instead of fifteen requests to the L2
            const int TCw = TC & ~63;
            for (int q0 = 0; q0 < L; q0 += TCw) {
                const int qn = min(TCw, L - q0);
                if (q0 > 0) __syncthreads();    // every wave is done with the previous window's columns
                for (int q = tid; q < qn; q += nt) {
                    const int64_t g_ = lo + q0 + q;
                    cI[q] = li[g_]; cJ[q] = lj[g_]; cS[q] = ls[g_];
                    if (GRAV) { cZa[q] = lza[g_]; cZb[q] = lzb[g_]; }
                    cPos[q] = rowPos[g_];
                }
                __syncthreads();
                upper += fill_item<GRAV, IdxT, 2, QUAD>(D, pd, L, it.row0, nrows, q0, (qn + 63) >> 6, w, wpb, lane, cI, cJ, cS, cZa, cZb, cBase, cPos, TA, TB,
                                                        maskPool + mo, prefPool + mo, R, qK, qQ, qE, cols + no, vals + no);
            }
        }
        // pad every row's slot column up to its slice width with inert entries (value 0, C-flag; the
        // column is the row's own position: a real, finite vector element whatever the solver gathers from)
        for (int r = w; r < nrows; r += wpb) {
            const uint32_t pos = (uint32_t)(it.row0 + r);        // (items are blocks of positions)
            const int k = (int)permPool[lo + pos];
            const uint32_t width = sliceWidth[lo + (pos >> 6)];
            const int64_t sb = no + sliceBase[lo + (pos >> 6)];
            for (uint32_t e = rowCnt[lo + k] + lane; e < width; e += WAVE) {
                cols[col_pos<QUAD>(sb, pos & 63u, e)] = fb_inert<IdxT>(pos);
                vals[val_pos<QUAD>(sb, pos & 63u, e)] = 0.0;
            }
        }
        if (QUAD && it.row0 + nrows == L && (L & 63)) {         // lane slots of the last slice that hold no row
            const uint32_t sl = (uint32_t)(L >> 6);
            const uint32_t width = sliceWidth[lo + sl];
            const int64_t sb = no + sliceBase[lo + sl];
            const uint32_t nfree = 64u - (uint32_t)(L & 63);
            for (uint32_t x = tid; x < nfree * width; x += nt) {
                const uint32_t slot = (uint32_t)(L & 63) + x / width, e = x % width;
                cols[col_pos<QUAD>(sb, slot, e)] = fb_inert<IdxT>(0u);                   // (column 0, flagged: gathers a real element, adds nothing)
                vals[val_pos<QUAD>(sb, slot, e)] = 0.0;
            }
        }
        for (int off = 32; off > 0; off >>= 1) upper += __shfl_xor(upper, off);
        if (lane == 0 && upper) atomicAdd(&st[b].nnzUpper, (unsigned long long)upper);
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_fill_list (stream layout): candidates -> values, written straight into the quad layout.  k_upper left every
// position row's kept candidates as a list of live column indices (ascending, padded to whole quads), so entry e of a
// row is simply element e of its list.  A wave takes one QUAD of one slice at a time — lane = lane slot = row, 4
// consecutive entries per lane: one 8-byte list read, four independent evaluation chains (two LDS column lookups, two
// table gathers, sqrt / exp / cbrt, fusion, affinityeps filter), then one 8-byte column store and two 16-byte value
// stores that are contiguous over the wave — the layout's own order, no staging image, no barriers inside a work
// item.  Entries beyond a row's count (slice padding) and filtered entries are written inert (value 0, the lane
// slot's dummy column with the C-flag).  Work items are groups of consecutive slices of one problem (NG groups per problem): the
// problem's column tile (objects, z, single score, position of every live association) is staged in LDS once per
// item, as are the rows of the group (live index, count, list offset); the quads of the group are dealt to the waves
// round-robin.  Items are ordered so that the groups of one problem run on one XCD (its tables stay in that L2).
// ---------------------------------------------------------------------------------------------
constexpr int FILLS_MAXSPI = STREAM_MAXL / 64;     // slices per work item (group): up to a whole problem

template <bool GRAV, bool FAST>
__device__ __forceinline__ double fill_value(const DevParams& D, double a, double bb, double dza, double dzb, double ss, double sk, double sq)
{
    double c;
    if (GRAV && D.gmode != 3) {
        const double ch = fabs(a - bb);
        const double hm = a > bb ? a : bb;
        double cv = fabs(dza - dzb) - D.sin_unc * hm;
        if (cv < 0.0) cv = 0.0;
        c = sqrt(ch * ch + cv * cv);
    } else {
        c = fabs(a - bb);                   // no gravity prior, or the z-gate reading (full lengths)
    }
    const double sa = fx_exp(((-0.5 * c) * c) / D.sig2);
    if (FAST) return (sa == 0.0) ? 0.0 : fx_cbrt(sa * ss);      // geometric mean, distance weight 1: fuse_pair()'s default branch
    return fuse_pair(D, sa, sk, sq);
}

template <bool GRAV, bool FAST, bool OBJ>
__global__ void __launch_bounds__(1024) k_fill_list(DevParams D, int B, const ProbDesc* __restrict__ probs,
                                                    ProbState* __restrict__ st, const BatchTotals* __restrict__ tot,
                                                    const double* __restrict__ tabPool, const double* __restrict__ feats, int NO /* OBJ: object capacity per map */,
                                                    const int32_t* __restrict__ li, const int32_t* __restrict__ lj,
                                                    const double* __restrict__ ls,
                                                    const double* __restrict__ lza, const double* __restrict__ lzb,
                                                    const uint16_t* __restrict__ listPool, const uint32_t* __restrict__ listOff,
                                                    const uint32_t* __restrict__ rowCnt,
                                                    const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rowPos,
                                                    const uint32_t* __restrict__ sliceWidth,
                                                    const uint32_t* __restrict__ sliceBase,
                                                    uint16_t* __restrict__ cols, double* __restrict__ vals, int TC, int NG, int SPI /* LDS capacity: slices per group */)
{
    // LDS: cS[TC] [OBJ: sO[2*NO] (x, y, z, -) | else GRAV: cZa[TC] cZb[TC]] cI[TC] cJ[TC] | gK gCnt gOff [SPI*64] | gQ[SPI+1] gSB[SPI+1] | cP[TC] (u16)
    // OBJ: the two distances of an entry are RECOMPUTED from the objects' coordinates (LDS) with k_tables' own operation
    // sequence — the same bits — instead of gathered from the tables: a wave's 64 lanes are 64 different rows, every table
    // gather touched 64 cache lines for 64 doubles (4 GB of L2 -> L1 line traffic per batch of 256 at config 3, the kernel's bound).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* cS = reinterpret_cast<double*>(smem);
    dbl2_t* sO = reinterpret_cast<dbl2_t*>(cS + TC);            // object o of map 1: sO[2o], sO[2o+1]; of map 2: sO[2(NO+o)], ...
    double* cZa = OBJ ? cS + TC + 8 * NO : cS + TC;
    double* cZb = cZa + ((GRAV && !OBJ) ? TC : 0);
    int32_t* cI = reinterpret_cast<int32_t*>(cZb + ((GRAV && !OBJ) ? TC : 0));
    int32_t* cJ = cI + TC;
    uint32_t* gK = reinterpret_cast<uint32_t*>(cJ + TC);         // rows of the group's slices: live index (~0: no row)
    uint32_t* gCnt = gK + SPI * 64;                     //   kept candidates
    uint32_t* gOff = gCnt + SPI * 64;                   //   list offset
    uint32_t* gQ = gOff + SPI * 64;                     // quads in front of every slice of the group (+ total)
    uint32_t* gSB = gQ + SPI + 1;                       // slice bases
    uint16_t* cP = reinterpret_cast<uint16_t*>(gSB + SPI + 1);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
    const int nGroups = tot->sliceGroups;
    // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin by id, so XCD x takes the CONTIGUOUS range
    // [x*Gx, (x+1)*Gx) of groups — the groups of one problem (which share its tables) run on one L2.
    const int Gx = (nGroups + 7) >> 3;
    for (int sIdx = blockIdx.x; (sIdx >> 3) < Gx; sIdx += gridDim.x) {
        const int t = (sIdx & 7) * Gx + (sIdx >> 3);
        if (t < nGroups) {
        int b = 0;                                              // last problem with sgBase <= t and at least one group
        {
            int lo_ = 0, hi_ = B - 1;
            while (lo_ < hi_) { const int mid = (lo_ + hi_ + 1) >> 1; if (st[mid].sgBase <= t) lo_ = mid; else hi_ = mid - 1; }
            b = lo_;
        }
        const ProbDesc pd = probs[b];
        const int L = st[b].L;
        const int W = (L + 63) >> 6;
        const int64_t lo = pd.liveOff, no = st[b].nnzOff;
        const uint16_t* lists = listPool + st[b].listOff;
        // every problem is cut into the same number of groups (min(NG, W)) of ceil(W / groups) slices: with a uniform
        // batch the static deal then hands every workgroup the same mix of heavy (first) and light (last) groups
        const int ngb = min(NG, W), spib = (W + ngb - 1) / max(ngb, 1);
        const int s_begin = (t - st[b].sgBase) * spib, s_end = min(W, s_begin + spib), ns = max(s_end - s_begin, 0);
        const double* TA = tabPool + pd.tabOff;
        const double* TB = TA + (int64_t)pd.n1 * pd.n1;
        __syncthreads();                        // every wave is done with the previous group's tile
        for (int q = tid; q < L; q += nt) {
            cI[q] = li[lo + q]; cJ[q] = lj[lo + q]; cS[q] = ls[lo + q]; cP[q] = (uint16_t)rowPos[lo + q];
            if (GRAV && !OBJ) { cZa[q] = lza[lo + q]; cZb[q] = lzb[lo + q]; }
        }
        if (OBJ) {
            const int pdim = D.p.point_dim;
            for (int o = tid; o < pd.n1 + pd.n2; o += nt) {
                const double* f = feats + (o < pd.n1 ? pd.off1 + o : pd.off2 + (o - pd.n1)) * D.F;
                const int slot = o < pd.n1 ? o : NO + (o - pd.n1);
                sO[2 * slot] = dbl2_t{f[0], pdim > 1 ? f[1] : 0.0};
                sO[2 * slot + 1] = dbl2_t{pdim > 2 ? f[2] : 0.0, 0.0};
            }
        }
        const bool horiz = D.gmode == 1 || D.gmode == 2;        // the tables hold horizontal distances (k_tables)
        for (int x = tid; x < ns * 64; x += nt) {
            const int p = s_begin * 64 + x;
            const bool row = p < L;
            gK[x] = row ? perm[lo + p] : 0xffffffffu; gCnt[x] = row ? rowCnt[lo + p] : 0u; gOff[x] = row ? listOff[lo + p] : 0u;
        }
        if (w == 0) {                           // quads in front of every slice of the group
            const uint32_t wq = (lane < ns) ? (sliceWidth[lo + s_begin + lane] >> 2) : 0u;
            const uint32_t ex = wave_excl_scan(wq, lane);
            if (lane < ns) { gQ[lane] = ex; gSB[lane] = sliceBase[lo + s_begin + lane]; }
            if (lane == ns) gQ[ns] = ex;        // ns <= STREAM_MAXL / 64 < 64
        }
        __syncthreads();
        const uint32_t Q = gQ[ns];
        uint32_t upper = 0;
        int si = 0;
        for (uint32_t u = (uint32_t)w; u < Q; u += (uint32_t)nw) {
            while (u >= gQ[si + 1]) ++si;                       // wave-uniform
            const uint32_t g = u - gQ[si];
            const int64_t sb = no + gSB[si];                    // first element of the slice (multiple of 256)
            const int x = si * 64 + lane;
            const uint32_t kraw = gK[x], cnt = gCnt[x];
            // Which list quad goes into slot quad g: the row's quads ROTATED by a row-dependent offset.  The order of a row's
            // entries is free (the solver's sums are exact), and the 64 rows of a slice — lanes of one wave — hold much the same
            // columns in the same (live-index) order: unrotated, the lanes of a solver step push onto the same few columns at
            // once (LDS atomics onto one address serialise: SQ_LDS_BANK_CONFLICT 0.55 of the LDS-active cycles of k_solve_up).
            // Measured (round 4, two boxes, alternating): solver launch in flight 1.41-1.45 -> 1.36-1.37 ms, p50 0.665-0.674 -> 0.640 ms,
            // throughput +0.5-1.5 %, isolated solver launch 1.09-1.11 -> 1.02-1.03 ms.  Starting lane l at l/64 of its row instead, with or
            // without a further rotation inside the quad: 1.04-1.06 ms, p50 0.64-0.65 (same boxes, alternating): the plain one stays.
            const uint32_t nq = (cnt + 3u) >> 2;
            uint32_t e0 = g << 2;
            if (!(D.solve_flags & 2)) { if (g < nq) e0 = ((g + (uint32_t)x * (uint32_t)(D.solve_flags >> 8)) % nq) << 2; }   // (ROMAN_FILL_ROTATE=0 keeps the list order)
            uint2 qq = make_uint2(0u, 0u);
            if (e0 < cnt) qq = *reinterpret_cast<const uint2*>(lists + gOff[x] + e0);
            const int k = (kraw == 0xffffffffu) ? 0 : (int)kraw;
            const double sk = cS[k];
            const double* TAr = TA + (int64_t)cI[k] * pd.n1;
            const double* TBr = TB + (int64_t)cJ[k] * pd.n2;
            const double zak = (GRAV && !OBJ) ? cZa[k] : 0.0, zbk = (GRAV && !OBJ) ? cZb[k] : 0.0;
            dbl2_t oa0 = {0.0, 0.0}, oa1 = {0.0, 0.0}, ob0 = {0.0, 0.0}, ob1 = {0.0, 0.0};
            if (OBJ) { oa0 = sO[2 * cI[k]]; oa1 = sO[2 * cI[k] + 1]; ob0 = sO[2 * (NO + cJ[k])]; ob1 = sO[2 * (NO + cJ[k]) + 1]; }
            const uint32_t inert = ((uint32_t)L + (uint32_t)lane) | 0x8000u;     // the lane slot's own dummy column
            uint32_t cw[4]; double vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool have = e0 + (uint32_t)j < cnt;
                const uint32_t qr = ((j < 2 ? qq.x : qq.y) >> (16 * (j & 1))) & 0xffffu;
                const int q = have ? (int)qr : k;               // (a padding entry evaluates the row against itself: discarded)
                double a, bb, dza, dzb;
                if (OBJ) {                                      // k_tables' sequence: dx*dx + dy*dy (+ dz*dz), one correctly rounded sqrt
                    const dbl2_t pa0 = sO[2 * cI[q]], pa1 = sO[2 * cI[q] + 1], pb0 = sO[2 * (NO + cJ[q])], pb1 = sO[2 * (NO + cJ[q]) + 1];
                    const double dxa = oa0.x - pa0.x, dya = oa0.y - pa0.y, dxb = ob0.x - pb0.x, dyb = ob0.y - pb0.y;
                    dza = oa1.x - pa1.x; dzb = ob1.x - pb1.x;
                    const double h2a = dxa * dxa + dya * dya, h2b = dxb * dxb + dyb * dyb;
                    a = horiz ? sqrt(h2a) : sqrt(h2a + dza * dza);
                    bb = horiz ? sqrt(h2b) : sqrt(h2b + dzb * dzb);
                } else {
                    a = TAr[cI[q]]; bb = TBr[cJ[q]];
                    dza = GRAV ? zak - cZa[q] : 0.0; dzb = GRAV ? zbk - cZb[q] : 0.0;
                }
                if (!have) { a = 0.0; bb = 0.0; }
                const double sq = cS[q];
                const double v = fill_value<GRAV, FAST>(D, a, bb, GRAV ? dza : 0.0, GRAV ? dzb : 0.0, sk * sq, sk, sq);
                const bool keep = have && v > D.p.affinityeps;  // otherwise the slot stays inert: neither in M nor in C
                cw[j] = keep ? (uint32_t)cP[q] : inert;
                vv[j] = keep ? v : 0.0;
                upper += keep ? 1u : 0u;                        // every stored entry is kept once: a strict-upper one
            }
            *reinterpret_cast<uint2*>(cols + sb + (int64_t)g * 256 + lane * 4) = make_uint2(cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16));
            *reinterpret_cast<double2*>(vals + sb + (int64_t)(2 * g) * 128 + lane * 2) = make_double2(vv[0], vv[1]);
            *reinterpret_cast<double2*>(vals + sb + (int64_t)(2 * g + 1) * 128 + lane * 2) = make_double2(vv[2], vv[3]);
        }
        for (int off = 32; off > 0; off >>= 1) upper += __shfl_xor(upper, off);
        if (lane == 0 && upper) atomicAdd(&st[b].nnzUpper, (unsigned long long)upper);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense problems (roman_set_matrix_data: the set_matrix_data / solve loop of [REF roman/align/object_registration.py:60-72]).
// The caller's M and C (n x n, row major) become the layouts the scored problems use, on the device: k_dense_mask writes
// the symmetric candidate bit matrix, the kernels of the scored path (k_rowprefix, k_rowsort, k_upper, k_slicegeom,
// k_probscan) turn it into positions, lists and slice geometry, k_dense_fill writes labels and values.  Like upstream
// only the STRICT UPPER triangles are read: entry (k, q) is M[min][max] / C[min][max]; it is stored when either is
// non-zero and carries the C flag when C is zero there.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dense_entry(const double* __restrict__ M, const double* __restrict__ C, int n, int k, int q, double& mv, double& cv)
{
    const int a = min(k, q), b = max(k, q);
    mv = M[(int64_t)a * n + b]; cv = C[(int64_t)a * n + b];
}

__global__ void __launch_bounds__(256) k_dense_mask(int n, const double* __restrict__ M, const double* __restrict__ C,
                                                    unsigned long long* __restrict__ mask /* n rows x ceil(n/64) words */,
                                                    int* __restrict__ flags /* [0]: some stored entry has C == 0 */)
{
    const int lane = threadIdx.x & 63;
    const int W = (n + 63) >> 6;
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    for (int k = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); k < n; k += nwaves) {
        bool cz = false;
        for (int w = 0; w < W; ++w) {
            const int q = (w << 6) + lane;
            bool cand = false;
            if (q < n && q != k) {
                double mv, cv;
                dense_entry(M, C, n, k, q, mv, cv);
                cand = (mv != 0.0) || (cv != 0.0);
                cz = cz || (cand && cv == 0.0);
            }
            const unsigned long long m = __ballot(cand);
            if (lane == 0) mask[(int64_t)k * W + w] = m;
        }
        if (__ballot(cz) != 0ull && lane == 0) atomicOr(flags, 1);
    }
}

// flags[1] |= 1 when a weight of the strict upper triangle lies outside [0, 1] or is not finite: such a matrix must not take
// the stream solver, whose fixed-point sums assume 0 <= v <= 1 (host: roman_set_matrix_data).
__global__ void __launch_bounds__(256) k_dense_range(int n, const double* __restrict__ M, int* __restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    bool bad = false;
    for (int k = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); k < n; k += nwaves)
        for (int q = k + 1 + lane; q < n; q += 64) {
            const double v = M[(int64_t)k * n + q];
            bad = bad || !(v >= 0.0 && v <= 1.0);               // (a NaN fails both comparisons)
        }
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(flags + 1, 1);
}

// One thread per slot of the problem's matrix segment (slot t of the COLUMN array; the value sits at val_pos of the same
// (slice, lane slot, entry)).  KIND 0: stream layout — entry e of position row p is element e of its candidate list;
// padding points at the dummy element n + lane slot.  KIND 1: symmetric sorted SELL-64 in quads — entry e of row k is the
// e-th set bit of its mask row (per-word prefix counts from k_rowprefix); padding = the row's own position, flagged.
template <int KIND>
__global__ void __launch_bounds__(256) k_dense_fill(int n, const double* __restrict__ M, const double* __restrict__ C,
                                                    ProbState* __restrict__ st,
                                                    const uint32_t* __restrict__ rowCnt, const uint32_t* __restrict__ rowPos, const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ sliceWidth, const uint32_t* __restrict__ sliceBase,
                                                    const uint16_t* __restrict__ listPool, const uint32_t* __restrict__ listOff,
                                                    const unsigned long long* __restrict__ mask, const uint32_t* __restrict__ pref,
                                                    uint16_t* __restrict__ cols16, uint32_t* __restrict__ cols32, double* __restrict__ vals)
{
    const uint32_t total = st[0].nnzCap;
    const int64_t no = st[0].nnzOff;
    const int nsl = (n + 63) >> 6, W = nsl;
    const uint16_t* lists = listPool + st[0].listOff;
    unsigned upper = 0;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int sl = 0;
        {   // last slice with sliceBase <= t (slices of width 0 share a base with their successor: skipped by the search)
            int lo_ = 0, hi_ = nsl - 1;
            while (lo_ < hi_) { const int mid = (lo_ + hi_ + 1) >> 1; if (sliceBase[mid] <= t) lo_ = mid; else hi_ = mid - 1; }
            sl = lo_;
        }
        const uint32_t sb = sliceBase[sl], off = t - sb;
        const uint32_t slot = (off & 255u) >> 2, e = ((off >> 8) << 2) | (off & 3u);
        const int pos = sl * 64 + (int)slot;
        const int k = pos < n ? (int)perm[pos] : -1;
        double val = 0.0;
        if (KIND == 0) {
            uint32_t label = ((uint32_t)n + slot) | 0x8000u;
            if (k >= 0 && e < rowCnt[pos]) {
                const int q = (int)lists[listOff[pos] + e];
                double mv, cv;
                dense_entry(M, C, n, k, q, mv, cv);
                label = rowPos[q] | (cv == 0.0 ? 0x8000u : 0u); val = mv; ++upper;
            }
            cols16[no + t] = (uint16_t)label;
        } else {
            uint32_t label = (uint32_t)(k >= 0 ? pos : 0) | 0x80000000u;
            if (k >= 0 && e < rowCnt[k]) {
                const unsigned long long* mrow = mask + (int64_t)k * W;
                const uint32_t* prow = pref + (int64_t)k * W;
                int lo_ = 0, hi_ = W - 1;                        // last word with prefix <= e
                while (lo_ < hi_) { const int mid = (lo_ + hi_ + 1) >> 1; if (prow[mid] <= e) lo_ = mid; else hi_ = mid - 1; }
                unsigned long long m = mrow[lo_];
                for (uint32_t i = prow[lo_]; i < e; ++i) m &= m - 1ull;
                const int q = (lo_ << 6) + __builtin_ctzll(m);
                double mv, cv;
                dense_entry(M, C, n, k, q, mv, cv);
                label = rowPos[q] | (cv == 0.0 ? 0x80000000u : 0u); val = mv;
                if (q > k) ++upper;
            }
            cols32[no + t] = label;
        }
        vals[no + val_pos<true>(sb, slot, e)] = val;
    }
    for (int off = 32; off > 0; off >>= 1) upper += __shfl_xor(upper, off);
    if ((threadIdx.x & 63) == 0 && upper) atomicAdd(&st[0].nnzUpper, (unsigned long long)upper);
}

// ---------------------------------------------------------------------------------------------
// solver
// ---------------------------------------------------------------------------------------------

struct SolveOut {           // device pointers of the batch outputs
    int32_t* assoc_out; int32_t* n_assoc_out; double* T_out; int32_t* status_out;
    roman_stats_t* stats_out; int32_t kmax;
    int32_t* nodesOrig;     // row pool: selected nodes as original association indices
    int32_t* nSel;          // per problem: number of selected nodes (untruncated)
    double*  uOut;          // row pool: final u over live associations
    unsigned long long* dbg;   // timing build only: 16 counters per problem
};

// Bounded launches of the stream solver (round 6).  A problem never leaves its workgroup, and a call with more problems than compute
// units hands them out from a queue: a problem of 400 passes that is claimed late holds its unit long after the others have nothing left
// to claim.  With a pass budget `cap` a problem that is still iterating after `cap` passes of this launch is SUSPENDED at the top of a
// pass: its iterate (u, the fused product of u, the vector about to be multiplied: 3 L doubles) and sixteen scalars go to a slot of
// `spill`, its number to `list`; a second launch of the same kernel (`resume`) picks the suspended problems up — all at once, one
// workgroup each — and runs them to the end.  The resumed iteration executes the same instructions on the same values (order-free sums,
// the same thread-to-element mapping, the same reduction trees): identical bits, identical pass counts.  No slot free: the problem
// simply keeps running.
struct SolveCont {
    double*  spill;        // slots * slotDoubles
    int32_t* list;         // problem number of every used slot
    int*     counters;     // [0]: slots handed out (may exceed `slots`), [1]: the resume launch's claim counter
    int32_t  cap;          // passes a problem may run in this launch before it is suspended (0: no limit)
    int32_t  slots, slotDoubles, maxL;   // a slot: 16 scalars, then u, Wu, x at distances of maxL doubles
    int32_t  resume;       // this launch takes the suspended problems
};

// The two copies between a slot and the workgroup's LDS (out of line: the solver's registers are full, and neither belongs in its loop).
// LDS image: x0 = u, x1 = the fused product of u, x2 = the vector about to be multiplied (L doubles each), sc = the sixteen scalars.
__device__ __noinline__ void cont_spill(const SolveCont* cont, int slot, int b, int L, const double* x0, const double* x1, const double* x2, const double* sc)
{
    double* sp = cont->spill + (size_t)slot * (size_t)cont->slotDoubles;
    const int Lm = cont->maxL;
    for (int p = threadIdx.x; p < L; p += blockDim.x) { sp[16 + p] = x0[p]; sp[16 + Lm + p] = x1[p]; sp[16 + 2 * Lm + p] = x2[p]; }
    if (threadIdx.x < 16) sp[threadIdx.x] = sc[threadIdx.x];
    if (threadIdx.x == 0) cont->list[slot] = b;
    __threadfence();                                            // the resume launch (same stream) reads it
}
__device__ __noinline__ void cont_load(const SolveCont* cont, int slot, int L, double* x0, double* x1, double* x2, double* sc)
{
    const double* sp = cont->spill + (size_t)slot * (size_t)cont->slotDoubles;
    const int Lm = cont->maxL;
    for (int p = threadIdx.x; p < L; p += blockDim.x) { x0[p] = sp[16 + p]; x1[p] = sp[16 + Lm + p]; x2[p] = sp[16 + 2 * Lm + p]; }
    if (threadIdx.x < 16) sc[threadIdx.x] = sp[threadIdx.x];
    __syncthreads();
}

// Sum of (a, b) over the block, identical in every thread; fixed reduction tree.  `red` holds two
// ping-pong scratch areas of 32 doubles (`par` flips on every call), so consecutive reductions need a
// single barrier each: a buffer is rewritten only after the barrier of the following reduction.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red, int& par, int tid, int nw)
{
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    double* rr = red + 32 * par;
    par ^= 1;
    if ((tid & 63) == 0) { rr[2 * (tid >> 6)] = a; rr[2 * (tid >> 6) + 1] = b; }
    __syncthreads();
    double ra = 0.0, rb = 0.0;
    for (int i = 0; i < nw; ++i) { ra += rr[2 * i]; rb += rr[2 * i + 1]; }
    a = ra; b = rb;
}

// (M_off u)_p and (C_off u)_p for every position p from the fallback layout (symmetric sorted SELL-64 in quads: the 4 column
// words of entries 4g..4g+3 of a lane are one 16-byte load, the values two 16-byte loads): a wave owns a slice (lane = row
// slot) and walks the quads, G quads in flight.  Padding entries — and the lane slots of the last slice that hold no
// row — are inert (value 0, C-flag, a valid column), so there are no predicates.
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
template <typename IdxT, int G>
__device__ __forceinline__ void spmv_sell(const double* u, int L, const uint32_t* __restrict__ perm,
                                          const uint32_t* __restrict__ sliceWidth, const uint32_t* __restrict__ sliceBase,
                                          const IdxT* __restrict__ cols, const double* __restrict__ vals,
                                          double* Mu, double* Cu, int tid, int nt)
{
    static_assert(sizeof(IdxT) == 4, "fallback layout: 32-bit column words");
    const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
    const int nsl = (L + 63) >> 6;
    for (int s = w; s < nsl; s += nw) {
        const int pos = (s << 6) + lane;
        const bool valid = pos < L;
        const uint32_t nq = sliceWidth[s] >> 2;
        const uint4_t* cp = reinterpret_cast<const uint4_t*>(cols + sliceBase[s]) + lane;
        const dbl2_t* vp = reinterpret_cast<const dbl2_t*>(vals + sliceBase[s]) + lane;
        double am = 0.0, ac = 0.0;
        for (uint32_t g0 = 0; g0 < nq; g0 += G) {
            uint4_t c_[G]; dbl2_t v0_[G], v1_[G];
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const uint32_t g = min(g0 + (uint32_t)t, nq - 1u);      // (clamped: a repeated quad is skipped below)
                c_[t] = cp[(size_t)g * 64]; v0_[t] = vp[(size_t)(2 * g) * 64]; v1_[t] = vp[(size_t)(2 * g + 1) * 64];
            }
#pragma unroll
            for (int t = 0; t < G; ++t) {
                if (g0 + (uint32_t)t < nq) {
                    const uint32_t c4[4] = {c_[t].x, c_[t].y, c_[t].z, c_[t].w};
                    const double v4[4] = {v0_[t].x, v0_[t].y, v1_[t].x, v1_[t].y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double uq = u[c4[e] & IdxTraits<IdxT>::MASK];
                        am = fma(v4[e], uq, am);
                        ac += (c4[e] & IdxTraits<IdxT>::CZ) ? 0.0 : uq;
                    }
                }
            }
        }
        if (valid) { Mu[pos] = am; Cu[pos] = ac; }
    }
}

// One-sided Jacobi (Hestenes) SVD of a dxd matrix (d = 2 or 3), then the proper rotation
// R = u1 v1' + u2 v2' + (u1 x u2)(v1 x v2)'   — equal to the reference's  U Vh  with the last
// row of Vh negated when det = -1 ([REF roman/align/object_registration.py:121-126]).
__device__ __forceinline__ double wave_sum63(double v);       // (defined with the stream solver's reductions below)
__device__ __forceinline__ double readlane63(double v);

// One-sided Jacobi step on the column pair (P, Q) of G (V accumulates the rotations).  Everything is indexed at compile time:
// G and V stay in registers (the round-1 form walked p, q, r with run-time indices — 76 scratch accesses of ~500 cycles each
// on the one thread that runs it: 47 k cycles per alignment, a twentieth of a config-3 solve and half of a demo-scale one).
template <int P, int Q>
__device__ __forceinline__ void jacobi_pair(double (&G)[9], double (&V)[9], bool& moved)
{
    double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) { al += G[r * 3 + P] * G[r * 3 + P]; be += G[r * 3 + Q] * G[r * 3 + Q]; ga += G[r * 3 + P] * G[r * 3 + Q]; }
    if (ga == 0.0) return;
    const double ab = al * be, g2 = ga * ga;                    // |ga| <= 1e-17 sqrt(al be), squared (no square root on the one thread's critical path)
    if (g2 <= 1e-34 * ab) return;
    if (!(g2 < 1e-30 * ab)) moved = true;                      // the sweep still found a pair of columns that is not orthogonal to working precision
    // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (be - al) / (2 ga), with numerator and denominator multiplied by |2 ga|:
    // one division and one square root instead of two and one; c = 1 / sqrt(1 + t^2) as a reciprocal square root
    const double a2 = be - al, b2 = 2.0 * ga;
    const double t = (((a2 >= 0.0) == (b2 >= 0.0)) ? fabs(b2) : -fabs(b2)) / (fabs(a2) + sqrt(a2 * a2 + b2 * b2));
    const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double gp = G[r * 3 + P], gq = G[r * 3 + Q];
        G[r * 3 + P] = c * gp - s * gq; G[r * 3 + Q] = s * gp + c * gq;
        const double vp = V[r * 3 + P], vq = V[r * 3 + Q];
        V[r * 3 + P] = c * vp - s * vq; V[r * 3 + Q] = s * vp + c * vq;
    }
}
// column c (run-time index) of a 3 x 3 matrix held in registers
__device__ __forceinline__ double col3(const double (&A)[9], int r3, int c) { return c == 0 ? A[r3] : (c == 1 ? A[r3 + 1] : A[r3 + 2]); }

// Rotation of the Kabsch / Umeyama fit from the d x d cross-covariance H (d = 2 or 3; rows and columns >= d of H are ignored):
// one-sided Jacobi SVD of H, the two leading singular pairs, the third by cross products (a proper rotation whatever the rank).
__device__ __forceinline__ void kabsch_rotation(const double (&H)[9], int d, double (&R)[9])
{
    double G[9], V[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            G[r * 3 + c] = (r < d && c < d) ? H[r * 3 + c] : 0.0;
            V[r * 3 + c] = (r == c) ? 1.0 : 0.0;
            R[r * 3 + c] = 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {                 // (d == 2: the pairs with column 2 find ga == 0 and return)
        bool moved = false;
        jacobi_pair<0, 1>(G, V, moved);
        jacobi_pair<0, 2>(G, V, moved);
        jacobi_pair<1, 2>(G, V, moved);
        if (!moved) break;
    }
    double sg[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { double s = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) s += G[r * 3 + c] * G[r * 3 + c];
        sg[c] = sqrt(s); }
    int i1 = 0;
#pragma unroll
    for (int c = 1; c < 3; ++c) if (c < d && sg[c] > (i1 == 0 ? sg[0] : (i1 == 1 ? sg[1] : sg[2]))) i1 = c;
    int i2 = -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (c < d && c != i1 && (i2 < 0 || sg[c] > (i2 == 0 ? sg[0] : (i2 == 1 ? sg[1] : sg[2])))) i2 = c;
    const double s1 = i1 == 0 ? sg[0] : (i1 == 1 ? sg[1] : sg[2]), s2 = i2 == 0 ? sg[0] : (i2 == 1 ? sg[1] : sg[2]);
    double u1[3] = {0, 0, 0}, u2[3] = {0, 0, 0}, v1[3] = {0, 0, 0}, v2[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < d) { v1[r] = col3(V, r * 3, i1); v2[r] = col3(V, r * 3, i2); }
    if (s1 > 0.0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) if (r < d) u1[r] = col3(G, r * 3, i1) / s1;
    } else { u1[0] = 1.0; }                                             // H == 0: any rotation
    if (d == 2) {
        // R = u1 v1' + perp(u1) perp(v1)'
        const double pu[2] = {-u1[1], u1[0]}, pv[2] = {-v1[1], v1[0]};
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) R[r * 3 + c] = u1[r] * v1[c] + pu[r] * pv[c];
        return;
    }
    if (s2 > 1e-300 && s2 > 1e-14 * s1) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] = col3(G, r * 3, i2) / s2;
    } else {                                                            // rank <= 1: complete u1 arbitrarily
        int m = 0;
        if (fabs(u1[1]) < fabs(u1[0])) m = 1;
        if (fabs(u1[2]) < fabs(m == 0 ? u1[0] : u1[1])) m = 2;
        const double dp = m == 0 ? u1[0] : (m == 1 ? u1[1] : u1[2]);
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { u2[r] = (r == m ? 1.0 : 0.0) - dp * u1[r]; nn += u2[r] * u2[r]; }
        nn = sqrt(nn);
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] /= nn;
    }
    const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c];
}

// T (row-major (d+1)x(d+1) in the leading entries of 16 doubles) from centred sums.
__device__ __forceinline__ void write_pose(double (&T)[16], int d, const double (&H)[9], const double (&m1)[3], const double (&m2)[3])
{
    double R[9];
    kabsch_rotation(H, d, R);
#pragma unroll
    for (int t = 0; t < 16; ++t) T[t] = 0.0;
    if (d == 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double tr = m1[r];
#pragma unroll
            for (int c = 0; c < 3; ++c) { T[r * 4 + c] = R[r * 3 + c]; tr -= R[r * 3 + c] * m2[c]; }
            T[r * 4 + 3] = tr;
        }
        T[15] = 1.0;
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double tr = m1[r];
#pragma unroll
            for (int c = 0; c < 2; ++c) { T[r * 3 + c] = R[r * 3 + c]; tr -= R[r * 3 + c] * m2[c]; }
            T[r * 3 + 2] = tr;
        }
        T[8] = 1.0;
    }
}

// Centred cross-covariance of the nsel selected correspondences and the pose, by ONE wave (all 64 lanes active): nsel is tens to a
// few hundred — lanes stride over them, the 6 + 9 sums are reduced with DPP (no workgroup barrier, no LDS round trip), lane 63's
// totals are broadcast, and every lane forms the same pose in registers (the caller lets one of them store it).
template <typename Fetch>
__device__ __forceinline__ void wave_pose(double (&T)[16], int dim, int nsel, int lane, Fetch&& fetch /* (t, a[3], b[3]): the t-th pair of points */)
{
    constexpr int KEEP = 2;                                     // pairs a lane keeps in registers for the second sweep (128 of them: beyond, the points are fetched again)
    double ka[KEEP][3], kq[KEEP][3];
    double m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
#pragma unroll
    for (int x = 0; x < KEEP; ++x) {
        const int t = lane + 64 * x;
        if (t < nsel) fetch(t, ka[x], kq[x]);
        else {
#pragma unroll
            for (int c = 0; c < 3; ++c) { ka[x][c] = 0.0; kq[x][c] = 0.0; }
        }
    }
#pragma unroll
    for (int x = 0; x < KEEP; ++x)
#pragma unroll
        for (int c = 0; c < 3; ++c) { m1[c] += ka[x][c]; m2[c] += kq[x][c]; }
    for (int t = lane + 64 * KEEP; t < nsel; t += 64) {
        double a[3], q[3];
        fetch(t, a, q);
#pragma unroll
        for (int c = 0; c < 3; ++c) { m1[c] += a[c]; m2[c] += q[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { m1[c] = readlane63(wave_sum63(m1[c])) / (double)nsel; m2[c] = readlane63(wave_sum63(m2[c])) / (double)nsel; }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](double (&a)[3], double (&q)[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] -= m1[c]; q[c] -= m2[c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[r * 3 + c] += a[r] * q[c];
    };
#pragma unroll
    for (int x = 0; x < KEEP; ++x) if (lane + 64 * x < nsel) add(ka[x], kq[x]);
    for (int t = lane + 64 * KEEP; t < nsel; t += 64) {
        double a[3], q[3];
        fetch(t, a, q);
        add(a, q);
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) H[e] = readlane63(wave_sum63(H[e]));
    write_pose(T, dim, H, m1, m2);
}

// Exact emulation of findIndicesOfkLargest (min-heap on (value,index); strict '<' replacement)
// over the FULL association index space (dead associations have u == 0).  Single thread; only
// runs when the k-th largest value is tied or fewer than omega entries are positive.
__device__ inline bool hp_less(double va, int ia, double vb, int ib) { return va < vb || (va == vb && ia < ib); }
__device__ __noinline__ void heap_select_serial(const double* u, const int32_t* lp /* ascending: live order */,
                                   const uint32_t* vecOfLive /* live index -> index into u, or nullptr = identity */,
                                   int L, int nA, int k, bool liveOnly /* removed associations are not candidates at all (pruned list) */,
                                   double* hv, int32_t* hi /* capacity k */, int32_t* outNodesOrig)
{
    int sz = 0, nl = 0;
    const int nIter = liveOnly ? L : nA;
    for (int it = 0; it < nIter; ++it) {
        int p = it; double x = 0.0;
        if (liveOnly) { p = lp[it]; x = u[vecOfLive ? (int)vecOfLive[it] : it]; }
        else if (nl < L && lp[nl] == p) { x = u[vecOfLive ? (int)vecOfLive[nl] : nl]; ++nl; }
        if (sz < k) {
            int c = sz++; hv[c] = x; hi[c] = p;
            while (c > 0) { const int par = (c - 1) >> 1; if (!hp_less(hv[c], hi[c], hv[par], hi[par])) break;
                const double tv = hv[c]; const int ti = hi[c]; hv[c] = hv[par]; hi[c] = hi[par]; hv[par] = tv; hi[par] = ti; c = par; }
        } else if (hv[0] < x) {
            hv[0] = x; hi[0] = p;
            int c = 0;
            for (;;) { int l = 2 * c + 1, r = l + 1, s = c;
                if (l < sz && hp_less(hv[l], hi[l], hv[s], hi[s])) s = l;
                if (r < sz && hp_less(hv[r], hi[r], hv[s], hi[s])) s = r;
                if (s == c) break;
                const double tv = hv[c]; const int ti = hi[c]; hv[c] = hv[s]; hi[c] = hi[s]; hv[s] = tv; hi[s] = ti; c = s; }
        }
    }
    const int kk = sz;
    for (int t = 0; t < kk; ++t) {
        outNodesOrig[kk - t - 1] = hi[0];
        hv[0] = hv[sz - 1]; hi[0] = hi[sz - 1]; --sz;
        int c = 0;
        for (;;) { int l = 2 * c + 1, r = l + 1, s = c;
            if (l < sz && hp_less(hv[l], hi[l], hv[s], hi[s])) s = l;
            if (r < sz && hp_less(hv[r], hi[r], hv[s], hi[s])) s = r;
            if (s == c) break;
            const double tv = hv[c]; const int ti = hi[c]; hv[c] = hv[s]; hi[c] = hi[s]; hv[s] = tv; hi[s] = ti; c = s; }
    }
}

// Tail shared by every solver variant: final u to the row pool, top-omega rounding (with the exact
// heap emulation on ties), selected associations, Umeyama pose, statistics.  `u` is indexed by "vector index":
// the live index (fallback solver) or the position (stream solver); lp[lo + v] is the association index of
// vector index v.  lpAsc is the ascending association list in live order and vecOfLive the live -> vector index
// map (nullptr: identity) — only the sequential heap emulation needs them.  pv / pidx / nodesLive are scratch
// arrays of capacity L.
__device__ __noinline__ void finish_one(const DevParams& D, int b, const ProbDesc& pd, const double* __restrict__ feats,
                                        const int32_t* __restrict__ assoc, const int32_t* __restrict__ lp,
                                        const int32_t* __restrict__ lpAsc, const uint32_t* __restrict__ vecOfLive,
                                        const uint32_t* __restrict__ outIdx /* vector index -> slot of uOut (nullptr: identity) */, const SolveOut& O,
                                        const double* u, double* pv, int32_t* pidx, int32_t* nodesLive,
                                        int L, int rb, int64_t lo, double F, int status, roman_stats_t S,
                                        double* red, int* sint)
{
    const roman_params_t& P = D.p;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int dim = P.point_dim;
    int nsel = 0;
    __syncthreads();
#ifdef ROMAN_SMALL_TIMING
    unsigned long long tf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; tf_[0] = __builtin_readcyclecounter();
#define FMARK(i_) tf_[i_] = __builtin_readcyclecounter()
#else
#define FMARK(i_) do { } while (0)
#endif
    if (L > 0) {
        // final u to the row pool (stepwise API, tests)
        for (int p = tid; p < L; p += nt) O.uOut[rb + (outIdx ? (int)outIdx[lo + p] : p)] = u[p];

        // ---- top-omega rounding --------------------------------------------------------------
        const double om = round(F);
        int omega = (om >= 2147483647.0) ? 2147483647 : (om < 1.0 ? 0 : (int)om);
        if (omega > (D.pruned ? L : pd.nA)) omega = D.pruned ? L : pd.nA;
        int32_t* nodesOrig = O.nodesOrig + rb;        // capacity L
        if (tid == 0) { sint[0] = 0; sint[1] = 0; }
        __syncthreads();
        if (omega > 0) {
            // compact the positive entries (order irrelevant: ranks below are order-free) as (value, association index)
            // (one LDS atomic per wave and sweep — the wave's count — instead of one per positive element on the same word)
            for (int p0 = 0; p0 < L; p0 += nt) {
                const int p = p0 + tid;
                const double up = p < L ? u[p] : 0.0;
                const unsigned long long pm = __ballot(up > 0.0);
                if (pm) {
                    int base_ = 0;
                    if ((tid & 63) == 0) base_ = atomicAdd(&sint[0], (int)__popcll(pm));
                    base_ = __builtin_amdgcn_readfirstlane(base_);
                    if (up > 0.0) {
                        const int pos = base_ + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                        pv[pos] = up; pidx[pos] = lp[lo + p];
                    }
                }
            }
            __syncthreads();
            FMARK(1);
            const int Pn = sint[0];
            bool fallback = (Pn < omega) || (omega > L);
            if (!fallback) {
                // rank of e = number of entries greater in (value, association index) order (the inner loop reads LDS only)
                // `parts` adjacent lanes share an entry (a power of two, as many as the threads allow: a hundred entries leave four fifths
                // of a 512-thread workgroup idle otherwise), each counts a part of the list, a butterfly inside the group adds the counts
                int parts = 1;
                while (parts < 8 && 2 * parts * Pn <= nt) parts <<= 1;
                const int chunk = (Pn + parts - 1) / parts;
                for (int base = 0; base < Pn * parts; base += nt) {
                    const int idx = base + tid, e = idx / parts, part = idx & (parts - 1);
                    const bool act = e < Pn;
                    const double ve = pv[act ? e : 0]; const int ae = pidx[act ? e : 0];
                    const int flo = part * chunk, fhi = min(Pn, flo + chunk);
                    int rank = 0;
                    for (int f0 = flo; f0 < fhi; f0 += 8) {         // eight LDS reads in flight (one per iteration: the loop ran at the LDS latency)
                        double vf[8]; int af[8];
#pragma unroll
                        for (int x = 0; x < 8; ++x) { const int f2 = min(f0 + x, Pn - 1); vf[x] = pv[f2]; af[x] = pidx[f2]; }
#pragma unroll
                        for (int x = 0; x < 8; ++x) rank += (f0 + x < fhi && ((vf[x] > ve) || (vf[x] == ve && af[x] > ae))) ? 1 : 0;
                    }
                    for (int off = 1; off < parts; off <<= 1) rank += __shfl_xor(rank, off);
                    if (act && part == 0) {
                        if (rank < omega) nodesLive[rank] = ae;
                        if (rank == omega - 1) { red[64] = ve; }
                    }
                }
                __syncthreads();
                FMARK(2);
                // a tie at the cut: an entry as large as the omega-th one ranks behind it <=> more than omega entries are >= that value
                const double vstar = red[64];
                int ge = 0;
                for (int e = tid; e < Pn; e += nt) ge += (pv[e] >= vstar) ? 1 : 0;
                if (ge) atomicAdd(&sint[1], ge);
                __syncthreads();
                fallback = sint[1] > omega;
            }
            if (fallback) {
                status |= ROMAN_ST_TIE_FALLBACK;
                __syncthreads();
                if (tid == 0) {
                    const int kk = min(omega, L);      // heap capacity bounded by the scratch size
                    heap_select_serial(u, lpAsc + lo, vecOfLive ? vecOfLive + lo : nullptr, L, pd.nA, kk, D.pruned != 0, pv, pidx, nodesOrig);
                    sint[0] = kk;
                }
                __syncthreads();
                nsel = sint[0];
            } else {
                nsel = omega;
                for (int t = tid; t < nsel; t += nt) nodesOrig[t] = nodesLive[t];
            }
            __syncthreads();
        }
    }

    FMARK(3);
    // ---- outputs: associations, pose, stats ------------------------------------------------------
    const int32_t* nodesOrigR = O.nodesOrig + rb;
    const int kout = min(nsel, O.kmax);
    if (nsel > O.kmax) status |= ROMAN_ST_ASSOC_TRUNCATED;
    for (int t = tid; t < kout; t += nt) {
        int i, j;
        decode_assoc(pd, assoc, nodesOrigR[t], i, j);
        O.assoc_out[((int64_t)b * O.kmax + t) * 2 + 0] = i;
        O.assoc_out[((int64_t)b * O.kmax + t) * 2 + 1] = j;
    }
    FMARK(4);
    // pose from ALL selected associations ([REF object_registration.py:110-128], unit weights)
    double Tp[16];
    bool have_pose = false;
    if (nsel >= dim && feats != nullptr && !(status & ROMAN_ST_EMPTY_MAP)) {
        if (tid < 64) {                                         // (wave 0: wave_pose)
            wave_pose(Tp, dim, nsel, tid, [&](int t, double (&a)[3], double (&q)[3]) {
                int i, j;
                decode_assoc(pd, assoc, nodesOrigR[t], i, j);
                const double* pa = feats + (pd.off1 + i) * D.F; const double* pb = feats + (pd.off2 + j) * D.F;
#pragma unroll
                for (int c = 0; c < 3; ++c) { a[c] = c < dim ? pa[c] : 0.0; q[c] = c < dim ? pb[c] : 0.0; }
            });
        }
        FMARK(5);
        FMARK(6);
        have_pose = true;
    } else {
        status |= ROMAN_ST_INSUFFICIENT;
    }
    if (tid == 0) {
        for (int t = 0; t < 16; ++t) O.T_out[(int64_t)b * 16 + t] = have_pose ? Tp[t] : d_nan();
        O.n_assoc_out[b] = kout;
        O.status_out[b] = status;
        O.nSel[b] = nsel;
        if (O.stats_out) O.stats_out[b] = S;
    }
    __syncthreads();
#ifdef ROMAN_SMALL_TIMING
    FMARK(7);
    if (tid == 0 && ((b & 255) == 0 || b < 2)) printf("[finish_one nt=%d] b=%d L=%d nsel=%d cycles: compact %llu ranks %llu ties %llu outputs %llu sums %llu rotation %llu write %llu\n", nt, b, L, nsel,
                                                       tf_[1] - tf_[0], tf_[2] - tf_[1], tf_[3] - tf_[2], tf_[4] - tf_[3], tf_[5] - tf_[4], tf_[6] - tf_[5], tf_[7] - tf_[6]);
#endif
#undef FMARK
}


/*
 * solve_one: CLIPPER findDenseClique on one problem, by one workgroup (k_solve: many mid-size problems of the fallback
 * kind; few large ones go to k_solve_wide, the whole device on one problem).
 * Mirrors oracle_solve() step for step (see there for the restated upstream algorithm):
 * gradF is never stored: it is recombined on the fly from (u, Mu, Cu, d, sum u), bitwise the same
 * value the oracle keeps in its gradF vector.
 */
// MODE 2: u, u_new, Mu, Cu, Mu_new, Cu_new and the diagonal all live in LDS (7 * Lcap doubles);
// MODE 1: only u and u_new (the gathered vectors) do, the row vectors sit in the L2-resident pools;
// MODE 0: nothing fits, everything is in the pools.
template <typename IdxT, int MODE>
__device__ void solve_one(const DevParams& D, int b, const ProbDesc& pd, ProbState* st,
                          const double* __restrict__ feats, const int32_t* __restrict__ assoc,
                          const int32_t* __restrict__ lp, const double* __restrict__ ls,
                          const uint32_t* __restrict__ permPool, const uint32_t* __restrict__ rowPosPool,
                          const uint32_t* __restrict__ sliceWidthPool, const uint32_t* __restrict__ sliceBasePool,
                          const IdxT* __restrict__ colsPool, const double* __restrict__ valsPool,
                          double* __restrict__ vMu, double* __restrict__ vCu,
                          double* __restrict__ vMun, double* __restrict__ vCun,
                          double* __restrict__ gU, double* __restrict__ gUn,
                          int32_t* __restrict__ plpPool /* position -> association index */, double* __restrict__ pldPool /* diagonal by position */,
                          const double* __restrict__ u0, const SolveOut& O,
                          double* sv /* LDS vectors */, int Lcap, double* red, int* sint)
{
    // Every vector is indexed by POSITION (the sorted row order of the layout; the matrix's column labels are positions):
    // element p belongs to the live association perm[p].
    const roman_params_t& P = D.p;
    const int ltid = threadIdx.x, nw = blockDim.x >> 6;
    const int nt = (int)blockDim.x;
    const int tid = ltid;
#define SUM2(a_, b_) block_sum2(a_, b_, red, par, ltid, nw)
    const int L = st[b].L, rb = st[b].rowBase;
    const int64_t lo = pd.liveOff;
    const uint32_t* perm = permPool + lo; const uint32_t* swid = sliceWidthPool + lo; const uint32_t* sbase = sliceBasePool + lo;
    const IdxT* cols = colsPool + st[b].nnzOff; const double* vals = valsPool + st[b].nnzOff;
    double* u = (MODE >= 1) ? sv : gU + rb;
    double* un = (MODE >= 1) ? sv + Lcap : gUn + rb;
    double* Mu = (MODE == 2) ? sv + 2 * Lcap : vMu + rb;
    double* Cu = (MODE == 2) ? sv + 3 * Lcap : vCu + rb;
    double* Mun = (MODE == 2) ? sv + 4 * Lcap : vMun + rb;
    double* Cun = (MODE == 2) ? sv + 5 * Lcap : vCun + rb;
    int32_t* plp = plpPool + lo; double* pld = pldPool + lo;
    for (int p = tid; p < L; p += nt) { const uint32_t k_ = perm[p]; plp[p] = lp[lo + k_]; pld[p] = ls[lo + k_]; }
    __syncthreads();
    const double* sd = pld;                           // diagonal M_pp = single score
    if (MODE == 2) {
        double* sdl = sv + 6 * Lcap;
        for (int p = tid; p < L; p += nt) sdl[p] = pld[p];
        sd = sdl;
    }

    int par = 0;

    int status = ROMAN_ST_OK;
    roman_stats_t S;
    S.n_assoc_in = pd.nA; S.n_live = L; S.nnz_upper = (int64_t)st[b].nnzUpper;
    S.n_pass = 0; S.outer_iters = 0; S.inner_iters = 0; S.ls_trials = 0; S.score = 0.0; S.d_final = 0.0;
    double F = 0.0, d = 0.0;

    if (pd.n1 == 0 || pd.n2 == 0) status |= ROMAN_ST_EMPTY_MAP;
    if (L > 0) {
        // ---- initialisation: u = normalize(M u0 + diag u0) ------------------------------------
        for (int p = tid; p < L; p += nt) u[p] = u0 ? u0[lo + plp[p]] : 1.0;
        __syncthreads();
        if (P.rescale_u0) {
            spmv_sell<IdxT, 4>(u, L, perm, swid, sbase, cols, vals, Mu, Cu, tid, nt); ++S.n_pass;
            __syncthreads();
            for (int p = tid; p < L; p += nt) u[p] = Mu[p] + sd[p] * u[p];
            __syncthreads();
        }
        {
            double ss = 0.0, dummy = 0.0;
            for (int p = tid; p < L; p += nt) ss += u[p] * u[p];
            SUM2(ss, dummy);
            const double nr = sqrt(ss);
            if (nr > 0.0) for (int p = tid; p < L; p += nt) u[p] /= nr;
            __syncthreads();
        }
        spmv_sell<IdxT, 4>(u, L, perm, swid, sbase, cols, vals, Mu, Cu, tid, nt); ++S.n_pass;
        double usum = 0.0;
        { double dummy = 0.0; for (int p = tid; p < L; p += nt) usum += u[p]; SUM2(usum, dummy); }
        // the barrier inside block_sum2 also orders the Mu/Cu stores of spmv_sell before the reads below
        {   // initial d: signed mean of (Mu)_p / Cbu_p over the active set
            double acc = 0.0, cnt = 0.0;
            for (int p = tid; p < L; p += nt) {
                const double up = u[p], Cbu = (usum - Cu[p]) - up;
                if (Cbu > P.eps && up > P.eps) { acc += (Mu[p] + sd[p] * up) / Cbu; cnt += 1.0; }
            }
            SUM2(acc, cnt);
            d = (cnt > 0.0) ? acc / cnt : 0.0;
        }
        // ---- projected gradient ascent with homotopy on d ---------------------------------------
        int i;
        for (i = 0; i < P.maxoliters; ++i) {
            {
                double f = 0.0, dummy = 0.0;
                for (int p = tid; p < L; p += nt) {
                    const double up = u[p];
                    const double g = (((sd[p] + d) * up - d * usum) + Mu[p]) + Cu[p] * d;
                    f += up * g;
                }
                SUM2(f, dummy);
                F = f;
            }
            for (int j = 0; j < P.maxiniters; ++j) {
                double alpha = 1.0, Fnew = 0.0, deltaF = 0.0, unsum = 0.0, du2 = 0.0;
                for (int k = 0; k < P.maxlsiters; ++k) {
                    double ss = 0.0, dummy = 0.0;
                    for (int p = tid; p < L; p += nt) {
                        const double up = u[p];
                        const double g = (((sd[p] + d) * up - d * usum) + Mu[p]) + Cu[p] * d;
                        double t = up + alpha * g;
                        t = t > 0.0 ? t : 0.0;
                        un[p] = t; ss += t * t;
                    }
                    SUM2(ss, dummy);
                    const double nr = sqrt(ss);
                    double s1 = 0.0, dd = 0.0;
                    for (int p = tid; p < L; p += nt) {
                        double t = un[p];
                        if (nr > 0.0) { t /= nr; un[p] = t; }
                        s1 += t;
                        const double df = t - u[p]; dd += df * df;
                    }
                    SUM2(s1, dd);
                    unsum = s1; du2 = dd;
                    spmv_sell<IdxT, 4>(un, L, perm, swid, sbase, cols, vals, Mun, Cun, tid, nt); ++S.n_pass; ++S.ls_trials;
                    __syncthreads();
                    double f = 0.0;
                    for (int p = tid; p < L; p += nt) {
                        const double up = un[p];
                        const double g = (((sd[p] + d) * up - d * unsum) + Mun[p]) + Cun[p] * d;
                        f += up * g;
                    }
                    SUM2(f, dummy);
                    Fnew = f;
                    deltaF = Fnew - F;
                    if (deltaF < -P.eps) alpha *= P.beta; else break;
                }
                const double du = sqrt(du2);
                F = Fnew; usum = unsum;
                { double* t; t = u; u = un; un = t; t = Mu; Mu = Mun; Mun = t; t = Cu; Cu = Cun; Cun = t; }
                ++S.inner_iters;
                if (du < P.tol_u || fabs(deltaF) < P.tol_F) break;
            }
            double acc = 0.0, cnt = 0.0;
            for (int p = tid; p < L; p += nt) {
                const double up = u[p], Cbu = (usum - Cu[p]) - up;
                if (Cbu > P.eps && up > P.eps) { acc += fabs((Mu[p] + sd[p] * up) / Cbu); cnt += 1.0; }
            }
            SUM2(acc, cnt);
            if (cnt > 0.0) d += acc / cnt; else break;
        }
        if (i >= P.maxoliters) status |= ROMAN_ST_MAXITER;
        S.outer_iters = i; S.score = F; S.d_final = d;

        __syncthreads();                         // every element of the final u is in memory
        finish_one(D, b, pd, feats, assoc, plpPool, lp, rowPosPool, permPool, O, u, Mun, (int32_t*)un, (int32_t*)Cun, L, rb, lo, F, status, S, red, sint);
        return;
    }
    finish_one(D, b, pd, feats, assoc, plpPool, lp, rowPosPool, permPool, O, nullptr, nullptr, nullptr, nullptr, L, rb, lo, F, status, S, red, sint);
#undef SUM2
}


template <typename IdxT, int MODE>
__global__ void __launch_bounds__(1024) k_solve(DevParams D, int B, const ProbDesc* __restrict__ probs,
                                                ProbState* __restrict__ st,
                                                const double* __restrict__ feats, const int32_t* __restrict__ assoc,
                                                const int32_t* __restrict__ lp, const double* __restrict__ ls,
                                                const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rowPos,
                                                const uint32_t* __restrict__ sliceWidth, const uint32_t* __restrict__ sliceBase,
                                                const IdxT* __restrict__ cols, const double* __restrict__ vals,
                                                double* __restrict__ vMu, double* __restrict__ vCu,
                                                double* __restrict__ vMun, double* __restrict__ vCun,
                                                double* __restrict__ gU, double* __restrict__ gUn,
                                                int32_t* __restrict__ plp, double* __restrict__ pld,
                                                const double* __restrict__ u0, SolveOut O,
                                                int* __restrict__ queue, int Lcap)
{
    // LDS: [MODE 2: 7 | MODE 1: 2 | MODE 0: 0] vectors of Lcap doubles, 72 doubles of reduction
    // scratch (2 x 32 ping-pong + 8), 4 ints
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NVEC = (MODE == 2) ? 7 : (MODE == 1 ? 2 : 0);
    double* sv = reinterpret_cast<double*>(smem);
    double* red = sv + (size_t)NVEC * Lcap;
    int* sint = reinterpret_cast<int*>(red + 72);
    for (;;) {
        if (threadIdx.x == 0) sint[2] = atomicAdd(queue, 1);
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(sint[2]);
        __syncthreads();
        if (b >= B) break;
        if (__builtin_amdgcn_readfirstlane(st[b].kind) == 1) {   // (else: the stream solver's problem, or a skipped one)
            const ProbDesc pd = probs[b];
            if (MODE > 0 && st[b].L <= Lcap)
                solve_one<IdxT, MODE>(D, b, pd, st, feats, assoc, lp, ls, perm, rowPos, sliceWidth, sliceBase, cols, vals,
                                      vMu, vCu, vMun, vCun, gU, gUn, plp, pld, u0, O, sv, Lcap, red, sint);
            else
                solve_one<IdxT, 0>(D, b, pd, st, feats, assoc, lp, ls, perm, rowPos, sliceWidth, sliceBase, cols, vals,
                                   vMu, vCu, vMun, vCun, gU, gUn, plp, pld, u0, O, sv, Lcap, red, sint);
        }
    }
}

#ifdef ROMAN_SOLVE_TIMING
#define TMARK(slot) do { const unsigned long long t__ = __builtin_readcyclecounter(); tacc[slot] += t__ - tlast; tcnt[slot] += 1; tlast = t__; } while (0)
#else
#define TMARK(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Stream solver (kind 0: L <= STREAM_MAXL, 16-bit position indices, upper-triangle quad layout).
//
// One workgroup of NW waves per problem.  Thread t OWNS the vector elements p = t, t + NT, t + 2 NT (positions):
// u, M u, C u, the diagonal and the trial vector with its products live in that thread's registers.  LDS holds
// only what the SpMV touches by index: the vector being multiplied (xg) and two accumulator arrays.
//
// SpMV on the upper triangle, pull + push.  A stored entry (p, q, v), p < q, serves both triangles:
//      (M x)_p += v x_q      pulled: the row's lane gathers x_q from LDS and accumulates in registers,
//      (M x)_q += v x_p      pushed: an LDS atomic add into the accumulator of q,
// and likewise (C x) with v replaced by 1.  Every pass therefore moves 10 bytes per non-zero of the UPPER
// triangle, and — positions being ranks by degree — a vector whose support lies in the first S slices needs only
// those S slices: rows beyond have x_p = 0 (nothing to push) and only columns q > p beyond the support (nothing
// to pull).  The solver's long tail runs on the clique found so far, i.e. on the first one or two slices; no
// column-compacted copies of the matrix are needed.
//
// Exact accumulation.  Atomic floating-point adds would make the sums depend on the order in which waves reach
// the LDS.  Every term is instead rounded ONCE to a fixed-point integer, rint(v x 2^s) (s chosen per pass from
// max x so that a term is below 2^49 and any sum below 2^62), and integers are added — in registers for the
// pulled part, with ds_add_u64 for the pushed part and for the flush of a row's pulled sum.  Integer addition is
// associative: the result does not depend on the order, on the number of waves, or on how the stream is split.
// The rounding unit 2^-s is 2^-48 relative to the largest element of x, the same order as the rounding of an f64
// summation.  The conversion is one fma against 2^52 + 2^51 (the integer appears in the low mantissa bits).
//
// Fused passes (round 5).  Inside the line search only the sum (M + d C) x enters the gradient; M x and C x are needed
// apart only where d itself is updated (once per outer iteration).  A line-search pass therefore accumulates ONE
// fixed-point sum per element with the weight v + d (v alone where C_pq = 0): one LDS gather and ONE ds_add_u64 per
// stored pair instead of one and two; the d update is preceded by a SPLIT pass (both sums) over the accepted vector.
// The same order is stated in the oracle (oracle_set_pass_mode(1), its default).
//
// One balanced stream: the S slices of a pass are contiguous in memory (quads); wave w streams the quads
// [w T / NW, (w+1) T / NW) with ST_D quads (9 wide loads) in flight per lane, whatever slices the range covers
// (lane = row slot of the current slice) and flushes its pulled sums when it leaves a slice or its range.
// ---------------------------------------------------------------------------------------------
#ifndef ROMAN_SOLVE_WAVES
#define ROMAN_SOLVE_WAVES 8              // waves of the stream solver's workgroup (one problem per workgroup; measured: 8 beats 16)
#endif
constexpr int ST_D = 3;                  // quads in flight per lane
constexpr int ST_MAXSL = STREAM_MAXL / 64;
constexpr int SMALL_MAXL = 128;           // live associations the one-wave-per-problem instantiation of k_solve_up takes
constexpr int LEAN_MAXL = STREAM_MAXL;    // live associations the 128-register instantiation takes (two workgroups per compute unit)
constexpr int LEAN_D = 2;                 // ... and its quads in flight per lane
constexpr int DEEP_D = 4;                 // quads in flight per lane of the ROMAN_SOLVE_DEEP=1 instantiation (round 5: four — six no longer fit 256 registers
                                          // next to the gathers held one quad ahead; measured 0.953-0.961 against 0.941-0.950 ms per launch: no gain)
constexpr int COO_E = 6;                  // one-wave instantiation: stored pairs a lane holds in registers (coordinate form)
constexpr int COO_CAP = 64 * COO_E;       // ... per problem; larger matrices take the quad stream
constexpr uint32_t ST_CZ = 0x8000u, ST_MASK = 0x7fffu;
constexpr unsigned long long FX_MAGIC_BITS = 0x4338000000000000ull;     // 2^52 + 2^51
#define FX_MAGIC 6755399441055744.0

// Explicit address spaces for the hot pointers of the stream: FLAT accesses would count against lgkmcnt as well and
// every wait for an LDS gather would drain the prefetched matrix loads.
#define ROMAN_GLOBAL __attribute__((address_space(1)))
#define ROMAN_LDS __attribute__((address_space(3)))
typedef const ROMAN_GLOBAL unsigned long long* g_quad_cp;    // 4 x u16 column indices of one lane
typedef const ROMAN_GLOBAL dbl2_t* g_pair_cp;                // 2 x f64 values of one lane
typedef const ROMAN_LDS double* l_vec_cp;                    // gathered vector in LDS
typedef ROMAN_LDS unsigned long long* l_acc_p;               // fixed-point accumulators in LDS

// wave-uniform values that the compiler cannot prove uniform (read from LDS, derived from threadIdx):
// force them into SGPRs so that loop control and matrix addressing run on the scalar unit
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWMASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum / maximum (of non-negative values) over the 64 lanes, valid in lane 63 (fixed tree: pairs, quads, rows of 16,
// then across rows; rows a row_bcast does not reach receive 0, the neutral element of both)
__device__ __forceinline__ double wave_sum63(double v)
{
    v += dpp_mov<0xB1, 0xf>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov<0x124, 0xf>(v);         // row_ror:4
    v += dpp_mov<0x128, 0xf>(v);         // row_ror:8
    v += dpp_mov<0x142, 0xa>(v);         // row_bcast:15 -> rows 1,3
    v += dpp_mov<0x143, 0xc>(v);         // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ double wave_max63(double v)
{
    v = fmax(v, dpp_mov<0xB1, 0xf>(v));
    v = fmax(v, dpp_mov<0x4E, 0xf>(v));
    v = fmax(v, dpp_mov<0x124, 0xf>(v));
    v = fmax(v, dpp_mov<0x128, 0xf>(v));
    v = fmax(v, dpp_mov<0x142, 0xa>(v));
    v = fmax(v, dpp_mov<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ double readlane63(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// NS sums and NM maxima (of non-negative values) over the NW waves of the block, identical in every thread; fixed
// reduction tree.  `red`: two ping-pong areas of NW*8 doubles (a buffer is rewritten only after the barrier of the
// next call).  The barrier inside also publishes whatever the caller wrote to LDS before the call.
constexpr int RED_STRIDE = 8;
// doubles of reduction scratch a stream-solver workgroup of NW waves owns: the solver's ping-pong areas, and at least the 72 that
// finish_one() uses whatever the workgroup size (two areas of 32 + the slot red[64]).  (With 2 * NW * RED_STRIDE + 8 = 24 the
// one-wave instantiation wrote red[32..33] over its unused slice table and red[64] BEHIND its LDS allocation.)
constexpr int red_doubles(int NW) { return 2 * NW * RED_STRIDE + 8 > 72 ? 2 * NW * RED_STRIDE + 8 : 72; }
template <int NW, int NS, int NM>
__device__ __forceinline__ void block_red(double (&sv)[NS > 0 ? NS : 1], double (&mv)[NM > 0 ? NM : 1], double* red, int& par, int tid)
{
    static_assert(NS + NM <= RED_STRIDE, "reduction scratch");
    static_assert(NW == 1 || NW == 4 || NW == 8 || NW == 16, "cross-wave butterfly");
    if (NW == 1) {                                              // a single wave: the wave reduction is the block reduction
#pragma unroll
        for (int i = 0; i < NS; ++i) sv[i] = uni(readlane63(wave_sum63(sv[i])));
#pragma unroll
        for (int i = 0; i < NM; ++i) mv[i] = uni(readlane63(wave_max63(mv[i])));
        __syncthreads();                                        // (publishes the caller's LDS writes like the general path; one wave: no wait)
        return;
    }
    double* rr = red + NW * RED_STRIDE * par;
    par ^= 1;
    const int w = uni(tid >> 6);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double s = readlane63(wave_sum63(sv[i]));
        if ((tid & 63) == 0) rr[RED_STRIDE * w + i] = s;
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const double m = readlane63(wave_max63(mv[i]));
        if ((tid & 63) == 0) rr[RED_STRIDE * w + NS + i] = m;
    }
    __syncthreads();
    // cross-wave step: lane l takes the partial of wave l % NW; the NW lanes of a DPP row are combined with a fixed
    // butterfly (every lane ends with the total), the result moves to an SGPR
    const int src = (tid & 63) & (NW - 1);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double v = rr[RED_STRIDE * src + i];
        v += dpp_mov<0xB1, 0xf>(v);          // quad_perm [1,0,3,2]
        v += dpp_mov<0x4E, 0xf>(v);          // quad_perm [2,3,0,1]
        if (NW > 4) v += dpp_mov<0x124, 0xf>(v);         // row_ror:4
        if (NW > 8) v += dpp_mov<0x128, 0xf>(v);     // row_ror:8
        sv[i] = uni(v);
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        double v = rr[RED_STRIDE * src + NS + i];
        v = fmax(v, dpp_mov<0xB1, 0xf>(v));
        v = fmax(v, dpp_mov<0x4E, 0xf>(v));
        if (NW > 4) v = fmax(v, dpp_mov<0x124, 0xf>(v));
        if (NW > 8) v = fmax(v, dpp_mov<0x128, 0xf>(v));
        mv[i] = uni(v);
    }
}

// The same reduction of NS sums in two halves, for a result that is not needed before the caller's NEXT workgroup barrier anyway:
// block_post() leaves every wave's partial in the ping-pong area (no barrier), block_collect() — behind any later __syncthreads()
// of the caller, and before the second block_red() / block_post() after the post — finishes it.  Same tree, same bits as block_red().
template <int NW, int NS>
__device__ __forceinline__ const double* block_post(const double (&sv)[NS], double* red, int& par, int tid)
{
    static_assert(NW == 4 || NW == 8 || NW == 16, "cross-wave butterfly");
    double* rr = red + NW * RED_STRIDE * par;
    par ^= 1;
    const int w = uni(tid >> 6);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double s = readlane63(wave_sum63(sv[i]));
        if ((tid & 63) == 0) rr[RED_STRIDE * w + i] = s;
    }
    return rr;
}
template <int NW, int NS>
__device__ __forceinline__ void block_collect(double (&sv)[NS], const double* rr, int tid)
{
    const int src = (tid & 63) & (NW - 1);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double v = rr[RED_STRIDE * src + i];
        v += dpp_mov<0xB1, 0xf>(v);          // quad_perm [1,0,3,2]
        v += dpp_mov<0x4E, 0xf>(v);          // quad_perm [2,3,0,1]
        if (NW > 4) v += dpp_mov<0x124, 0xf>(v);         // row_ror:4
        if (NW > 8) v += dpp_mov<0x128, 0xf>(v);     // row_ror:8
        sv[i] = uni(v);
    }
}

// fixed-point accumulator -> double (the accumulated integer is below 2^62)
__device__ __forceinline__ double fx_decode(unsigned long long a, double inv)
{
    return fma((double)(uint32_t)(a >> 32), 4294967296.0, (double)(uint32_t)a) * inv;
}

template <int NW, bool HASCZ, int MAXL, int DEPTH = ST_D, bool LEAN = false, bool COOONLY = false /* the matrix is ALWAYS a pre-built coordinate list (k_small): no quad stream in the code */,
          bool BUDGET = false /* pass budgets / suspended problems (SolveCont): an instantiation of its own — the extra live state cost the default one 4 % */>
__device__ void solve_up(const DevParams& D, int b, const ProbDesc& pd, ProbState* st,
                         const double* __restrict__ feats, const int32_t* __restrict__ assoc,
                         const int32_t* __restrict__ plp /* position -> association index */, const int32_t* __restrict__ lpAsc,
                         const uint32_t* __restrict__ rowPosPool, const double* __restrict__ pld /* diagonal, position order */,
                         const uint32_t* __restrict__ sliceBasePool,
                         const uint16_t* __restrict__ colsPool, const double* __restrict__ valsPool,
                         const double* __restrict__ u0, const SolveOut& O,
                         double* xg /* [Lc] */, unsigned long long* accM /* [Lc] */, unsigned long long* accC /* [Lc] */, int Lc,
                         uint32_t* cumQ /* [ST_MAXSL + 1] */, double* red, int* sint, unsigned char* cooLds /* one-wave instantiation: COO_CAP * 12 bytes */,
                         int cooPre = -1 /* >= 0: cooLds already holds that many entries (k_small); no quad layout exists for the problem */,
                         int rbPre = -1 /* >= 0: offset of the problem in the row pools (k_small runs before k_rowbase) */,
                         const SolveCont* cont = nullptr /* pass budget / suspended problems (general instantiation only) */, int resumeSlot = -1)
{
    constexpr int NT = NW * 64;
#ifdef ROMAN_SOLVE_TIMING
    const unsigned long long tentry_ = __builtin_readcyclecounter();
#endif
#ifdef ROMAN_SMALL_TIMING
    const unsigned long long tw0_ = wall_clock64();
#endif
    constexpr int KMAX = (MAXL + NT - 1) / NT;                 // elements per thread (the kernel takes problems of up to MAXL live associations)
    // The three LDS vectors lie at a FIXED distance from each other — accM == xg + LCAP, accC == xg + 2 LCAP (the callers lay them
    // out so; Lc == LCAP) —: the stream addresses the accumulators of a column through the column's gather address + an immediate
    constexpr int LCAP = MAXL + 64;
    const roman_params_t& P = D.p;
    // the solver's parameters as scalars (the argument block sits in scratch memory: its address is taken for the shared tail)
    const double p_eps = uni(P.eps), p_beta = uni(P.beta), p_tol_u = uni(P.tol_u), p_tol_F = uni(P.tol_F);
    const int p_maxin = uni(P.maxiniters), p_maxout = uni(P.maxoliters), p_maxls = uni(P.maxlsiters), p_rescale = uni(P.rescale_u0);
    const int tid = threadIdx.x, lane = tid & 63, w = uni(tid >> 6);
    // FOR_K: the thread's elements; in the one-wave instantiation only those that exist in THIS problem (k < ceil(L / 64): a problem
    // of 60 live associations uses one of the two element slots — the other is zero and stays zero, its half of the element-wise
    // work, the bulk of a pass at this size, is skipped; in the general instantiation the same guard costs 20 registers and
    // buys a sixth of 8 % of the time: not applied).  FOR_K_ALL: every slot (initialisation)
#define FOR_K_ALL(k_, p_) _Pragma("unroll") for (int k_ = 0; k_ < KMAX; ++k_) if ([[maybe_unused]] const int p_ = tid + k_ * NT; true)
#define FOR_K(k_, p_) _Pragma("unroll") for (int k_ = 0; k_ < KMAX; ++k_) if ([[maybe_unused]] const int p_ = tid + k_ * NT; NW != 1 || k_ < kUsed)
    const int L = uni(st[b].L), rb = rbPre >= 0 ? rbPre : uni(st[b].rowBase);
    const int kUsed = uni((L + NT - 1) / NT);
    const int64_t lo = pd.liveOff;
    const int nsl = (L + 63) >> 6;
    int par = 0;

    int status = ROMAN_ST_OK;
    roman_stats_t S;
    S.n_assoc_in = pd.nA; S.n_live = L; S.nnz_upper = (int64_t)st[b].nnzUpper;
    S.n_pass = 0; S.outer_iters = 0; S.inner_iters = 0; S.ls_trials = 0; S.score = 0.0; S.d_final = 0.0;
    int n_pass = 0, ls_trials = 0, inner_iters = 0;
    double F = 0.0, d = 0.0;
    if (pd.n1 == 0 || pd.n2 == 0) status |= ROMAN_ST_EMPTY_MAP;
    if (L <= 0) {
        finish_one(D, b, pd, feats, assoc, plp, lpAsc, rowPosPool, nullptr, O, nullptr, nullptr, nullptr, nullptr, L, rb, lo, F, status, S, red, sint);
        return;
    }
#ifdef ROMAN_SOLVE_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tcnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    g_quad_cp cbase = (g_quad_cp)(colsPool + st[b].nnzOff) + lane;
    g_pair_cp vbase = (g_pair_cp)(valsPool + st[b].nnzOff) + lane;
    l_vec_cp xl = (l_vec_cp)xg;
    l_acc_p aM = (l_acc_p)accM; l_acc_p aC = (l_acc_p)accC;

    // ---- per-problem set-up: quad prefix of the slices, owned elements, clean LDS --------------------------
    __syncthreads();
    // slice table: lane s of every wave holds the first quad of slice s (lane nsl: the total), read with v_readlane
    const uint32_t cqv = cooPre >= 0 ? 0u : ((lane < nsl) ? (sliceBasePool[lo + lane] >> 8) : (st[b].nnzCap >> 8));
#define CUMQ(s_) ((uint32_t)__builtin_amdgcn_readlane((int)cqv, (s_)))
    for (int p = tid; p < Lc; p += NT) { xg[p] = 0.0; accM[p] = 0ull; accC[p] = 0ull; }
    // Wu: the fused product (M + d C) u of the current u at the current d; Mn / Cn: the products of the pass just done —
    // a fused pass leaves (M + d C) x in Mn and nothing in Cn, a split pass M x and C x
    double u[KMAX], Wu[KMAX], sd[KMAX], tk[KMAX], Mn[KMAX], Cn[KMAX];
    FOR_K_ALL(k, p) {
        const bool in = p < L;
        sd[k] = in ? pld[lo + p] : 0.0;
        u[k] = in ? (u0 ? u0[lo + plp[lo + p]] : 1.0) : 0.0;
        Wu[k] = tk[k] = Mn[k] = Cn[k] = 0.0;
    }
    __syncthreads();

    // ---- one-wave instantiation: the matrix as a COORDINATE list in registers ----------------------------------------
    // A problem of the reference's demo scale has ~60 live associations and ~140 stored pairs; its one slice of 64 rows is
    // as wide as its longest row, so those 140 entries occupy ~1500 slots of the quad layout, and every pass walks them all.
    // The wave compacts the real entries once (ballot ranks, through LDS) into COO_E registers per lane — (row, column, flag)
    // in one word + the value — and a pass is COO_E rounds of two gathers and four pushes, no memory traffic at all.  The sums
    // are the same integers in another order: identical bits.  More than 64 * COO_E stored pairs: the quad stream below.
    [[maybe_unused]] uint32_t cpq[COO_E]; [[maybe_unused]] double cv[COO_E];
    [[maybe_unused]] int cooRounds = -1;                       // -1: not in use
    if constexpr (NW == 1) {
        uint32_t* lpq = reinterpret_cast<uint32_t*>(cooLds); double* lv = reinterpret_cast<double*>(cooLds + 4 * COO_CAP);
        const uint32_t Tq = CUMQ(nsl);
        const uint32_t q1 = nsl > 1 ? CUMQ(1) : Tq;             // (L <= 128: at most two slices)
        uint32_t cnt = 0;
        if (cooPre >= 0) {                                      // the caller (k_small) compacted the entries itself
            cnt = (uint32_t)cooPre;
            cooRounds = (int)((cnt + 63u) >> 6);
#pragma unroll
            for (int e = 0; e < COO_E; ++e) {
                // (lane l holds the CONSECUTIVE entries l R .. l R + R - 1: the list is row-major, so the 64 lanes of a round
                //  push onto 64 different rows instead of onto the few rows 64 consecutive entries belong to)
                const uint32_t at = (uint32_t)lane * (uint32_t)cooRounds + (uint32_t)e;
                const bool have = e < cooRounds && at < cnt;
                const uint32_t dummy = (uint32_t)L + (uint32_t)lane;
                cpq[e] = have ? lpq[at] : (dummy | (dummy << 8) | 0x10000u);
                cv[e] = have ? lv[at] : 0.0;
            }
            __syncthreads();
        } else
        if (!COOONLY && (unsigned long long)st[b].nnzUpper <= (unsigned long long)COO_CAP && nsl <= 2 && !(D.solve_flags & 1)) {
            for (uint32_t qq = 0; qq < Tq; ++qq) {
                const unsigned long long cw = cbase[(size_t)qq * 64];
                const dbl2_t v0 = vbase[(size_t)(2 * qq) * 64], v1 = vbase[(size_t)(2 * qq + 1) * 64];
                const uint32_t row = (qq >= q1 ? 64u : 0u) + (uint32_t)lane;
                const double v4[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lab = (uint32_t)(cw >> (16 * e)) & 0xffffu, col = lab & ST_MASK;
                    const bool real = col < (uint32_t)L;        // (padding points at the dummy elements L + lane)
                    const unsigned long long m_ = __ballot(real);
                    if (real) {
                        const uint32_t at = cnt + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u));
                        if (at < (uint32_t)COO_CAP) { lpq[at] = row | (col << 8) | ((lab & ST_CZ) ? 0x10000u : 0u); lv[at] = v4[e]; }
                    }
                    cnt += (uint32_t)__popcll(m_);
                }
            }
            __syncthreads();
            if (cnt <= (uint32_t)COO_CAP) {
                cooRounds = (int)((cnt + 63u) >> 6);
#pragma unroll
                for (int e = 0; e < COO_E; ++e) {
                    const uint32_t at = (uint32_t)lane * (uint32_t)cooRounds + (uint32_t)e;   // (consecutive entries per lane, see above)
                    const bool have = e < cooRounds && at < cnt;
                    const uint32_t dummy = (uint32_t)L + (uint32_t)lane;                // inert: value 0 between two dummy elements
                    cpq[e] = have ? lpq[at] : (dummy | (dummy << 8) | 0x10000u);
                    cv[e] = have ? lv[at] : 0.0;
                }
            }
            __syncthreads();
        }
    }

    // ---- products of the vector x held in tk[] (elements >= 0), with xmax = max x and mp1 = 1 + the largest position with
    //      x > 0 (0: x is the zero vector):  split: (M x, C x) -> (Mn, Cn);  fused: (M + de C) x -> Mn (de >= 0; 0: M x alone)
    [[maybe_unused]] const double* pendRR = nullptr; [[maybe_unused]] bool pend = false;    // a posted reduction of (sum u', |u' - u|^2) waiting for its barrier
    double usum = 0.0, alpha = 1.0, unsum = 0.0, du2 = 0.0, xmaxT = 0.0;
    auto collect_pending = [&]() {
        if constexpr (NW != 1) {
            if (pend) { double q2[2]; block_collect<NW, 2>(q2, pendRR, tid); unsum = q2[0]; du2 = q2[1]; pend = false; }
        }
    };
    auto spmv = [&](double xmax, int mp1, const bool split, const double de_) {
        TMARK(3);
        if (!(xmax > 0.0) || mp1 <= 0) {                       // zero vector: zero products, nothing to publish
            if constexpr (NW != 1) { if (pend) { __syncthreads(); collect_pending(); } }
            FOR_K(k, p) { Mn[k] = 0.0; Cn[k] = 0.0; }
            ++n_pass;
            return;
        }
        const double de = (!split && de_ > 0.0) ? de_ : 0.0;    // what is added to a weight (wave-uniform)
        // scale: the largest term w x 2^s stays below 2^49: x_max 2^s in [2^48, 2^49) when the weights are those of M (<= 1),
        // and a further factor 2^(exponent(1 + de) + 1) >= 1 + de down when de is added to them
        int e_ = (int)((__double_as_longlong(xmax) >> 52) & 0x7ff) - 1023;
        int s_ = 48 - e_;
        if (de > 0.0) s_ -= (int)((__double_as_longlong(1.0 + de) >> 52) & 0x7ff) - 1023 + 1;
        s_ = s_ > 960 ? 960 : (s_ < -960 ? -960 : s_);
        const double sc = bits_f64((unsigned long long)(1023 + s_) << 52), inv = bits_f64((unsigned long long)(1023 - s_) << 52);
        // This wave's range of the stream and its first DEPTH quads, requested BEFORE the vector is published: the matrix does
        // not depend on x, and a narrow pass is a dozen quads per wave — four or five dependent round trips to the L2 —, so the
        // first of them now runs under the publish, the barrier and its skew instead of behind them.  (Unconditional loads at
        // clamped indices, as in the loop below: the wait counters stay exact.  No quads at all: any valid address.)
        bool streamed = false;
        if constexpr (NW == 1) streamed = cooRounds >= 0;       // the matrix is in registers: pushes only
        const int Sx = (COOONLY || streamed) ? 0 : min(nsl, (mp1 + 63) >> 6);
        const uint32_t T = COOONLY ? 0u : CUMQ(Sx);
        const uint32_t qs = (uint32_t)(((unsigned long long)T * (unsigned)w) / NW), qe = (uint32_t)(((unsigned long long)T * (unsigned)(w + 1)) / NW);
        [[maybe_unused]] unsigned long long rc[DEPTH]; [[maybe_unused]] dbl2_t rv0[DEPTH], rv1[DEPTH];
        if constexpr (!COOONLY) {
            const uint32_t qlast = (qe > qs ? qe : qs + 1u) - 1u;
            g_quad_cp cb0 = T ? cbase : (g_quad_cp)colsPool + lane;
            g_pair_cp vb0 = T ? vbase : (g_pair_cp)valsPool + lane;
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                const uint32_t qq = min(qs + (uint32_t)t, qlast);
                rc[t] = cb0[(size_t)qq * 64];
                rv0[t] = vb0[(size_t)(2 * qq) * 64]; rv1[t] = vb0[(size_t)(2 * qq + 1) * 64];
            }
        }
        FOR_K(k, p) if (p < L) xg[p] = tk[k] * sc;
        __syncthreads();                                        // the scaled vector is published; accumulators are clean
        collect_pending();                                      // (the trial vector's sums, posted by finish_trial in front of this barrier)
        TMARK(6);
        if constexpr (NW == 1) {
            if (cooRounds >= 0) {
#pragma unroll
                for (int e = 0; e < COO_E; ++e) {
                    if (e < cooRounds) {
                        const uint32_t p_ = cpq[e] & 0xffu, q_ = (cpq[e] >> 8) & 0xffu;
                        const bool cz = HASCZ && (cpq[e] & 0x10000u);
                        const double xp_ = xl[p_], xq_ = xl[q_];
                        const double w_ = cz ? cv[e] : cv[e] + de;      // (split: de == 0, the weight of M)
                        if (xq_ != 0.0) {
                            __hip_atomic_fetch_add(aM + p_, (unsigned long long)__double_as_longlong(fma(w_, xq_, FX_MAGIC)) - FX_MAGIC_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (split && !cz) __hip_atomic_fetch_add(aC + p_, (unsigned long long)__double_as_longlong(xq_ + FX_MAGIC) - FX_MAGIC_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (xp_ != 0.0) {
                            __hip_atomic_fetch_add(aM + q_, (unsigned long long)__double_as_longlong(fma(w_, xp_, FX_MAGIC)) - FX_MAGIC_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (split && !cz) __hip_atomic_fetch_add(aC + q_, (unsigned long long)__double_as_longlong(xp_ + FX_MAGIC) - FX_MAGIC_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        __builtin_amdgcn_sched_barrier(0);      // (one round at a time: hoisting all six rounds' gathers costs a wave of occupancy)
                    }
                }
            }
        }
        // The stream itself, once per kind of pass (SPLIT a compile-time constant: the line search's fused passes carry no code
        // for the second sum).  A wave is alone with one other on its SIMD: nothing hides a latency for it, so the four gathers of
        // quad q + 1 are requested BEFORE quad q is worked on (its labels arrived with the ring), and the accumulators sit at fixed
        // distances LCAP behind the gathered vector — gather and push of a column share one address register.
        auto stream = [&](auto split_tag) {
            constexpr bool SPLIT = decltype(split_tag)::value;
            int s = 0;
            {   // largest s with cumQ[s] <= qs (skips empty slices)
                int lo_ = 0, hi_ = Sx;
                while (hi_ - lo_ > 1) { const int mid_ = (lo_ + hi_) >> 1; if (CUMQ(mid_) <= qs) lo_ = mid_; else hi_ = mid_; }
                s = lo_;
            }
            uint32_t nextB = CUMQ(s + 1);
            // ring of DEPTH quads in flight per lane (filled above, in front of the barrier); loads are issued unconditionally
            // (clamped index) so that the wait counters stay exact
            unsigned long long smI = 0ull; [[maybe_unused]] unsigned long long scI = 0ull;   // pulled sums of this lane's row: sum of bits(MAGIC + term)
            uint32_t pieceQ = qs;                               // first quad of the current piece
            double xp = xl[s * 64 + lane];                      // this lane's row element (scaled); rows >= L read zeros
            double xpn = xl[(s + 1) * 64 + lane];               // ... and the next slice's, fetched ahead of the transition
            [[maybe_unused]] unsigned long long iCp = (unsigned long long)__double_as_longlong(xp + FX_MAGIC) - FX_MAGIC_BITS;
            // gathered elements of a quad: addresses (the accumulators of the same columns lie LCAP and 2 LCAP elements behind) and values
            l_vec_cp ga[4], gn[4]; double xa[4], xn[4];
#define UP_GATHER(C_, G_, X_)                                                                               \
            {                                                                                               \
                const uint32_t lo__ = (uint32_t)(C_), hi__ = (uint32_t)((C_) >> 32);                        \
                (G_)[0] = xl + (lo__ & ST_MASK); (G_)[1] = xl + ((lo__ >> 16) & ST_MASK);                   \
                (G_)[2] = xl + (hi__ & ST_MASK); (G_)[3] = xl + ((hi__ >> 16) & ST_MASK);                   \
                (X_)[0] = *(G_)[0]; (X_)[1] = *(G_)[1]; (X_)[2] = *(G_)[2]; (X_)[3] = *(G_)[3];             \
            }
#define UP_PUSH(G_, OFF_, V_) __hip_atomic_fetch_add((l_acc_p)(G_) + (OFF_), (V_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define UP_CONSUME(Q_, C_, V0_, V1_)                                                                        \
            {                                                                                               \
                [[maybe_unused]] const uint32_t clo = (uint32_t)(C_), chi = (uint32_t)((C_) >> 32);         \
                /* weights of this pass: v (split; fused with de == 0) or v + de, v alone where C_pq == 0 (inert entries  \
                   carry the flag when HASCZ: they keep their 0) */                                         \
                const double w0 = SPLIT ? (V0_).x : ((HASCZ && (clo & ST_CZ)) ? (V0_).x : (V0_).x + de);    \
                const double w1 = SPLIT ? (V0_).y : ((HASCZ && (clo & (ST_CZ << 16))) ? (V0_).y : (V0_).y + de); \
                const double w2 = SPLIT ? (V1_).x : ((HASCZ && (chi & ST_CZ)) ? (V1_).x : (V1_).x + de);    \
                const double w3 = SPLIT ? (V1_).y : ((HASCZ && (chi & (ST_CZ << 16))) ? (V1_).y : (V1_).y + de); \
                smI += (unsigned long long)__double_as_longlong(fma(w0, xa[0], FX_MAGIC));                  \
                smI += (unsigned long long)__double_as_longlong(fma(w1, xa[1], FX_MAGIC));                  \
                smI += (unsigned long long)__double_as_longlong(fma(w2, xa[2], FX_MAGIC));                  \
                smI += (unsigned long long)__double_as_longlong(fma(w3, xa[3], FX_MAGIC));                  \
                if constexpr (SPLIT) {                                                                      \
                    if (HASCZ) {                                                                            \
                        scI += (unsigned long long)__double_as_longlong(((clo & ST_CZ) ? 0.0 : xa[0]) + FX_MAGIC); \
                        scI += (unsigned long long)__double_as_longlong(((clo & (ST_CZ << 16)) ? 0.0 : xa[1]) + FX_MAGIC); \
                        scI += (unsigned long long)__double_as_longlong(((chi & ST_CZ) ? 0.0 : xa[2]) + FX_MAGIC); \
                        scI += (unsigned long long)__double_as_longlong(((chi & (ST_CZ << 16)) ? 0.0 : xa[3]) + FX_MAGIC); \
                    } else {          /* the only C-flagged entries are inert: they gather a zero */        \
                        scI += (unsigned long long)__double_as_longlong(xa[0] + FX_MAGIC);                  \
                        scI += (unsigned long long)__double_as_longlong(xa[1] + FX_MAGIC);                  \
                        scI += (unsigned long long)__double_as_longlong(xa[2] + FX_MAGIC);                  \
                        scI += (unsigned long long)__double_as_longlong(xa[3] + FX_MAGIC);                  \
                    }                                                                                       \
                }                                                                                           \
                if (xp != 0.0) {                      /* push this row's element to the columns */          \
                    UP_PUSH(ga[0], LCAP, (unsigned long long)__double_as_longlong(fma(w0, xp, FX_MAGIC)) - FX_MAGIC_BITS); \
                    UP_PUSH(ga[1], LCAP, (unsigned long long)__double_as_longlong(fma(w1, xp, FX_MAGIC)) - FX_MAGIC_BITS); \
                    UP_PUSH(ga[2], LCAP, (unsigned long long)__double_as_longlong(fma(w2, xp, FX_MAGIC)) - FX_MAGIC_BITS); \
                    UP_PUSH(ga[3], LCAP, (unsigned long long)__double_as_longlong(fma(w3, xp, FX_MAGIC)) - FX_MAGIC_BITS); \
                    if constexpr (SPLIT) {                                                                  \
                        if (!HASCZ || !(clo & ST_CZ)) UP_PUSH(ga[0], 2 * LCAP, iCp);                        \
                        if (!HASCZ || !(clo & (ST_CZ << 16))) UP_PUSH(ga[1], 2 * LCAP, iCp);                \
                        if (!HASCZ || !(chi & ST_CZ)) UP_PUSH(ga[2], 2 * LCAP, iCp);                        \
                        if (!HASCZ || !(chi & (ST_CZ << 16))) UP_PUSH(ga[3], 2 * LCAP, iCp);                \
                    }                                                                                       \
                }                                                                                           \
                if ((Q_) + 1 == nextB || (Q_) + 1 == qe) {            /* end of this slice's piece: flush the pulled sums */ \
                    const unsigned long long nterm = (unsigned long long)(((Q_) + 1 - pieceQ) * 4u) * FX_MAGIC_BITS; \
                    UP_PUSH(xl + (s * 64 + lane), LCAP, smI - nterm);                                       \
                    if constexpr (SPLIT) UP_PUSH(xl + (s * 64 + lane), 2 * LCAP, scI - nterm);              \
                    smI = 0ull; scI = 0ull; pieceQ = (Q_) + 1;                                              \
                    if ((Q_) + 1 < qe) {                                                                    \
                        const int s_old_ = s;                                                               \
                        do { ++s; nextB = CUMQ(s + 1); } while (nextB <= (Q_) + 1);                         \
                        xp = (s == s_old_ + 1) ? xpn : xl[s * 64 + lane];                                   \
                        xpn = xl[(s + 1) * 64 + lane];                                                      \
                        iCp = (unsigned long long)__double_as_longlong(xp + FX_MAGIC) - FX_MAGIC_BITS;      \
                    }                                                                                       \
                }                                                                                           \
            }
            UP_GATHER(rc[0], ga, xa)
            uint32_t q0 = qs;
            for (; q0 + DEPTH <= qe; q0 += DEPTH) {
#pragma unroll
                for (int t = 0; t < DEPTH; ++t) {
                    const uint32_t q = q0 + t;
                    const unsigned long long c = rc[t];
                    const dbl2_t v0 = rv0[t], v1 = rv1[t];
                    const uint32_t qn = min(q + (uint32_t)DEPTH, qe - 1u);
                    rc[t] = cbase[(size_t)qn * 64];
                    rv0[t] = vbase[(size_t)(2 * qn) * 64]; rv1[t] = vbase[(size_t)(2 * qn + 1) * 64];
                    UP_GATHER(rc[(t + 1) % DEPTH], gn, xn)           // quad q + 1 (behind the last quad: a duplicate, unused)
                    UP_CONSUME(q, c, v0, v1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ga[e] = gn[e]; xa[e] = xn[e]; }
                }
            }
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {                    // tail: fewer than DEPTH quads, already in the ring
                const uint32_t q = q0 + t;
                if (q < qe) {
                    UP_GATHER(rc[(t + 1) % DEPTH], gn, xn)
                    UP_CONSUME(q, rc[t], rv0[t], rv1[t])
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ga[e] = gn[e]; xa[e] = xn[e]; }
                }
            }
#undef UP_CONSUME
#undef UP_PUSH
#undef UP_GATHER
        };
        if (!COOONLY && qs < qe) {
            if (split) stream(std::true_type{}); else stream(std::false_type{});
        }
        TMARK(Sx <= 2 ? 4 : 0);                                 // (timing build: the stream of narrow passes — at most two slices — apart)
        __syncthreads();                                        // every contribution has landed
        TMARK(1);
        FOR_K(k, p) {
            if (p < L) {
                Mn[k] = fx_decode(accM[p], inv); accM[p] = 0ull;            // clean for the next pass (published by its barrier)
                if (split) { Cn[k] = fx_decode(accC[p], inv); accC[p] = 0ull; } else Cn[k] = 0.0;
                // LEAN: the multiplied vector was not kept in registers across the stream: xg holds x * 2^s, and a power-of-two
                // scaling is exact in both directions
                if (LEAN) tk[k] = xg[p] * inv;
            } else { Mn[k] = 0.0; Cn[k] = 0.0; if (LEAN) tk[k] = 0.0; }
        }
        ++n_pass;
        TMARK(2);
    };

    // ---- the iteration as a state machine around ONE SpMV call site ------------------------------------
    enum { PH_RESCALE, PH_INIT, PH_TRIAL, PH_SPLIT };
    int phase = p_rescale ? PH_RESCALE : PH_INIT;
    int mp1T = 0;
    int i = 0, j = 0, kk = 0;

    auto normalize_u = [&]() {                                  // u /= |u|
        double r[1] = {0.0}, m0[1] = {0.0};
        FOR_K(k, p) r[0] += u[k] * u[k];
        block_red<NW, 1, 0>(r, m0, red, par, tid);
        const double nr = sqrt(r[0]);
        if (nr > 0.0) FOR_K(k, p) u[k] /= nr;                   // (three passes per problem: the exact division stays)
    };
    auto load_u_as_x = [&]() {                                  // x = u (initial passes): tk, its maximum and support bound
        double r0[1] = {0.0}, m2[2] = {0.0, 0.0};
        FOR_K(k, p) { tk[k] = u[k]; m2[0] = fmax(m2[0], u[k]); if (u[k] > 0.0) m2[1] = (double)(p + 1); }
        block_red<NW, 0, 2>(r0, m2, red, par, tid);             // NS == 0: r0 is not touched
        xmaxT = m2[0]; mp1T = (int)m2[1];
    };
    // gradient of the current u: ((s_p + d) u_p - d sum(u)) + ((M + d C) u)_p  (oracle: grad_and_F_fused)
    // trial vector u' = normalize(max(u + alpha g, 0)) into tk (+ its sums and support)
    // second half of a trial vector: tk holds max(u + alpha g, 0), r2 is its squared norm, m2 its maximum and 1 + its last
    // non-zero position -> tk normalised, its sums, its distance from u, its maximum and support bound
    auto finish_trial = [&](double r2, const double (&m2)[2]) {
        const double nr = sqrt(r2);
        double q[2] = {0.0, 0.0}, m0[1] = {0.0};                // sum u', |u'-u|^2
        FOR_K(k, p) {
            double t = tk[k];
            // (0 / nr == 0: an element outside the support skips the division — once the support has collapsed that is every
            //  element of seven of the eight waves, and an f64 division is ~40 instructions)
            if (nr > 0.0 && t != 0.0) { t /= nr; tk[k] = t; }
            q[0] += t;
            const double df = t - u[k]; q[1] += df * df;
        }
        TMARK(3);
        // sum u' and |u' - u|^2 are not looked at before the products of u' are back: their reduction rides on the barrier that
        // publishes u' (spmv collects it behind that barrier) instead of having one of its own — four workgroup barriers per
        // pass instead of five
        if constexpr (NW == 1) { block_red<NW, 2, 0>(q, m0, red, par, tid); unsum = q[0]; du2 = q[1]; }
        else { pendRR = block_post<NW, 2>(q, red, par, tid); pend = true; }
        TMARK(7);
        xmaxT = (nr > 0.0) ? m2[0] / nr : m2[0]; mp1T = (int)m2[1];
    };
    auto build_trial = [&]() {
        double r[1] = {0.0}, m2[2] = {0.0, 0.0};
        FOR_K(k, p) {
            const double up = u[k];
            const double g = ((sd[k] + d) * up - d * usum) + Wu[k];
            double t = up + alpha * g;
            t = (p < L && t > 0.0) ? t : 0.0;
            tk[k] = t; r[0] += t * t;
            m2[0] = fmax(m2[0], t);
            if (t > 0.0) m2[1] = (double)(p + 1);               // ascending p: the last one stays
        }
        TMARK(3);
        block_red<NW, 1, 2>(r, m2, red, par, tid);
        TMARK(6);
        finish_trial(r[0], m2);
    };
    auto objective = [&](const double (&uu)[KMAX], const double (&ww)[KMAX], double us) -> double {
        double r[1] = {0.0}, m0[1] = {0.0};
        FOR_K(k, p) {
            const double up = uu[k];
            const double g = ((sd[k] + d) * up - d * us) + ww[k];
            r[0] += up * g;
        }
        block_red<NW, 1, 0>(r, m0, red, par, tid);
        return r[0];
    };
    // mean of (M u)_p / Cbu_p over the active set, from the products (Mn, Cn) of the split pass over u just done
    auto d_ratio = [&](bool absval, double& acc, double& cnt) {
        double r2[2] = {0.0, 0.0}, m0[1] = {0.0};
        FOR_K(k, p) {
            const double up = u[k], Cbu = (usum - Cn[k]) - up;
            if (p < L && Cbu > p_eps && up > p_eps) { const double r_ = (Mn[k] + sd[k] * up) / Cbu; r2[0] += absval ? fabs(r_) : r_; r2[1] += 1.0; }
        }
        block_red<NW, 2, 0>(r2, m0, red, par, tid);
        acc = r2[0]; cnt = r2[1];
    };

    [[maybe_unused]] int passes0 = 0;                           // n_pass when this launch took the problem
    [[maybe_unused]] bool budget = false;
    if constexpr (BUDGET && NW != 1 && !COOONLY) budget = cont != nullptr && cont->cap > 0;
    bool resumed = false;
    if constexpr (BUDGET && NW != 1 && !COOONLY) {
        if (cont != nullptr && resumeSlot >= 0) {               // a suspended problem: the state it was spilled with, through LDS
            double* x1 = reinterpret_cast<double*>(accM); double* x2 = reinterpret_cast<double*>(accC);
            cont_load(cont, resumeSlot, L, xg, x1, x2, red);
            phase = (int)uni(red[1]); i = (int)uni(red[2]); j = (int)uni(red[3]); kk = (int)uni(red[4]); n_pass = (int)uni(red[5]);
            ls_trials = (int)uni(red[6]); inner_iters = (int)uni(red[7]); mp1T = (int)uni(red[8]);
            alpha = uni(red[9]); F = uni(red[10]); d = uni(red[11]); usum = uni(red[12]); unsum = uni(red[13]); du2 = uni(red[14]); xmaxT = uni(red[15]);
            FOR_K_ALL(k, p) {
                const bool in = p < L;
                u[k] = in ? xg[p] : 0.0; Wu[k] = in ? x1[p] : 0.0; tk[k] = in ? x2[p] : 0.0;
            }
            __syncthreads();
            for (int p = tid; p < L; p += NT) { xg[p] = 0.0; accM[p] = 0ull; accC[p] = 0ull; }     // clean again: the first pass publishes into them
            __syncthreads();
            passes0 = n_pass; resumed = true;
        }
    }
    if (!resumed) {
        if (phase == PH_INIT) normalize_u();
        load_u_as_x();
    }
    for (;;) {
        if constexpr (BUDGET && NW != 1 && !COOONLY) {
            if (budget && n_pass - passes0 >= cont->cap) {      // out of budget: suspend here, in front of a pass (workgroup-uniform)
                if (pend) { __syncthreads(); collect_pending(); }
                __syncthreads();
                if (tid == 0) sint[5] = atomicAdd(cont->counters, 1);
                __syncthreads();
                const int slot = uni(sint[5]);
                if (slot < cont->slots) {
                    double* x1 = reinterpret_cast<double*>(accM); double* x2 = reinterpret_cast<double*>(accC);
                    FOR_K(k, p) if (p < L) { xg[p] = u[k]; x1[p] = Wu[k]; x2[p] = tk[k]; }
                    if (tid == 0) {
                        red[0] = (double)b; red[1] = (double)phase; red[2] = (double)i; red[3] = (double)j; red[4] = (double)kk; red[5] = (double)n_pass;
                        red[6] = (double)ls_trials; red[7] = (double)inner_iters; red[8] = (double)mp1T; red[9] = alpha; red[10] = F; red[11] = d;
                        red[12] = usum; red[13] = unsum; red[14] = du2; red[15] = xmaxT;
                    }
                    __syncthreads();
                    cont_spill(cont, slot, b, L, xg, x1, x2, red);
                    return;
                }
                budget = false;                                 // no slot left: this problem runs on
            }
        }
        // RESCALE: M x alone (a fused pass with nothing added); INIT / SPLIT: both products; TRIAL: (M + d C) x
        spmv(xmaxT, mp1T, phase == PH_INIT || phase == PH_SPLIT, phase == PH_TRIAL ? d : 0.0);
        if (phase == PH_RESCALE) {                              // u = normalize(M u0 + diag u0)
            FOR_K(k, p) u[k] = (p < L) ? Mn[k] + sd[k] * u[k] : 0.0;
            normalize_u();
            load_u_as_x();
            phase = PH_INIT;
            continue;
        }
        if (phase == PH_INIT) {
            double r1[1] = {0.0}, m0[1] = {0.0};
            FOR_K(k, p) r1[0] += u[k];
            block_red<NW, 1, 0>(r1, m0, red, par, tid);
            usum = r1[0];
            double acc, cnt;
            d_ratio(false, acc, cnt);
            d = (cnt > 0.0) ? acc / cnt : 0.0;
            i = 0;
            if (i >= p_maxout) break;
        } else if (phase == PH_SPLIT) {                         // end of an inner loop: homotopy update of d from (M u, C u)
            double acc, cnt;
            d_ratio(true, acc, cnt);
            if (cnt > 0.0) d += acc / cnt; else break;
            ++i;
            if (i >= p_maxout) break;
        } else {                                                // PH_TRIAL: the fused product of the trial vector
            ++ls_trials;
            // (Forming the first half of the NEXT trial — max(u' + g(u'), 0), its norm, maximum and support bound — in the same
            //  reduction as this objective saves a workgroup barrier per accepted trial and was built in round 5: the four-value
            //  reduction and the discarded work of rejected trials cost more than the barrier, 0.99 against 0.96 ms per launch.)
            const double Fnew = objective(tk, Mn, unsum);
            const double deltaF = Fnew - F;
            if (deltaF < -p_eps && kk + 1 < p_maxls) {          // backtrack
                alpha *= p_beta; ++kk;
                build_trial();
                continue;
            }
            // accept: the trial vector and its product become the current ones
            const double du = sqrt(du2);
            F = Fnew; usum = unsum;
            FOR_K(k, p) { u[k] = tk[k]; Wu[k] = Mn[k]; }
            ++inner_iters; ++j;
            const bool stop = du < p_tol_u || fabs(deltaF) < p_tol_F;
            if (stop || j >= p_maxin) {                         // the d update needs M u and C u apart: one split pass over
                phase = PH_SPLIT;                               // the accepted vector (tk, xmaxT, mp1T are still its own)
                continue;
            }
        }
        if (phase != PH_TRIAL) {                                // a new outer iteration (INIT / SPLIT): the fused product at the new d
            FOR_K(k, p) Wu[k] = Mn[k] + Cn[k] * d;
            F = objective(u, Wu, usum); j = 0;
        }
        alpha = 1.0; kk = 0;
        build_trial();
        phase = PH_TRIAL;
    }
    if (i >= p_maxout) status |= ROMAN_ST_MAXITER;
    S.n_pass = n_pass; S.ls_trials = ls_trials; S.inner_iters = inner_iters;
    S.outer_iters = i; S.score = F; S.d_final = d;
#ifdef ROMAN_SOLVE_TIMING
    TMARK(3);
#endif
    // final u (unscaled) to LDS for the shared tail; scratch: the two accumulator arrays
    __syncthreads();
    FOR_K(k, p) if (p < L) xg[p] = u[k];
#ifdef ROMAN_SMALL_TIMING
    const unsigned long long tt0_ = __builtin_readcyclecounter();
    const unsigned long long tw1_ = wall_clock64();              // (100 MHz, the same on every compute unit: when did this problem's iteration end?)
#endif
    finish_one(D, b, pd, feats, assoc, plp, lpAsc, rowPosPool, nullptr, O, xg, reinterpret_cast<double*>(accM),
               reinterpret_cast<int32_t*>(accC), reinterpret_cast<int32_t*>(accC) + Lc, L, rb, lo, F, status, S, red, sint);
#ifdef ROMAN_SOLVE_TIMING
    if (tid == 0 && O.dbg) {                                    // slot 5: everything outside the phases (set-up in front of the first pass, the tail: selection, pose, outputs)
        unsigned long long sum_ = 0; for (int t = 0; t < 8; ++t) sum_ += tacc[t];
        tacc[5] = __builtin_readcyclecounter() - tentry_ - sum_; tcnt[5] = 1;
        unsigned long long* dg = O.dbg + (size_t)b * 16;
        for (int t = 0; t < 8; ++t) { dg[t] = tacc[t]; dg[8 + t] = tcnt[t]; }
    }
#endif
#ifdef ROMAN_SMALL_TIMING
    if (tid == 0 && ((b & 255) == 0 || b < 2 || (NW == 1 && n_pass >= 100)))
        printf("[solve_up<%d>] b=%d L=%d nnz=%llu passes %d tail (selection, pose, outputs) %llu cycles; wall clock (10 ns): entered %llu iteration done %llu left %llu\n", NW, b, L,
               (unsigned long long)st[b].nnzUpper, n_pass, __builtin_readcyclecounter() - tt0_, tw0_, tw1_, wall_clock64());
#endif
#undef FOR_K
#undef FOR_K_ALL
#undef CUMQ
}

// MAXL < STREAM_MAXL with NW = 1: the SMALL-problem instantiation — one wave per problem (no cross-wave barriers or
// reductions at all), 64-thread workgroups, many of them per compute unit: submaps of the reference's demo scale (20-40
// objects, ~60 live associations) would otherwise occupy a whole 8-wave workgroup — a whole compute unit, given the
// registers of the general instantiation — for ~100 entries of matrix.  [Llo, Lhi]: the live-set sizes this launch takes.
template <int NW, bool HASCZ, int MAXL, int DEPTH = ST_D, bool LEAN = false, bool BUDGET = false>
__global__ void __launch_bounds__(NW * 64, LEAN ? 4 : (NW == 1 ? 3 : 1)) k_solve_up(DevParams D, int B, const ProbDesc* __restrict__ probs,
                                                      ProbState* __restrict__ st,
                                                      const double* __restrict__ feats, const int32_t* __restrict__ assoc,
                                                      const int32_t* __restrict__ plp, const int32_t* __restrict__ lpAsc,
                                                      const uint32_t* __restrict__ rowPos, const double* __restrict__ pld,
                                                      const uint32_t* __restrict__ sliceBase,
                                                      const uint16_t* __restrict__ cols, const double* __restrict__ vals,
                                                      const double* __restrict__ u0, SolveOut O,
                                                      int* __restrict__ queue, int Lc, int Llo, int Lhi, int R /* problems per claim: 1..64 */,
                                                      SolveCont cont /* pass budget / the suspended problems of the launch in front (NW == 1: unused) */)
{
    // LDS: xg[Lc] f64 | accM[Lc] u64 | accC[Lc] u64 | red[red_doubles(NW)] | cumQ[ST_MAXSL + 2] u32 | sint[8]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (Lc == MAXL + 64: solve_up addresses the accumulators at fixed distances from xg)
    double* xg = reinterpret_cast<double*>(smem);
    unsigned long long* accM = reinterpret_cast<unsigned long long*>(xg + Lc);
    unsigned long long* accC = accM + Lc;
    double* red = reinterpret_cast<double*>(accC + Lc);
    uint32_t* cumQ = reinterpret_cast<uint32_t*>(red + red_doubles(NW));
    int* sint = reinterpret_cast<int*>(cumQ + ST_MAXSL + 2);
    unsigned char* cooLds = reinterpret_cast<unsigned char*>(sint + 8);        // (one-wave instantiation only: COO_CAP * 12 bytes)
    if constexpr (BUDGET && NW != 1) {
        if (cont.resume) {                                      // the suspended problems of the launch in front, one workgroup each
            for (;;) {
                if (threadIdx.x == 0) sint[2] = atomicAdd(cont.counters + 1, 1);
                __syncthreads();
                const int slot = uni(sint[2]);
                __syncthreads();
                if (slot >= min(*cont.counters, cont.slots)) break;
                const int b = cont.list[slot];
                const ProbDesc pd = probs[b];
                solve_up<NW, HASCZ, MAXL, DEPTH, LEAN, false, true>(D, b, pd, st, feats, assoc, plp, lpAsc, rowPos, pld, sliceBase, cols, vals, u0, O,
                                          xg, accM, accC, Lc, cumQ, red, sint, cooLds, -1, -1, &cont, slot);
                __syncthreads();
            }
            return;
        }
    }
    for (;;) {
        // The first wave claims R consecutive problems at a time (R = 1 unless this launch expects to find nothing: a batch
        // of small problems passes through the general instantiation and vice versa) until the range holds one this launch
        // takes (stream layout, live-set size in [Llo, Lhi]): the others — the fallback solver's, skipped ones, the other
        // instantiation's — cost one atomic and two loads per range, no barrier.
        if (threadIdx.x < WAVE) {
            const int ln = threadIdx.x;
            int base_ = B; unsigned long long m_ = 0ull;
            for (;;) {
                int t_ = 0;
                if (ln == 0) t_ = atomicAdd(queue, R);
                base_ = __builtin_amdgcn_readfirstlane(t_);
                if (base_ >= B) break;
                const int b_ = base_ + ln;
                bool take = false;
                if (ln < R && b_ < B) { const int Lq = st[b_].L; take = st[b_].kind == 0 && Lq >= Llo && Lq <= Lhi; }
                m_ = __ballot(take);
                if (m_) break;
            }
            if (ln == 0) { sint[2] = base_; sint[3] = (int)(uint32_t)m_; sint[4] = (int)(uint32_t)(m_ >> 32); }
        }
        __syncthreads();
        const int base = uni(sint[2]);
        unsigned long long mask = ((unsigned long long)(uint32_t)uni(sint[4]) << 32) | (unsigned long long)(uint32_t)uni(sint[3]);
        __syncthreads();
        if (base >= B) break;
        while (mask) {
        const int b = base + __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const ProbDesc pd = probs[b];
        solve_up<NW, HASCZ, MAXL, DEPTH, LEAN, false, BUDGET>(D, b, pd, st, feats, assoc, plp, lpAsc, rowPos, pld, sliceBase, cols, vals, u0, O,
                                  xg, accM, accC, Lc, cumQ, red, sint, cooLds, -1, -1, (BUDGET && NW != 1) ? &cont : nullptr, -1);
        __syncthreads();                                         // the next problem of the range reuses the LDS state
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_small: a problem of the reference's DEMO scale (submaps of 20-40 objects: ~60 live associations, ~110 stored pairs)
// from the live list to the pose in ONE kernel, one wave per problem, nothing but the result written to memory.  The general
// path spends eleven launches on such a problem (pair tests, mirror, prefix, sort, lists, geometry, scans, fill, solve ...):
// at 4096 problems per call every one of them costs 5-130 us for a few hundred bytes of work each.  Here lane = live row
// (two rows per lane: L <= SMALL_MAXL = 128):
//   pair tests   row k against every live column q, sequentially over q (both triangles: no transposition), the same
//                gate on the same table entries as k_count -> the row's candidate bits in registers;
//   positions    rank by (degree descending, row ascending): one sweep over the degrees in LDS — the order k_rowsort's
//                bitonic sort produces;
//   kept pairs   a candidate pair belongs to its endpoint of smaller position: listed in LDS, then evaluated 64 at a time
//                with k_fill_list's own value sequence (fill_value) and filter;
//   solve        the kept entries ARE the coordinate list the one-wave solver keeps in registers (solve_up, cooPre).
// Same candidates, same positions, same values, exact sums: the same bits as the general path.  A problem with more than
// SMALL_PAIR_CAP candidates or COO_CAP stored pairs is left to the general path (kind stays 0); a finished one becomes kind 3.
// ---------------------------------------------------------------------------------------------
constexpr int SMALL_PAIR_CAP = 2304;      // candidate pairs the LDS of the solver's three vectors can list (16 bits each)
// dynamic LDS of a k_small workgroup (the host launches with this): solver vectors | reduction scratch | slice table | sint | (16-byte
// alignment) | coordinate list | column data z / single score / objects / positions / degrees of up to SMALL_MAXL live associations
constexpr size_t small_lds_bytes()
{
    size_t o = (size_t)3 * 8 * (SMALL_MAXL + 64) + sizeof(double) * red_doubles(1) + sizeof(uint32_t) * (ST_MAXSL + 2) + sizeof(int) * 8;
    o = (o + 15) & ~(size_t)15;
    return o + (size_t)12 * COO_CAP + (size_t)SMALL_MAXL * (16 + 8 + 4 + 2 + 2);
}

__device__ __forceinline__ bool pair_gate_rt(const DevParams& D, int gm, double a, double bb, double dz)
{
    switch (gm) {
    case 1: return pair_gate<1>(D, a, bb, dz);
    case 2: return pair_gate<2>(D, a, bb, dz);
    case 3: return pair_gate<3>(D, a, bb, dz);
    default: return pair_gate<0>(D, a, bb, dz);
    }
}

template <bool FAST>
__global__ void __launch_bounds__(64, 3) k_small(DevParams D, int B, const ProbDesc* __restrict__ probs, ProbState* __restrict__ st,
                                              const double* __restrict__ feats, const int32_t* __restrict__ assoc,
                                              const double* __restrict__ tabPool,
                                              const int32_t* __restrict__ lp, const int32_t* __restrict__ li, const int32_t* __restrict__ lj,
                                              const double* __restrict__ ls, const double* __restrict__ ld,
                                              const double* __restrict__ lza, const double* __restrict__ lzb,
                                              int32_t* __restrict__ plp, double* __restrict__ pld, uint32_t* __restrict__ rowPos,
                                              const double* __restrict__ u0, SolveOut O, int* __restrict__ queue)
{
    // LDS: xg | accM | accC [Lc1 each] (before the solve: the candidate pair list) | red | cumQ | sint | coordinate list | cIJ | cZ | cS | posS | degS
    constexpr int Lc1 = SMALL_MAXL + 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* xg = reinterpret_cast<double*>(smem);
    unsigned long long* accM = reinterpret_cast<unsigned long long*>(xg + Lc1);
    unsigned long long* accC = accM + Lc1;
    double* red = reinterpret_cast<double*>(accC + Lc1);
    uint32_t* cumQ = reinterpret_cast<uint32_t*>(red + red_doubles(1));
    int* sint = reinterpret_cast<int*>(cumQ + ST_MAXSL + 2);
    unsigned char* cooLds = smem + (((size_t)(reinterpret_cast<unsigned char*>(sint + 8) - smem) + 15) & ~(size_t)15);   // 16-byte aligned (cZ behind it holds double2)
    uint32_t* lpq = reinterpret_cast<uint32_t*>(cooLds); double* lv = reinterpret_cast<double*>(cooLds + 4 * COO_CAP);
    double2* cZ = reinterpret_cast<double2*>(cooLds + 12 * COO_CAP);
    static_assert((12 * COO_CAP) % 16 == 0, "cZ alignment");
    double* cS = reinterpret_cast<double*>(cZ + SMALL_MAXL);
    uint32_t* cIJ = reinterpret_cast<uint32_t*>(cS + SMALL_MAXL);
    uint16_t* posS = reinterpret_cast<uint16_t*>(cIJ + SMALL_MAXL);
    uint16_t* degS = posS + SMALL_MAXL;
    uint16_t* pairs = reinterpret_cast<uint16_t*>(smem);       // (the solver's vectors are not in use yet)
    // (layout above == small_lds_bytes(): degS ends at cooLds + 12 COO_CAP + SMALL_MAXL (16 + 8 + 4 + 2 + 2))
    static_assert(SMALL_PAIR_CAP * 2 <= 3 * 8 * Lc1, "pair list fits the solver's vectors");
    const int lane = threadIdx.x;
    const int gm = D.gmode;
    for (;;) {
        {                                                       // one problem per claim: thousands of short problems balance over ~3000 resident waves
            int t_ = 0;
            if (lane == 0) t_ = atomicAdd(queue, 1);
            const int b = __builtin_amdgcn_readfirstlane(t_);
            if (b >= B) break;
            {
                const int Lq = uni_i(st[b].L);
                if (!(uni_i(st[b].kind) == 0 && Lq >= 1 && Lq <= SMALL_MAXL)) continue;
            }
            const ProbDesc pd = probs[b];
            const int L = uni_i(st[b].L);
            const int64_t lo = pd.liveOff;
            const double* TA = tabPool + pd.tabOff;
            const double* TB = TA + (int64_t)pd.n1 * pd.n1;
            __syncthreads();                                    // the previous problem is done with the LDS
#ifdef ROMAN_SMALL_TIMING
            unsigned long long ts_[8]; ts_[0] = __builtin_readcyclecounter();
#define SMARK(i_) ts_[i_] = __builtin_readcyclecounter()
#else
#define SMARK(i_) do { } while (0)
#endif
            // ---- rows ------------------------------------------------------------------------------------
            int ri[2], rj[2], rlp[2]; double rd[2], rza[2], rzb[2]; bool rv[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = r * 64 + lane;
                rv[r] = k < L;
                const int kk = rv[r] ? k : 0;
                ri[r] = li[lo + kk]; rj[r] = lj[lo + kk]; rlp[r] = lp[lo + kk]; rd[r] = ld[lo + kk];
                rza[r] = lza[lo + kk]; rzb[r] = lzb[lo + kk];
                if (rv[r]) { cIJ[k] = (uint32_t)ri[r] | ((uint32_t)rj[r] << 16); cZ[k] = make_double2(rza[r], rzb[r]); cS[k] = ls[lo + kk]; }
            }
            __syncthreads();
            SMARK(1);
            // ---- pair tests: row k against every live column ---------------------------------------------------
            unsigned long long m[2][2] = {{0ull, 0ull}, {0ull, 0ull}};
            const double* rowA[2] = {TA + (int64_t)ri[0] * pd.n1, TA + (int64_t)ri[1] * pd.n1};
            const double* rowB[2] = {TB + (int64_t)rj[0] * pd.n2, TB + (int64_t)rj[1] * pd.n2};
            const int nr = L > 64 ? 2 : 1;                      // row halves in use
#pragma unroll
            for (int wd = 0; wd < 2; ++wd) {
                const int q1 = min(L, (wd + 1) * 64);
                for (int q0 = wd * 64; q0 < q1; q0 += 4) {       // four columns at a time: their table gathers (global memory) in flight together
                    uint32_t pk[4]; double2 zz[4]; double a[2][4], bb[2][4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) { const int q = min(q0 + x, q1 - 1); pk[x] = cIJ[q]; zz[x] = cZ[q]; }
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
                            if (r < nr) { a[r][x] = rowA[r][pk[x] & 0xffffu]; bb[r][x] = rowB[r][pk[x] >> 16]; }
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        if (q0 + x < q1) {
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                if (r < nr) {
                                    const double dz = gm ? fabs((rza[r] - zz[x].x) - (rzb[r] - zz[x].y)) : 0.0;
                                    const bool is = pair_gate_rt(D, gm, a[r][x], bb[r][x], dz);
                                    m[r][wd] |= (unsigned long long)(is ? 1u : 0u) << ((q0 + x) & 63);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) if (!rv[r]) { m[r][0] = 0ull; m[r][1] = 0ull; }
            SMARK(2);
            // ---- positions: rank by (degree descending, row ascending) -----------------------------------------
            int dg[2], pos[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) { dg[r] = __popcll(m[r][0]) + __popcll(m[r][1]); if (rv[r]) degS[r * 64 + lane] = (uint16_t)dg[r]; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int rank = 0;
                const int k = r * 64 + lane;
                if (r < nr) for (int k2 = 0; k2 < L; ++k2) { const int d2 = (int)degS[k2]; rank += (d2 > dg[r] || (d2 == dg[r] && k2 < k)) ? 1 : 0; }
                pos[r] = rank;
                if (rv[r]) { posS[k] = (uint16_t)rank; rowPos[lo + k] = (uint32_t)rank; plp[lo + rank] = rlp[r]; pld[lo + rank] = rd[r]; }
            }
            __syncthreads();
            SMARK(3);
            // ---- the candidates a row KEEPS: those whose position is larger than its own ---------------------------
            uint32_t cnt[2] = {0u, 0u};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int wd = 0; wd < 2; ++wd) {
                    unsigned long long x = m[r][wd], keep = 0ull;
                    while (x) {
                        const int bit = __builtin_ctzll(x);
                        x &= x - 1ull;
                        if ((int)posS[wd * 64 + bit] > pos[r]) { keep |= 1ull << bit; ++cnt[r]; }
                    }
                    m[r][wd] = keep;
                }
            const uint32_t inc0 = wave_incl_scan(cnt[0]), tot0 = (uint32_t)__builtin_amdgcn_readlane((int)inc0, 63);
            const uint32_t inc1 = wave_incl_scan(cnt[1]), tot1 = (uint32_t)__builtin_amdgcn_readlane((int)inc1, 63);
            const uint32_t cand = tot0 + tot1;
            if (cand > (uint32_t)SMALL_PAIR_CAP) continue;       // too many candidates for the list: the general path takes the problem
            uint32_t off[2] = {inc0 - cnt[0], tot0 + inc1 - cnt[1]};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int wd = 0; wd < 2; ++wd) {
                    unsigned long long x = m[r][wd];
                    while (x) {
                        const int bit = __builtin_ctzll(x);
                        x &= x - 1ull;
                        pairs[off[r]++] = (uint16_t)((r * 64 + lane) | ((wd * 64 + bit) << 8));
                    }
                }
            __syncthreads();
            SMARK(4);
            // ---- values of the kept candidates, 64 at a time; entries above affinityeps form the coordinate list -------------
            uint32_t nk = 0;
            for (uint32_t c0 = 0; c0 < cand; c0 += 64u) {
                const uint32_t idx = c0 + (uint32_t)lane;
                const bool have = idx < cand;
                const uint32_t pr = pairs[have ? idx : 0u];
                const int k = (int)(pr & 0xffu), q = (int)(pr >> 8);
                const uint32_t pkk = cIJ[k], pkq = cIJ[q];
                const double a = TA[(int64_t)(pkk & 0xffffu) * pd.n1 + (pkq & 0xffffu)], bb = TB[(int64_t)(pkk >> 16) * pd.n2 + (pkq >> 16)];
                const double2 zk = cZ[k], zq = cZ[q];
                const double sk = cS[k], sq = cS[q];
                double v;
                if (D.gravity) v = fill_value<true, FAST>(D, a, bb, zk.x - zq.x, zk.y - zq.y, sk * sq, sk, sq);
                else v = fill_value<false, FAST>(D, a, bb, 0.0, 0.0, sk * sq, sk, sq);
                const bool keep = have && v > D.p.affinityeps;
                const unsigned long long km = __ballot(keep);
                if (keep) {
                    const uint32_t at = nk + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                    if (at < (uint32_t)COO_CAP) { lpq[at] = (uint32_t)posS[k] | ((uint32_t)posS[q] << 8); lv[at] = v; }
                }
                nk += (uint32_t)__popcll(km);
            }
            if (nk > (uint32_t)COO_CAP) continue;                // more stored pairs than the registers hold: the general path
            if (lane == 0) st[b].nnzUpper = (unsigned long long)nk;
            __syncthreads();                                    // the pair list is spent: its LDS becomes the solver's vectors
            SMARK(5);
            solve_up<1, false, SMALL_MAXL, ST_D, false, true>(D, b, pd, st, feats, assoc, plp, lp, rowPos, pld, nullptr, nullptr, nullptr, u0, O,
                                           xg, accM, accC, Lc1, cumQ, red, sint, cooLds, (int)nk, (int)lo);
            if (lane == 0) st[b].kind = 3;                      // done: the general kernels pass it by
#ifdef ROMAN_SMALL_TIMING
            SMARK(6);
            if (lane == 0 && ((b & 255) == 0 || B <= 4)) printf("[k_small] b=%d L=%d cand=%u nnz=%u cycles: rows %llu tests %llu ranks %llu keep %llu values %llu solve+tail %llu\n", b, L, cand, nk,
                                                                ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3], ts_[5] - ts_[4], ts_[6] - ts_[5]);
#endif
#undef SMARK
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_solve_wide: the solver for FEW, LARGE problems of the fallback kind (symmetric sorted SELL-64): every compute unit
// works on ONE problem at a time.  (`method='gravity'` / `'clipper'` at n = m = 200, or the ROMAN_SINGLE_DIAG_KEEP
// reading: L = 40 000 live associations, 0.5 GB of matrix; one workgroup would stream that at a few GB/s.)
//
//  * One persistent launch, one 512-thread workgroup per compute unit (cooperative launch: all resident).
//  * The matrix is ONE flat stream of T "steps" (a step = one quad of the 64 lanes of a slice: 4 column words and 4
//    values per lane, three 16-byte loads), cut into chunks that are dealt round-robin to the waves of the grid (chunk c
//    to wave c mod NWG: at any moment the grid reads one contiguous window of the matrix; every wave takes the same
//    number of chunks) whatever slices they belong to; a wave keeps two blocks of WIDE_U quads in flight and leaves the partial row sums of every slice piece it
//    covered in a partials buffer (piece id = chunk + slice).  The rows' owners add the pieces of their slice in
//    ascending order — a fixed order, no atomics.  (The round-2 solver gave a whole slice to one wave: 625 waves of
//    4096 busy, each walking ~1000 dependent steps: 0.5 ms per pass of pure latency.)
//  * Vector elements live in REGISTERS: thread (wave g, lane l) owns positions (g + k NWG) * 64 + l — the rows of the slices
//    whose partials it adds.  Only the vector being multiplied is in memory (live order, gathered by column index).
//  * The trial vector is published UNNORMALISED: t = max(u + alpha grad, 0) goes out together with the partials of
//    sum t^2 and sum t under ONE grid barrier; M t, C t come back and are scaled by 1 / |t| in registers
//    (M (t / |t|) = (M t) / |t| up to rounding).  A line-search trial costs TWO grid barriers: behind the stream, and one
//    that carries the trial's objective together with the sums of BOTH vectors that can come next (the backtracked
//    trial and the first trial from the accepted vector are published speculatively before the objective is known).
//  * Grid barrier: one monotone counter; everything other workgroups read is stored write-through (`sc1`), so an
//    arrival is: every wave drains its stores, workgroup barrier, one relaxed atomic add; departure: relaxed poll,
//    ONE agent-scope acquire, workgroup barrier (MI355X_MICROARCH.md "barrier-counter": 7 us against 26 us for
//    cooperative_groups' grid.sync()).  Grid sums ride on it: workgroup partials into a ping-pong slot array before the
//    arrival, every workgroup adds the slots in one fixed order afterwards.  Every spin is bounded (4 s): on a timeout the
//    kernel gives up and reports ROMAN_ST_INTERNAL instead of hanging the device.
//  * Column compaction (round 4): the team keeps books on the published support bit maps and, when the support has left half
//    of the columns in use, rewrites the matrix without them into a mirror of the pools and streams that copy (see `stream`).
//  Mirrors oracle_solve() like solve_one; the sums are plain doubles in a fixed (different) order.
// ---------------------------------------------------------------------------------------------
constexpr int WIDE_NT = 512;             // threads per workgroup (one workgroup per compute unit; 8 waves: 256 registers each)
constexpr int WIDE_NW = WIDE_NT / 64;
constexpr int WIDE_KW = 2;               // vector elements a thread can own: L <= WIDE_KW * 64 * (waves of the grid)
constexpr int WIDE_NRED = 10;            // doubles per workgroup slot of a grid reduction
constexpr int WIDE_MAXBLK = 8;           // column blocks of the half copy (pull + push passes) at most
#ifndef ROMAN_WIDE_U
#define ROMAN_WIDE_U 3
#endif
constexpr int WIDE_U = ROMAN_WIDE_U;     // quads (of 4 entries per lane) per block of the stream; two blocks in flight per wave
constexpr int WIDE_MAXCH = 8;            // chunks of the flat stream a wave takes per pass at most

// dynamic LDS of k_solve_wide in front of the gathered vector's leading part: three bit maps of bmWords 64-bit words and the
// cumulative slice widths (bmWords + 2 32-bit words), rounded up to 16 bytes (host and kernel use this one function)
__host__ __device__ constexpr size_t wide_fixed_lds(int bmWords) { return ((size_t)3 * 8 * (size_t)bmWords + 4 * ((size_t)bmWords + 2) + 15) & ~(size_t)15; }

struct WideShared {
    double wred[2][WIDE_NW][WIDE_NRED];  // wave partials of a reduction (ping-pong)
    double bc[2][WIDE_NRED];             // reduced values for the whole workgroup (ping-pong)
    double red[72];                      // finish_one scratch
    int sint[4];
    int abort_;
    int sS[WIDE_NT / 64][8];             // per wave: first slice of each of its chunks (-1: no such chunk)
    // the half copy (pull + push passes): per column block its stream's steps, steps per chunk, first piece id, first step in the
    // mirror pools, the team's workgroups that stream it (count, first rank)
    uint32_t upT[WIDE_MAXBLK], upCS[WIDE_MAXBLK], upPB[WIDE_MAXBLK], upSB[WIDE_MAXBLK];
    int upG[WIDE_MAXBLK], upCU[WIDE_MAXBLK];
    int upI[4];                          // column blocks, columns per block, my block, steps of the copy (read where needed: not kept in registers)
    double upFx;                         // 2^-s of the pass on the copy
};

// write-through store of a value other workgroups will read (global_store ... sc1: no release fence needed later)
__device__ __forceinline__ void st_pub(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barrier, XCD-hierarchical (MI355X_MICROARCH.md "barrier-xcd": 4-5 us against 7+ for one flat counter and 26 for
// cooperative_groups' grid.sync()): a workgroup arrives at the counter of the XCD it runs on; the last arriver of an XCD
// arrives at the top counter, waits for the other XCDs' leaders and then releases its own XCD through a generation word that
// the others poll — 32 + 8 serialised atomics instead of 256, and nobody polls a line that arrivals are still hitting.
// Which XCD a workgroup is on is read from the hardware (HW_REG_XCC_ID), how many workgroups each XCD holds is counted at
// kernel start (wide_census): nothing is assumed about placement.  Words of `bar` (zeroed before every launch), one 128-byte
// line each: [0] abort flag, [1+x] arrivals of XCD x, [9] top arrivals, [10+x] generation of XCD x, [18] census (8 words),
// [19] flat counter of the census barrier.  RELEASE: plain stores issued before the barrier must be visible behind it as well.
// Every spin is bounded (4 s): on a timeout the abort flag goes up and every workgroup leaves.
// TEAM mode (several fallback problems in a batch): the workgroups of an XCD — or of a half / a quarter of it (`sub` teams per
// XCD) — form a team that solves its own problem; a team barrier is ONE level (the team's counter, released by the last
// arriver through the team's generation word) and the teams never meet after the census.  Further lines of `bar`: [20] the
// problem queue, [21] number of fallback problems (written by k_skipped), [24 + t] arrivals of team t, [56 + t] generation
// of team t (t < 32).
constexpr int WIDE_MAX_TEAMS = 32;
constexpr int WIDE_BAR_WORDS = (56 + WIDE_MAX_TEAMS) * 32;
struct WideBar { unsigned* bar; unsigned epoch; int G; int xcc; unsigned nX; unsigned nActive; unsigned long long budget /* wall-clock ticks a spin may last */;
                 int teamMode; int tG /* workgroups of my team */; int tRank /* my rank in it */; int team; };

__device__ __forceinline__ bool wide_spin(const unsigned* word, unsigned target, unsigned* bar, unsigned long long budget)
{
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        // (budget 0 — the test hook ROMAN_WIDE_SPIN_MS=0 — makes the very first unsuccessful poll a timeout)
        if (((++n & 255u) == 0u || budget == 0ull) && (wall_clock64() - t0 > budget || __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

template <bool RELEASE>
__device__ __forceinline__ bool wide_sync(WideShared& sh, WideBar& wb, int ltid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave: its stores have left the compute unit
    __syncthreads();
    ++wb.epoch;
    if (ltid == 0) {
        if (RELEASE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        unsigned* bar = wb.bar;
        bool ok = true;
        if (wb.teamMode) {                                      // one level: the team's counter, the last arriver releases the team
            const unsigned old = __hip_atomic_fetch_add(bar + 32 * (24 + wb.team), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == wb.epoch * (unsigned)wb.tG) __hip_atomic_store(bar + 32 * (56 + wb.team), wb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ok = wide_spin(bar + 32 * (56 + wb.team), wb.epoch, bar, wb.budget);
            if (!ok) sh.abort_ = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
        const unsigned old = __hip_atomic_fetch_add(bar + 32 * (1 + wb.xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == wb.epoch * wb.nX) {                     // the last workgroup of this XCD: on to the top level
            __hip_atomic_fetch_add(bar + 32 * 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = wide_spin(bar + 32 * 9, wb.epoch * wb.nActive, bar, wb.budget);
            // (a leader that gave up does NOT release its XCD: the abort flag is up, its peers see it within 256 polls and
            //  leave from this barrier instead of walking into the next one)
            if (ok) __hip_atomic_store(bar + 32 * (10 + wb.xcc), wb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = wide_spin(bar + 32 * (10 + wb.xcc), wb.epoch, bar, wb.budget);
        }
        if (!ok) sh.abort_ = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // this compute unit's L1 holds nothing older than the barrier
        }
    }
    __syncthreads();
    return sh.abort_ == 0;
}

// Once per launch: which XCD am I on, how many workgroups does each XCD hold (one flat-counter barrier).
__device__ __forceinline__ bool wide_census(WideShared& sh, WideBar& wb, int ltid)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    wb.xcc = (int)(x & 7u);
    if (ltid == 0) {
        unsigned* bar = wb.bar;
        sh.sint[2] = (int)__hip_atomic_fetch_add(bar + 32 * 18 + wb.xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // my rank among the XCD's workgroups
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar + 32 * 19, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = wide_spin(bar + 32 * 19, (unsigned)wb.G, bar, wb.budget);
        unsigned act = 0;
        for (int t = 0; t < 8; ++t) {
            const unsigned c_ = __hip_atomic_load(bar + 32 * 18 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            act += c_ ? 1u : 0u;
            if (t == wb.xcc) sh.sint[0] = (int)c_;
        }
        sh.sint[1] = (int)act;
        if (!ok) sh.abort_ = 1;
    }
    __syncthreads();
    wb.nX = (unsigned)sh.sint[0]; wb.nActive = (unsigned)sh.sint[1];
    const int rankX = sh.sint[2];
    __syncthreads();
    if (wb.teamMode) {                                          // `teamMode` sub-teams per XCD: ranks [first(s), first(s + 1)), first(s) = ceil(s nX / sub)
        const int sub = wb.teamMode, nX = (int)wb.nX;
        const int sidx = min(sub - 1, (rankX * sub) / max(nX, 1));
        const int f0 = (sidx * nX + sub - 1) / sub, f1 = ((sidx + 1) * nX + sub - 1) / sub;
        // (rank r belongs to sub-team floor(r sub / nX) — first(s) is the smallest r with r sub >= s nX)
        wb.tG = f1 - f0; wb.tRank = rankX - f0; wb.team = wb.xcc * sub + sidx;
    } else { wb.tG = wb.G; wb.tRank = (int)blockIdx.x; wb.team = 0; }
    return sh.abort_ == 0;
}

// Grid barrier carrying N values: v[i] <- sum (i < N - NM) or maximum (the last NM, non-negative values) over every thread
// of the grid, identical in all of them (fixed order: lanes by butterfly, waves ascending, workgroups lane-strided then
// butterfly).  N == 0: the barrier alone.
template <int N, int NM = 0>
__device__ __forceinline__ bool wide_reduce(double (&v)[N > 0 ? N : 1], WideShared& sh, double* slots, WideBar& wb, int ltid)
{
    static_assert(N <= WIDE_NRED && NM <= N, "slot width");
    const int lane = ltid & 63, w = ltid >> 6, par = (int)(wb.epoch & 1u);
    const int G = wb.tG;                                        // the team's workgroups (whole-device mode: the grid)
    slots += (size_t)wb.team * 2 * (size_t)wb.G * WIDE_NRED;    // a team's own ping-pong slot array
    if (N > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double x = v[i];
            if (i < N - NM) { for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off); }
            else { for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off)); }
            if (lane == 0) sh.wred[par][w][i] = x;
        }
        __syncthreads();
        if (ltid < N) {
            double t = 0.0;
            if (ltid < N - NM) { for (int ww = 0; ww < WIDE_NW; ++ww) t += sh.wred[par][ww][ltid]; }
            else { for (int ww = 0; ww < WIDE_NW; ++ww) t = fmax(t, sh.wred[par][ww][ltid]); }
            st_pub(slots + ((size_t)par * G + wb.tRank) * WIDE_NRED + ltid, t);
        }
    }
    if (!wide_sync<false>(sh, wb, ltid)) return false;
    if (N > 0) {
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double a = 0.0;
                if (i < N - NM) {
                    for (int g = lane; g < G; g += WAVE) a += slots[((size_t)par * G + g) * WIDE_NRED + i];
                    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
                } else {
                    for (int g = lane; g < G; g += WAVE) a = fmax(a, slots[((size_t)par * G + g) * WIDE_NRED + i]);
                    for (int off = 32; off > 0; off >>= 1) a = fmax(a, __shfl_xor(a, off));
                }
                if (lane == 0) sh.bc[par][i] = a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = sh.bc[par][i];
    }
    return true;
}

template <typename IdxT, bool HALF /* the instantiation that can run pull + push passes over a half copy of the matrix (build_upper; 16-bit labels) */>
__global__ void __launch_bounds__(WIDE_NT) k_solve_wide(DevParams D, int B, const ProbDesc* __restrict__ probs,
                                                        ProbState* __restrict__ st,
                                                        const double* __restrict__ feats, const int32_t* __restrict__ assoc,
                                                        const int32_t* __restrict__ lp, const double* __restrict__ ld,
                                                        const uint32_t* __restrict__ permPool, const uint32_t* __restrict__ rowPosPool,
                                                        const uint32_t* __restrict__ sliceBasePool,
                                                        const IdxT* __restrict__ colsPool, const double* __restrict__ valsPool,
                                                        double* __restrict__ vU /* final u by position */,
                                                        double* __restrict__ vXa, double* __restrict__ vXb /* published vectors, by position */,
                                                        double* __restrict__ vS0, double* __restrict__ vS1, double* __restrict__ vS2 /* scratch of the shared tail */,
                                                        int32_t* __restrict__ plp /* position -> association index (written here) */,
                                                        const double* __restrict__ u0, SolveOut O,
                                                        double* __restrict__ part /* [(chunks + slices)][64][2] */,
                                                        double* __restrict__ slots /* [2][G][WIDE_NRED] */, unsigned* __restrict__ bar,
                                                        unsigned long long* __restrict__ bmPool /* [2][bmWords]: support bit maps of the two published vectors */,
                                                        int bmWords /* 64-bit words of one bit map (and of its LDS copy) */,
                                                        int xcap /* doubles of dynamic LDS behind the bit map: the gathered vector's leading part */,
                                                        int tune /* experiments: bit 0 never gather from LDS, bit 1 non-temporal matrix loads, bits 8.. chunks per wave */,
                                                        unsigned long long spinTicks /* wall-clock ticks a barrier wait may last (host: 4 s at the device's wall-clock rate) */,
                                                        int teams /* 0: the whole device on one problem at a time; s >= 1: s teams per XCD, a problem each */,
                                                        const int32_t* __restrict__ fbList /* the batch's fallback problems (k_skipped) */,
                                                        long long partStride /* doubles of `part` a team owns */,
                                                        IdxT* colsC, double* valsC /* (no __restrict__: a compacted copy is compacted again in place) mirror of the matrix pools: the column-compacted copy (same offsets) */,
                                                        int ccfg /* column compaction: bits 0-7 compactions allowed per problem (0: off), 8-15 threshold (x / 256 of the columns in use), 16-23 passes per window */,
                                                        int ucfg /* bit 0: pull + push passes over a half copy of the matrix — every pair stored once — (16-bit labels only) while no compacted copy exists */,
                                                        unsigned long long* __restrict__ yPart /* [team][ySlots][2][ycap]: a workgroup's pushed fixed-point sums of its column block (M x, C x) */,
                                                        int ySlots /* workgroups of a team the host sized yPart for */, int ycap /* columns of a block at most (multiple of 64, 3 ycap <= xcap) */,
                                                        uint32_t* __restrict__ upMeta /* [team][WIDE_MAXBLK][bmWords + 1]: slice widths (steps) of the half copy per column block */)
{
    __shared__ WideShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_smem[];
    unsigned long long* bml = reinterpret_cast<unsigned long long*>(wide_smem);      // support bit map of the vector being multiplied
    unsigned long long* bmC = bml + bmWords;                                         // columns of the compacted copy in use
    unsigned long long* bmA = bmC + bmWords;                                         // union of the supports multiplied in the current window
    uint32_t* cw = reinterpret_cast<uint32_t*>(bmA + bmWords);                       // cumulative slice widths (steps) of the stream in use: bmWords + 1 (+ 1 pad)
    double* xl = reinterpret_cast<double*>(wide_smem + wide_fixed_lds(bmWords));     // the multiplied vector's leading xcap elements (16-byte aligned)
    const roman_params_t& P = D.p;
    const int ltid = threadIdx.x, lane = ltid & 63, w = uni_i(ltid >> 6);
    if (ltid == 0) { sh.abort_ = 0; sh.upI[3] = 0; }
    __syncthreads();
    WideBar wb{bar, 0u, (int)gridDim.x, 0, 1u, 1u, spinTicks, teams, (int)gridDim.x, (int)blockIdx.x, 0};
    // (a problem this launch does not finish keeps the record k_skipped pre-wrote for it: ROMAN_ST_INTERNAL, no associations,
    //  NaN pose — whatever exit a workgroup takes after a bounded wait expired, nothing stale is left behind)
    if (!wide_census(sh, wb, ltid)) return;
    const int G = wb.tG, NWG = G * WIDE_NW;                      // my team (whole-device mode: the grid)
    const int gw = w * G + wb.tRank;                            // wave id in the team: consecutive ids on different compute units
    const int gwc = (tune & 4) ? wb.tRank * WIDE_NW + w : gw;   // wave id for the deal of stream chunks (experiment: the waves of a compute unit take consecutive chunks)
    part += (size_t)wb.team * (size_t)partStride;
    bmPool += (size_t)wb.team * 3 * (size_t)bmWords;            // two bit maps + the new slice widths of a compaction
    uint32_t* wNew = reinterpret_cast<uint32_t*>(bmPool + 2 * (size_t)bmWords);
    const int cmax = ccfg & 0xff, cthr = (ccfg >> 8) & 0xff, cwin = max(1, (ccfg >> 16) & 0xff);
    const int nFb = (int)__hip_atomic_load(bar + 32 * 21, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int jq = 0; ; ++jq) {
        int b;
        if (teams) {                                            // the team's first workgroup claims the next problem; a barrier carries it to the others
            double claim[1] = {0.0};
            if (wb.tRank == 0) {
                if (ltid == 0) { const unsigned j_ = __hip_atomic_fetch_add(bar + 32 * 20, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh.sint[3] = j_ < (unsigned)nFb ? fbList[j_] + 1 : 0; }
                __syncthreads();
                claim[0] = (double)sh.sint[3];
                __syncthreads();
            }
            if (!wide_reduce<1, 1>(claim, sh, slots, wb, ltid)) return;
            b = (int)claim[0] - 1;
            if (b < 0) break;
        } else {
            if (jq >= nFb) break;
            b = fbList[jq];
        }
        b = uni_i(b);
        // beyond this team's registers or its share of the partials buffer (the host sizes teams so that this does not happen):
        // the problem keeps its pre-written ROMAN_ST_INTERNAL record
        if ((int64_t)WIDE_KW * NWG * 64 < (int64_t)st[b].L ||
            ((int64_t)NWG * WIDE_MAXCH + (((int64_t)st[b].L + 63) >> 6) + 4) * 128 > (int64_t)partStride) continue;
        const ProbDesc pd = probs[b];
        const int L = uni_i(st[b].L), rb = uni_i(st[b].rowBase);
        const int64_t lo = pd.liveOff;
        const int nsl = (L + 63) >> 6;
        const uint32_t Tfull = st[b].nnzCap >> 8;               // steps of the flat stream (a step = one quad of the 64 lanes of a slice: 256 entries)
        const uint32_t* perm = permPool + lo; const uint32_t* sbase = sliceBasePool + lo;
        const IdxT* cols = colsPool + st[b].nnzOff; const double* vals = valsPool + st[b].nnzOff;
        IdxT* colsK = colsC + st[b].nnzOff; double* valsK = valsC + st[b].nnzOff;   // the compacted copy: same slice bases, fewer steps per slice
        double* xva = vXa + rb; double* xvb = vXb + rb;
        unsigned long long* bma = bmPool; unsigned long long* bmb = bmPool + bmWords;
        const int kw = min(WIDE_KW, (nsl + NWG - 1) / NWG);     // element rounds in use (the host guarantees L <= WIDE_KW * NWG * 64)
#define CUMW(s_) (cw[(s_)])
#define MEMW(s_) (sbase[(s_)] >> 8)                              /* first step of slice s_ in memory (both copies) */
#define FORK(k_) _Pragma("unroll") for (int k_ = 0; k_ < WIDE_KW; ++k_) if (k_ < kw)
        // The stream in use: the full matrix (cw = the layout's own slice bases) or its column-compacted copy (fewer steps per
        // slice, the slices at their old places).  Chunks of CS steps, dealt round-robin to the waves (chunk c to wave c % NWG):
        // at any moment the waves of the grid read one contiguous window of the matrix.  Every wave takes the same number
        // m <= WIDE_MAXCH of chunks (the last round may be short): about 96 steps per chunk (chunk_steps), more when the matrix is larger.
        uint32_t T = Tfull, CS = 1u, nCh = 0u;
        int cmode = 0 /* stream in use: 0 full, 1 the column-compacted copy, 2 my column block of the half copy (pull + push) */, ncomp = 0, winPass = 0, winOut = 0; bool haveCopy = false;
        int cNWG = NWG, cgw = gwc;                              // the waves the stream in use is dealt to and my id among them (half copy: the workgroups of my column block)
        // the half copy: column blocks, their width, my block, its workgroups / my rank among them, its first column, the fixed-point scale of the pass
        bool upOn = false;                                      // (everything else about the copy lives in sh.up*: read where needed, no registers across the passes)
        uint32_t copyCols = 0u, Tcopy = 0u;                     // compaction state (identical in every workgroup of the team)
        uint32_t pcf[WIDE_KW], pcn[WIDE_KW];
        static_assert(WIDE_MAXCH <= 8, "sS capacity");
        auto chunk_steps = [&](uint32_t T_, uint32_t nwg_) -> uint32_t {   // steps per chunk of a stream of T_ steps dealt to nwg_ waves
            // (about 96 steps per chunk: one or two long chunks per wave — round 3 had settled on 16 steps, up to eight chunks per wave; measured
            //  again in round 6, one / two / four / auto chunks per wave: 64 x L = 10 000 37.8 / 37.9 / 38.4 / 40.0 ms of solve, on the half copy
            //  35.5 / 35.1 / 37.4 / 37.0; n = m = 200 21.8 / 22.3 / 22.8 / 22.4 — fewer pieces per slice for the owners to collect)
            uint32_t mch = max(1u, min((uint32_t)WIDE_MAXCH, (T_ + nwg_ * 96u - 1u) / (nwg_ * 96u)));
            if ((tune >> 8) & 0xff) mch = min((uint32_t)WIDE_MAXCH, (uint32_t)((tune >> 8) & 0xff));
            return max(1u, (T_ + nwg_ * mch - 1u) / (nwg_ * mch));
        };
        auto geometry = [&]() {                                 // (cw, T, cNWG and cgw are set; every thread of the workgroup calls this)
            CS = chunk_steps(T, (uint32_t)cNWG);
            nCh = (T + CS - 1u) / CS;                            // <= cNWG * WIDE_MAXCH
            __syncthreads();                                    // (the previous stream is done with sS; cw is complete)
            if (lane < WIDE_MAXCH) {                            // lane j: first slice of this wave's chunk j (-1: no such chunk)
                const uint32_t c = (uint32_t)cgw + (uint32_t)lane * (uint32_t)cNWG;
                int s_ = -1;
                if (c < nCh) {                                  // largest s with cumW[s] <= first step of the chunk
                    const uint32_t t0 = c * CS;
                    int lo_ = 0, hi_ = nsl;
                    while (hi_ - lo_ > 1) { const int mid_ = (lo_ + hi_) >> 1; if (CUMW(mid_) <= t0) lo_ = mid_; else hi_ = mid_; }
                    s_ = lo_;
                }
                sh.sS[w][lane] = s_;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < WIDE_KW; ++k) {                 // pieces (chunk c, slice s_) have id c + s_
                const int s_ = gw + k * NWG;
                pcf[k] = 0u; pcn[k] = 0u;
                if (k < kw && s_ < nsl) {
                    const uint32_t a0 = CUMW(s_), e0 = CUMW(s_ + 1);
                    if (e0 > a0) { pcf[k] = a0 / CS + (uint32_t)s_; pcn[k] = (e0 - 1u) / CS - a0 / CS + 1u; }
                }
            }
        };
        auto full_stream = [&]() {                              // cw <- the layout's own slice bases
            __syncthreads();
            for (int p = ltid; p <= nsl; p += WIDE_NT) cw[p] = p < nsl ? MEMW(p) : Tfull;
            T = Tfull; cmode = 0; cNWG = NWG; cgw = gwc;
            geometry();
        };
        auto copy_stream = [&]() {                              // cw <- prefix of the copy's slice widths (wNew, written by its compaction)
            __syncthreads();
            if (w == 0) {
                uint32_t run = 0u;
                for (int p0 = 0; p0 < nsl; p0 += WAVE) {
                    const uint32_t wv = (p0 + lane < nsl) ? __hip_atomic_load(wNew + p0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    const uint32_t inc = wave_incl_scan(wv);
                    if (p0 + lane < nsl) cw[p0 + lane] = run + inc - wv;
                    run += (uint32_t)__shfl((int)inc, 63);
                }
                if (lane == 0) cw[nsl] = run;
            }
            __syncthreads();
            T = cw[nsl]; Tcopy = T; cmode = 1; cNWG = NWG; cgw = gwc;
            geometry();
        };
        full_stream();

        int status = ROMAN_ST_OK;
        roman_stats_t S;
        S.n_assoc_in = pd.nA; S.n_live = L; S.nnz_upper = (int64_t)st[b].nnzUpper;
        S.n_pass = 0; S.outer_iters = 0; S.inner_iters = 0; S.ls_trials = 0; S.score = 0.0; S.d_final = 0.0;
        if (pd.n1 == 0 || pd.n2 == 0) status |= ROMAN_ST_EMPTY_MAP;
        double F = 0.0, d = 0.0, usum = 0.0;
        int n_pass = 0, ls_trials = 0, inner_iters = 0, i = 0;

        // owned elements (registers) and the pieces of their slices
        double u[WIDE_KW], Mu[WIDE_KW], Cu[WIDE_KW], sd[WIDE_KW], tt[WIDE_KW], Mn[WIDE_KW], Cn[WIDE_KW], tb[WIDE_KW], ta[WIDE_KW];
        int kk[WIDE_KW]; bool in[WIDE_KW];
#pragma unroll
        for (int k = 0; k < WIDE_KW; ++k) {
            const int s_ = gw + k * NWG;
            const int pos = (s_ << 6) + lane;
            in[k] = k < kw && pos < L;
            kk[k] = in[k] ? (int)perm[pos] : 0;
            sd[k] = in[k] ? ld[lo + kk[k]] : 0.0;
            const int a_ = in[k] ? lp[lo + kk[k]] : 0;
            if (in[k]) plp[lo + pos] = a_;
            u[k] = in[k] ? (u0 ? u0[lo + a_] : 1.0) : 0.0;
            Mu[k] = Cu[k] = tt[k] = Mn[k] = Cn[k] = tb[k] = ta[k] = 0.0;
        }
        double dummy1[1] = {0.0};
        bool alive = true;
#ifdef ROMAN_SOLVE_TIMING
        unsigned long long wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wcnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long wlast = wall_clock64();
#define WMARK(slot) do { const unsigned long long t__ = wall_clock64(); wacc[slot] += t__ - wlast; wcnt[slot] += 1; wlast = t__; } while (0)
#else
#define WMARK(slot) do { } while (0)
#endif

        // x (one value per owned element) -> the vector other compute units gather from (by position, like the column labels)
        // together with its support bit map: one ballot word per wave and round (a wave owns 64 consecutive positions)
        auto publish = [&](double* xv, unsigned long long* bm, const double (&x)[WIDE_KW]) {
            FORK(k) {
                if (in[k]) st_pub(xv + (((gw + k * NWG) << 6) + lane), x[k]);
                const unsigned long long m_ = __ballot(in[k] && x[k] > 0.0);
                if (lane == 0 && gw + k * NWG < nsl) __hip_atomic_store(bm + (gw + k * NWG), m_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // one piece [t, stop) of slice s: partial row sums -> partials buffer.  Blocks of WIDE_U quads (three 16-byte loads
        // per lane and quad), the next block's matrix loads in flight while the current block gathers and accumulates.
        constexpr bool W16 = sizeof(IdxT) == 2;                  // 16-bit column words: 8 bytes per lane and quad, no C flag, 0xffff = no column
        // (the half copy's code costs the whole kernel 20 registers and 220 bytes of scratch per lane — several us per pass of EVERY problem
        //   —: an instantiation of its own, taken where the copy saves more than that: teams on live sets of >= 8 000)
        constexpr bool HALF_ON = HALF && W16;
        typedef typename std::conditional<W16, unsigned long long, uint4_t>::type cword_t;
        auto piece = [&](const double* xv, uint32_t nl, uint32_t tm /* first step in memory */, uint32_t n /* steps */, uint32_t pid) {
            double am = 0.0, ac = 0.0;
            const cword_t* cp = reinterpret_cast<const cword_t*>(cmode ? (const IdxT*)colsK : cols) + (size_t)tm * 64 + lane;
            const dbl2_t* vp = reinterpret_cast<const dbl2_t*>(cmode ? (const double*)valsK : vals) + (size_t)tm * 128 + lane;
            cword_t cA[WIDE_U], cB[WIDE_U]; dbl2_t vA0[WIDE_U], vA1[WIDE_U], vB0[WIDE_U], vB1[WIDE_U];
#define WIDE_ISSUE(C_, V0_, V1_, off_, n_)                                                                    \
            _Pragma("unroll") for (int e = 0; e < WIDE_U; ++e) {                                              \
                const uint32_t q_ = (off_) + min((uint32_t)e, (n_) - 1u);     /* clamped: a repeated quad is skipped by the consumer */ \
                if (tune & 2) { C_[e] = __builtin_nontemporal_load(cp + (size_t)q_ * 64); V0_[e] = __builtin_nontemporal_load(vp + (size_t)(2u * q_) * 64); V1_[e] = __builtin_nontemporal_load(vp + (size_t)(2u * q_ + 1u) * 64); } \
                else { C_[e] = cp[(size_t)q_ * 64]; V0_[e] = vp[(size_t)(2u * q_) * 64]; V1_[e] = vp[(size_t)(2u * q_ + 1u) * 64]; } \
            }
#define WIDE_CONSUME(C_, V0_, V1_, n_)                                                                        \
            _Pragma("unroll") for (int e = 0; e < WIDE_U; ++e) {                                              \
                if ((uint32_t)e < (n_)) {                                                                     \
                    uint32_t c4[4];                                                                           \
                    if constexpr (W16) { const unsigned long long cw_ = *reinterpret_cast<const unsigned long long*>(&C_[e]); \
                        c4[0] = (uint32_t)(cw_ & 0xffffu); c4[1] = (uint32_t)((cw_ >> 16) & 0xffffu); c4[2] = (uint32_t)((cw_ >> 32) & 0xffffu); c4[3] = (uint32_t)(cw_ >> 48); } \
                    else { const uint4_t cw_ = *reinterpret_cast<const uint4_t*>(&C_[e]); c4[0] = cw_.x; c4[1] = cw_.y; c4[2] = cw_.z; c4[3] = cw_.w; } \
                    const double v4[4] = {V0_[e].x, V0_[e].y, V1_[e].x, V1_[e].y};                            \
                    _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                           \
                        const uint32_t ci = W16 ? c4[h] : (c4[h] & 0x7fffffffu);                              \
                        double uq = 0.0;                                                                      \
                        if (ci < nl) uq = xl[ci];                                                             \
                        else if (ci < (uint32_t)L && ((bml[ci >> 6] >> (ci & 63u)) & 1ull)) uq = xv[ci];     \
                        am = fma(v4[h], uq, am);                                                              \
                        ac += (!W16 && (c4[h] & 0x80000000u)) ? 0.0 : uq;                                     \
                    }                                                                                         \
                }                                                                                             \
            }
            WIDE_ISSUE(cA, vA0, vA1, 0u, n)
            for (uint32_t o = 0; o < n; o += 2 * WIDE_U) {
                const uint32_t n1 = (n > o + WIDE_U) ? n - o - WIDE_U : 0u;
                if (n1) { WIDE_ISSUE(cB, vB0, vB1, o + WIDE_U, n1) }
                WIDE_CONSUME(cA, vA0, vA1, n - o)
                if (n1) {
                    const uint32_t n2 = (n > o + 2 * WIDE_U) ? n - o - 2 * WIDE_U : 0u;
                    if (n2) { WIDE_ISSUE(cA, vA0, vA1, o + 2 * WIDE_U, n2) }
                    WIDE_CONSUME(cB, vB0, vB1, n1)
                }
            }
#undef WIDE_CONSUME
            double* pp = part + ((size_t)pid * 64 + lane) * 2;
            st_pub(pp, am); st_pub(pp + 1, ac);
        };
        // One piece of my column block of the HALF copy (round 6): a stored entry (p, q, v) serves both (p, q) and (q, p) — pulled
        // into the row's registers ((M x)_p += v x_q, (C x)_p += x_q: x of the block's columns is in LDS, all of it), and pushed
        // into the block's accumulators ((M x)_q += v x_p, (C x)_q += x_p: two ds_add_u64 of fixed-point terms rint(. 2^s), s from the
        // vector's largest element so that a column's sum stays below 2^62 — integer sums: no order, k_solve_up's "exact accumulation").
        // Column labels of the copy are relative to the block's first column; 0xffff (>= Wc) is padding.
        auto piece_up = [&](const double* xv, int s_, uint32_t tm /* first step in the mirror pools */, uint32_t n /* steps */, uint32_t pid, double fxScale, int Wc, const uint16_t* rowAt /* my block's row order */) {
            if constexpr (HALF_ON) {
            const uint32_t rp = (uint32_t)rowAt[((size_t)s_ << 6) + lane];       // (a slice of the block's own order: the lane's row is looked up)
            const double xr = rp < (uint32_t)L ? xv[rp] : 0.0;
            const cword_t* cp = reinterpret_cast<const cword_t*>((const IdxT*)colsK) + (size_t)tm * 64 + lane;
            const dbl2_t* vp = reinterpret_cast<const dbl2_t*>((const double*)valsK) + (size_t)tm * 128 + lane;
            cword_t cA[WIDE_U], cB[WIDE_U]; dbl2_t vA0[WIDE_U], vA1[WIDE_U], vB0[WIDE_U], vB1[WIDE_U];
            WIDE_ISSUE(cA, vA0, vA1, 0u, n)
            double am = 0.0, ac = 0.0;
            const double xrS = xr * fxScale;
            const unsigned long long fC = (unsigned long long)__double_as_longlong(xrS + FX_MAGIC) - FX_MAGIC_BITS;
            const bool push = xr > 0.0;
            l_vec_cp xloc = (l_vec_cp)xl;
            l_acc_p aM = (l_acc_p)(xl + Wc), aC = (l_acc_p)(xl + 2 * Wc);
#define WIDE_CONSUME_UP(C_, V0_, V1_, n_)                                                                     \
            _Pragma("unroll") for (int e = 0; e < WIDE_U; ++e) {                                              \
                if ((uint32_t)e < (n_)) {                                                                     \
                    const unsigned long long cw_ = *reinterpret_cast<const unsigned long long*>(&C_[e]);      \
                    const uint32_t c4[4] = {(uint32_t)(cw_ & 0xffffu), (uint32_t)((cw_ >> 16) & 0xffffu), (uint32_t)((cw_ >> 32) & 0xffffu), (uint32_t)(cw_ >> 48)}; \
                    const double v4[4] = {V0_[e].x, V0_[e].y, V1_[e].x, V1_[e].y};                            \
                    _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                           \
                        if (c4[h] < (uint32_t)Wc) {                                                           \
                            const double uq = xloc[c4[h]];                                                    \
                            am = fma(v4[h], uq, am); ac += uq;                                                \
                            if (push) {                                                                       \
                                __hip_atomic_fetch_add(aM + c4[h], (unsigned long long)__double_as_longlong(fma(v4[h], xrS, FX_MAGIC)) - FX_MAGIC_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                                __hip_atomic_fetch_add(aC + c4[h], fC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                            }                                                                                 \
                        }                                                                                     \
                    }                                                                                         \
                }                                                                                             \
            }
            for (uint32_t o = 0; o < n; o += 2 * WIDE_U) {
                const uint32_t n1 = (n > o + WIDE_U) ? n - o - WIDE_U : 0u;
                if (n1) { WIDE_ISSUE(cB, vB0, vB1, o + WIDE_U, n1) }
                WIDE_CONSUME_UP(cA, vA0, vA1, n - o)
                if (n1) {
                    const uint32_t n2 = (n > o + 2 * WIDE_U) ? n - o - 2 * WIDE_U : 0u;
                    if (n2) { WIDE_ISSUE(cA, vA0, vA1, o + 2 * WIDE_U, n2) }
                    WIDE_CONSUME_UP(cB, vB0, vB1, n1)
                }
            }
#undef WIDE_CONSUME_UP
            double* pp = part + ((size_t)pid * 64 + lane) * 2;
            st_pub(pp, am); st_pub(pp + 1, ac);
            }
        };
#undef WIDE_ISSUE
        // Where the gathered values come from.  A 320 KB vector against a 32 KB L1 makes every gather an L2 access, and the
        // whole device does 280 G of those per second: 150 us for the 42 M entries of the n = m = 200 problem, more than
        // the matrix stream itself.  So every workgroup first copies into LDS (a) the support bit map of x (one bit per
        // position) and (b) the leading min(mp1, xcap) elements of x — positions are ranks by degree, so the leading
        // elements are the columns most entries point at, and the support collapses onto them within a few passes.  A
        // gather is then an LDS read; only a column beyond the LDS part whose bit is set goes to L2.
        // Column compaction (round 4).  The iterate's support collapses in plateaus (n = m = 200, method 'gravity': 40 000 -> 22 000
        // for 67 passes -> 3 800 for 89 -> 655 for 29 -> ~100 for 57) while its LARGEST position stays high, so streaming a prefix
        // of the rows gains nothing; but a pass only needs the entries whose COLUMN is in the support.  Every workgroup sees the
        // published support bit map anyway: it keeps the union of the supports multiplied during a window of `cwin` passes, and
        // when that union has fallen to `cthr`/256 of the columns in use the team rewrites the matrix without the other columns —
        // a wave per slice, every lane packs its row's kept entries to the front of the slice (same slice bases, in a mirror of
        // the matrix pools; a compacted copy is compacted again in place), the new widths go through one barrier — and streams
        // the copy for every vector whose support lies in the copy's columns.  A vector that leaves them (the line search can
        // re-admit an element) is multiplied with the full matrix, the copy stays; a window that stayed inside the copy and has
        // shrunk again compacts the copy in place, one that left it twice or more makes a new copy from the full matrix.
        // Dropped entries multiplied zeros: the products are the same up to the grouping of the partial sums (pieces are cut
        // from the shorter stream).
        auto slice_compact = [&](int s_, const IdxT* srcC, const double* srcV) {
            const uint32_t ms = MEMW(s_), nst = CUMW(s_ + 1) - CUMW(s_);
            const cword_t* cp = reinterpret_cast<const cword_t*>(srcC) + (size_t)ms * 64 + lane;
            const dbl2_t* vp = reinterpret_cast<const dbl2_t*>(srcV) + (size_t)ms * 128 + lane;
            cword_t* cq = reinterpret_cast<cword_t*>(colsK) + (size_t)ms * 64 + lane;
            dbl2_t* vq = reinterpret_cast<dbl2_t*>(valsK) + (size_t)ms * 128 + lane;
            const uint32_t inert = W16 ? 0xffffu : ((uint32_t)((s_ << 6) + lane) | 0x80000000u);     // fb_inert of this lane's row
            uint32_t kc = 0u, pc[4] = {inert, inert, inert, inert}; double pv[4] = {0.0, 0.0, 0.0, 0.0};
            auto put = [&](uint32_t q_) {                        // the pending quad -> quad q_ of this lane
                if constexpr (W16) cq[(size_t)q_ * 64] = (unsigned long long)pc[0] | ((unsigned long long)pc[1] << 16) | ((unsigned long long)pc[2] << 32) | ((unsigned long long)pc[3] << 48);
                else { uint4_t o_; o_.x = pc[0]; o_.y = pc[1]; o_.z = pc[2]; o_.w = pc[3]; cq[(size_t)q_ * 64] = o_; }
                vq[(size_t)(2u * q_) * 64] = dbl2_t{pv[0], pv[1]}; vq[(size_t)(2u * q_ + 1u) * 64] = dbl2_t{pv[2], pv[3]};
            };
            // (a wave walks its slice alone, step by step: WIDE_CD steps of loads in flight — one step ahead, the walk of the widest slices, a
            //  thousand dependent round trips, WAS the compaction: 560 us per copy of the n = m = 200 matrix, 0.76 TB/s)
            constexpr int WIDE_CD = 4;
            cword_t cR[WIDE_CD]; dbl2_t aR[WIDE_CD], bR[WIDE_CD];
#pragma unroll
            for (int d = 0; d < WIDE_CD; ++d) {
                cR[d] = cword_t{}; aR[d] = dbl2_t{0.0, 0.0}; bR[d] = dbl2_t{0.0, 0.0};
                if ((uint32_t)d < nst) { cR[d] = cp[(size_t)d * 64]; aR[d] = vp[(size_t)(2 * d) * 64]; bR[d] = vp[(size_t)(2 * d + 1) * 64]; }
            }
            for (uint32_t g0 = 0; g0 < nst; g0 += WIDE_CD) {
#pragma unroll
                for (int d = 0; d < WIDE_CD; ++d) {
                    const uint32_t g = g0 + (uint32_t)d;
                    if (g < nst) {
                        const cword_t cC = cR[d]; const dbl2_t v0 = aR[d], v1 = bR[d];
                        if (g + WIDE_CD < nst) { cR[d] = cp[(size_t)(g + WIDE_CD) * 64]; aR[d] = vp[(size_t)(2u * (g + WIDE_CD)) * 64]; bR[d] = vp[(size_t)(2u * (g + WIDE_CD) + 1u) * 64]; }
                        uint32_t c4[4];
                        if constexpr (W16) { const unsigned long long cw_ = *reinterpret_cast<const unsigned long long*>(&cC);
                            c4[0] = (uint32_t)(cw_ & 0xffffu); c4[1] = (uint32_t)((cw_ >> 16) & 0xffffu); c4[2] = (uint32_t)((cw_ >> 32) & 0xffffu); c4[3] = (uint32_t)(cw_ >> 48); }
                        else { const uint4_t cw_ = *reinterpret_cast<const uint4_t*>(&cC); c4[0] = cw_.x; c4[1] = cw_.y; c4[2] = cw_.z; c4[3] = cw_.w; }
                        const double v4[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const uint32_t ci = W16 ? c4[h] : (c4[h] & 0x7fffffffu);
                            // (an inert slot: 0xffff, or flagged with value 0 — a real flagged entry has a non-zero value)
                            const bool real = W16 ? (ci != 0xffffu) : !((c4[h] & 0x80000000u) && v4[h] == 0.0);
                            if (real && ci < (uint32_t)L && ((bmA[ci >> 6] >> (ci & 63u)) & 1ull)) {
                                const uint32_t j_ = kc & 3u;
#pragma unroll
                                for (int z = 0; z < 4; ++z) if ((uint32_t)z == j_) { pc[z] = c4[h]; pv[z] = v4[h]; }
                                ++kc;
                                if ((kc & 3u) == 0u) {
                                    put((kc >> 2) - 1u);
#pragma unroll
                                    for (int z = 0; z < 4; ++z) { pc[z] = inert; pv[z] = 0.0; }
                                }
                            }
                        }
                    }
                }
            }
            if (kc & 3u) {
                put(kc >> 2);
#pragma unroll
                for (int z = 0; z < 4; ++z) { pc[z] = inert; pv[z] = 0.0; }
            }
            const uint32_t nq = (kc + 3u) >> 2;
            uint32_t wmax = nq;
            for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, off));
            for (uint32_t q_ = nq; q_ < wmax; ++q_) put(q_);     // (pending quad is inert here)
            if (lane == 0) __hip_atomic_store(wNew + s_, wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // Where the gathered values come from.  A 320 KB vector against a 32 KB L1 makes every gather an L2 access, and the
        // whole device does 280 G of those per second: 150 us for the 42 M entries of the n = m = 200 problem, more than
        // the matrix stream itself.  So every workgroup first copies into LDS (a) the support bit map of x (one bit per
        // position) and (b) the leading min(mp1, xcap) elements of x — positions are ranks by degree, so the leading
        // elements are the columns most entries point at, and the support collapses onto them within a few passes.  A
        // gather is then an LDS read; only a column beyond the LDS part whose bit is set goes to L2.
        auto stream = [&](const double* xv, const unsigned long long* bm, uint32_t mp1, double xmax /* largest element of xv */) -> bool {
            const uint32_t nl = (tune & 1) ? 0u : min(mp1, (uint32_t)xcap);
            for (uint32_t p = (uint32_t)ltid; p < (uint32_t)nsl; p += WIDE_NT) bml[p] = bm[p];
            const bool upIn = upOn;
            if (!upOn) for (uint32_t p = (uint32_t)ltid; p < nl; p += WIDE_NT) xl[p] = xv[p];
            if (cmax > 0 && Tfull >= 16u * (uint32_t)NWG) {     // support bookkeeping: the same words, the same verdict in every workgroup of the team
                // (a matrix of fewer than 16 steps per wave is not worth a copy: its passes are barriers, and the books cost 3 us a pass)
                if (ltid == 0) { sh.sint[0] = 0; sh.sint[1] = 0; sh.sint[2] = 0; }
                __syncthreads();
                const bool wend = winPass + 1 >= cwin;          // this pass closes a window
                {
                    int out_ = 0, outA = 0, c_ = 0;
                    for (uint32_t p = (uint32_t)ltid; p < (uint32_t)nsl; p += WIDE_NT) {
                        const unsigned long long w_ = bml[p], a_ = (winPass == 0 ? 0ull : bmA[p]) | w_, c0 = haveCopy ? bmC[p] : 0ull;
                        bmA[p] = a_;
                        if (w_ & ~c0) out_ = 1;
                        if (a_ & ~c0) outA = 1;
                        if (wend) c_ += __popcll(a_);
                    }
                    if (out_ | outA) atomicOr(&sh.sint[0], out_ | (outA << 1));
                    if (wend) {
                        for (int off = 32; off > 0; off >>= 1) c_ += __shfl_xor(c_, off);
                        if (lane == 0 && c_) atomicAdd(&sh.sint[1], c_);
                    }
                }
                __syncthreads();
                bool fits = haveCopy && !(sh.sint[0] & 1);      // the vector's support lies in the copy's columns
                ++winPass;
                if (haveCopy && !fits) ++winOut;
                int doc = 0;                                     // 1: compact the copy in place, 2: a new copy from the full matrix
                const uint32_t cntA = (uint32_t)sh.sint[1];
                if (wend) {
                    const bool sub = haveCopy && !(sh.sint[0] & 2);          // the whole window stayed inside the copy
                    if (ncomp < cmax) {
                        if (sub) { if ((unsigned long long)cntA * 256ull <= (unsigned long long)copyCols * (unsigned long long)cthr && Tcopy >= 16u * (uint32_t)NWG) doc = 1; }
                        else if ((!haveCopy || winOut >= 2) && (unsigned long long)cntA * 256ull <= (unsigned long long)L * (unsigned long long)cthr) doc = 2;
                    }
                    winPass = 0; winOut = 0;
                }
#ifdef ROMAN_SOLVE_TIMING
                const unsigned long long tc0_ = wall_clock64();
#endif
                if (doc) {
                    if (doc == 1 && cmode == 0) copy_stream();  // (slice_compact reads the source's widths from cw)
                    if (doc == 2 && cmode != 0) full_stream();
                    upOn = false;                               // (the compacted copy takes the mirror pools: the half copy is gone)
                    const IdxT* srcC = doc == 1 ? (const IdxT*)colsK : cols; const double* srcV = doc == 1 ? (const double*)valsK : vals;
                    for (int s_ = gwc; s_ < nsl; s_ += NWG) slice_compact(s_, srcC, srcV);
                    for (uint32_t p = (uint32_t)ltid; p < (uint32_t)nsl; p += WIDE_NT) bmC[p] = bmA[p];
                    if (!wide_sync<true>(sh, wb, ltid)) return false;
                    haveCopy = true; copyCols = cntA; ++ncomp; fits = true;
                    copy_stream();
                } else if (fits != (cmode == 1)) {
                    if (fits) copy_stream(); else full_stream();
                }
#ifdef ROMAN_SOLVE_TIMING
                if (!HALF_ON && doc) { const unsigned long long tc1_ = wall_clock64(); wacc[7] += tc1_ - tc0_; wcnt[7] += 1; wlast += tc1_ - tc0_; }   // (plain kernel: slot 7 = the compactions, taken out of the stream's slot)
                if (!HALF_ON && cmode == 0) wcnt[6] += 1;       // (passes on the full matrix)
#endif
            }
            if (upIn && !upOn) for (uint32_t p = (uint32_t)ltid; p < nl; p += WIDE_NT) xl[p] = xv[p];   // (left the half copy in this very pass)
            double fxScale = 1.0;
            int Wc = 0, jb = 0; uint32_t upSB_ = 0u, upPB_ = 0u; const uint16_t* rowAt = nullptr;
            if (HALF_ON && upOn) {                              // x over my block's columns, all of it; the pass's fixed-point scale
                Wc = uni_i(sh.upI[1]); jb = uni_i(sh.upI[2]); upSB_ = uni((uint32_t)sh.upSB[jb]); upPB_ = uni((uint32_t)sh.upPB[jb]);
                rowAt = reinterpret_cast<const uint16_t*>(upMeta + (size_t)wb.team * (size_t)bmWords * 1800 + (size_t)WIDE_MAXBLK * bmWords) + (size_t)(2 * WIDE_MAXBLK + jb) * ((size_t)bmWords * 64);
                const int c0 = jb * Wc;
                for (int p = ltid; p < Wc; p += WIDE_NT) xl[p] = (c0 + p < L) ? xv[c0 + p] : 0.0;
                // a term v x 2^s (0 <= v <= 1) below 2^tb, a column's sum of fewer than L terms below 2^62
                int e_ = 0;
                if (xmax > 0.0 && xmax < 1.0e300) (void)frexp(xmax, &e_);
                const int tb = min(49, 62 - (32 - __clz((unsigned)max(L - 1, 1))));
                fxScale = ldexp(1.0, tb - e_);
                if (ltid == 0) sh.upFx = ldexp(1.0, e_ - tb);
            }
            __syncthreads();
            auto chunks = [&](auto half_) {                     // (two copies of the loop: the allocator sees one kind of piece in each)
                constexpr bool H_ = decltype(half_)::value;
                for (int j = 0; j < WIDE_MAXCH; ++j) {
                    int s = uni_i(sh.sS[w][j]);
                    if (s >= 0) {
                        const uint32_t c = (uint32_t)cgw + (uint32_t)j * (uint32_t)cNWG;
                        uint32_t t = c * CS;
                        const uint32_t tEnd = min(T, t + CS);
                        uint32_t sEnd = CUMW(s + 1);
                        while (t < tEnd) {
                            const uint32_t stop = min(sEnd, tEnd);
                            if constexpr (H_) piece_up(xv, s, upSB_ + t, stop - t, upPB_ + c + (uint32_t)s, fxScale, Wc, rowAt);
                            else piece(xv, nl, MEMW(s) + (t - CUMW(s)), stop - t, c + (uint32_t)s);
                            t = stop;
                            if (t < tEnd) { do { ++s; sEnd = CUMW(s + 1); } while (sEnd <= t); }
                        }
                    }
                }
            };
            if (HALF_ON && upOn) chunks(std::true_type{}); else chunks(std::false_type{});
            if (HALF_ON && upOn) {                              // my block's pushed sums -> yPart (write-through), accumulators clean for the next pass
                __syncthreads();
                unsigned long long* ys = yPart + ((size_t)wb.team * (size_t)ySlots + (size_t)wb.tRank) * 2 * (size_t)ycap;
                unsigned long long* aM = reinterpret_cast<unsigned long long*>(xl + Wc);
                const int nv = min(Wc, L - jb * Wc);
                for (int p = ltid; p < nv; p += WIDE_NT) {
                    __hip_atomic_store(ys + p, aM[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ys + ycap + p, aM[Wc + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    aM[p] = 0ull; aM[Wc + p] = 0ull;
                }
            }
            ++n_pass;
            return true;
        };
        // (M x, C x) of the owned rows: the pieces of their slice in ascending order
        auto collect = [&](double (&om)[WIDE_KW], double (&oc)[WIDE_KW]) {
            static_assert(WIDE_KW == 2, "the two owned slices of a wave are collected together");
            // the pieces of BOTH owned slices are loaded together (four of each in flight) and added per slice in ascending order:
            // half as many dependent round trips to the L2 as one slice after the other
            double m_[2] = {0.0, 0.0}, c_[2] = {0.0, 0.0};
            auto add_pieces2 = [&](uint32_t f0, uint32_t n0, uint32_t f1, uint32_t n1) {
                const dbl2_t* p0 = reinterpret_cast<const dbl2_t*>(part) + (size_t)f0 * 64 + lane;
                const dbl2_t* p1 = reinterpret_cast<const dbl2_t*>(part) + (size_t)f1 * 64 + lane;
                const uint32_t nm = max(n0, n1);
                for (uint32_t q0 = 0; q0 < nm; q0 += 4) {
                    dbl2_t x0[4], x1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[e] = p0[(size_t)((q0 + (uint32_t)e < n0) ? q0 + (uint32_t)e : 0u) * 64];
                        x1[e] = p1[(size_t)((q0 + (uint32_t)e < n1) ? q0 + (uint32_t)e : 0u) * 64];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q0 + (uint32_t)e < n0) { m_[0] += x0[e].x; c_[0] += x0[e].y; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q0 + (uint32_t)e < n1) { m_[1] += x1[e].x; c_[1] += x1[e].y; }
                }
            };
            if (HALF_ON && upOn) {                              // pulled pieces of every block's stream, then the pushed sums of the slices' own columns' block
                // (a slice lies in ONE column block: Wc is a multiple of 64; the pushed sums are loaded first and fly while the pieces are added)
                const int nblk = uni_i(sh.upI[0]), Wc = uni_i(sh.upI[1]); const double fxInv = sh.upFx;
                // a row's pulled pieces: per block the pieces of the slice of the block's order that holds the row, at the row's lane slot there
                const uint2* gPin = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(upMeta + (size_t)wb.team * (size_t)bmWords * 1800 + (size_t)WIDE_MAXBLK * bmWords) + (size_t)3 * WIDE_MAXBLK * ((size_t)bmWords * 64));
                {   // (both owned rows of a lane together: their records, then their pieces, in flight at once)
                    const int pos0 = (gw << 6) + lane, pos1 = ((gw + NWG) << 6) + lane;
                    const bool on0 = kw > 0 && in[0], on1 = kw > 1 && in[1];
                    for (int j_ = 0; j_ < nblk; ++j_) {
                        const size_t jo = (size_t)j_ * ((size_t)bmWords * 64);
                        const uint2 pin0 = on0 ? gPin[jo + pos0] : make_uint2(0u, 0u), pin1 = on1 ? gPin[jo + pos1] : make_uint2(0u, 0u);
                        const uint32_t n0 = pin0.y & 0xffffu, n1 = pin1.y & 0xffffu;
                        uint32_t nmax = max(n0, n1);
                        for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
                        const dbl2_t* p0 = reinterpret_cast<const dbl2_t*>(part) + (size_t)pin0.x * 64 + (pin0.y >> 16);
                        const dbl2_t* p1 = reinterpret_cast<const dbl2_t*>(part) + (size_t)pin1.x * 64 + (pin1.y >> 16);
                        for (uint32_t q0 = 0; q0 < nmax; q0 += 4) {
                            dbl2_t x0[4], x1[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                x0[e] = p0[(size_t)((q0 + (uint32_t)e < n0) ? q0 + (uint32_t)e : 0u) * 64];
                                x1[e] = p1[(size_t)((q0 + (uint32_t)e < n1) ? q0 + (uint32_t)e : 0u) * 64];
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (q0 + (uint32_t)e < n0) { m_[0] += x0[e].x; c_[0] += x0[e].y; }
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (q0 + (uint32_t)e < n1) { m_[1] += x1[e].x; c_[1] += x1[e].y; }
                        }
                    }
                }
                // (a slice lies in ONE column block: Wc is a multiple of 64)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pos = ((gw + k * NWG) << 6) + lane;
                    int jr = 0;
#pragma unroll
                    for (int z = 1; z < WIDE_MAXBLK; ++z) jr += (z < nblk && pos >= z * Wc) ? 1 : 0;
                    const bool on_ = k < kw && in[k];
                    const unsigned long long* ys = yPart + ((size_t)wb.team * (size_t)ySlots + (size_t)sh.upCU[on_ ? jr : 0]) * 2 * (size_t)ycap + (on_ ? pos - jr * Wc : 0);
                    const int ng = on_ ? sh.upG[jr] : 0;
                    unsigned long long sm = 0ull, sc = 0ull;
                    for (int g0 = 0; g0 < ng; g0 += 4) {
                        unsigned long long a2[4], b2[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const size_t o_ = (size_t)min(g0 + e, ng - 1) * 2 * (size_t)ycap; a2[e] = ys[o_]; b2[e] = ys[o_ + ycap]; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (g0 + e < ng) { sm += a2[e]; sc += b2[e]; }
                    }
                    if (ng > 0) { m_[k] += fx_decode(sm, fxInv); c_[k] += fx_decode(sc, fxInv); }
                }
            } else if (kw < 2) {                                // one owned slice (the whole device on a problem of fewer slices than waves): eight in flight
                const dbl2_t* pp = reinterpret_cast<const dbl2_t*>(part) + (size_t)pcf[0] * 64 + lane;
                for (uint32_t q0 = 0; q0 < pcn[0]; q0 += 8) {
                    dbl2_t x_[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x_[e] = pp[(size_t)min(q0 + (uint32_t)e, pcn[0] - 1u) * 64];
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (q0 + (uint32_t)e < pcn[0]) { m_[0] += x_[e].x; c_[0] += x_[e].y; }
                }
            } else add_pieces2(pcf[0], pcn[0], pcf[1], pcn[1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) { om[k] = (k < kw && in[k]) ? m_[k] : 0.0; oc[k] = (k < kw && in[k]) ? c_[k] : 0.0; }
        };
        // trial vector max(base + alpha grad(base), 0) of the owned elements, with the partials of sum t^2 and sum t
        auto trial = [&](const double (&ub)[WIDE_KW], const double (&mb)[WIDE_KW], const double (&cb)[WIDE_KW], double us, double alpha,
                         double (&out)[WIDE_KW], double& ss, double& s1, double& mp, double& mx /* largest element: the scale of a pull + push pass */) {
            FORK(k) {
                const double up = ub[k];
                const double g = (((sd[k] + d) * up - d * us) + mb[k]) + cb[k] * d;
                double t = up + alpha * g;
                t = (in[k] && t > 0.0) ? t : 0.0;
                out[k] = t; ss += t * t; s1 += t; mx = fmax(mx, t);
                if (t > 0.0) mp = fmax(mp, (double)(((gw + k * NWG) << 6) + lane + 1));
            }
        };
        // mean of (M u)_p / Cbu_p over the active set, and the two sums F = A + d B is made of — one grid reduction
        auto d_ratio = [&](bool absval, double (&r4)[4]) -> bool {
            r4[0] = r4[1] = r4[2] = r4[3] = 0.0;
            FORK(k) if (in[k]) {
                const double up = u[k], Cbu = (usum - Cu[k]) - up, mdu = Mu[k] + sd[k] * up;
                if (Cbu > P.eps && up > P.eps) { const double r_ = mdu / Cbu; r4[0] += absval ? fabs(r_) : r_; r4[1] += 1.0; }
                r4[2] += up * mdu; r4[3] += up * ((up - usum) + Cu[k]);
            }
            return wide_reduce<4>(r4, sh, slots, wb, ltid);
        };

        // The iteration as a state machine around ONE stream call site (rescale pass / initial pass / line-search trial).
        // A trial costs TWO grid barriers: the one behind the stream, and one that carries the objective of the trial just
        // multiplied TOGETHER with the sums of both vectors that can come next — the backtracked trial from the same base
        // (published to xva) and the first trial from the accepted vector (xvb): whichever the objective selects is
        // already out when the barrier opens.  (The host sends problems with maxiniters < 1 or maxlsiters < 1 to k_solve.)
        // The HALF copy (round 6; `upOn`, cmode 2).  While the iterate's support is wide no column-compacted copy pays (the columns of the
        // early plateau hold three quarters of the entries), but every pair is stored twice.  At the start of a problem the team writes,
        // into the mirror pools, ONE of the two stored copies of every pair (up_keeps: a checkerboard rule — every row keeps about half of
        // its entries) — cut into COLUMN BLOCKS of at most ycap columns, a flat quad stream per block (a slice's rows' entries of block
        // 0, padded to the slice's width there, then the next slice; then block 1 ...), labels relative to the block — and deals the
        // team's workgroups to the blocks in proportion to their steps.  A workgroup keeps ALL of x over its block's columns in LDS
        // next to two fixed-point accumulators per column: its waves pull into the rows' partials as before and push into the
        // accumulators (piece_up); after the stream the accumulators go to yPart, and the owner of a row adds its pulled pieces
        // (every block) and the pushed sums of the workgroups of its own column's block.  10 bytes per stored pair and pass instead
        // of 20 — and 1.08 x the entries in padding: the rows of a block's stream are in the block's OWN order (windows of 512 positions
        // sorted by the number of entries the row keeps in that block: the 64 rows of a slice hold nearly equal numbers; by position a
        // (slice, block) cell was as wide as the longest of 64 binomial counts: 1.43 x with two blocks, 1.5 x with seven) —, + 32 bytes per
        // column and workgroup of the block.  The first column compaction takes the mirror pools (and every pass thereafter the old
        // path).  Measured: the stream of a pass is bandwidth-bound at the same rate either way; 64 x L = 10 000 on teams
        // 39.7 -> 36.6 ms of solve, smaller live sets lose (the rest of a pass is slower in this instantiation).
        auto build_upper = [&]() -> bool {
            if constexpr (HALF_ON) {
            int nblk = 1, Wc = 0, jb = 0, GU = 1, rkU = 0; uint32_t upSteps = 0u;
            if (!(ucfg & 1) || L < 1024 || Tfull < 16u * (uint32_t)NWG || G > ySlots || G < 2) return true;
            const int WcMax = min(ycap, (int)((uint32_t)xcap / 3u) & ~63);
            if (WcMax < 64) return true;
            nblk = (L + WcMax - 1) / WcMax;
            if (nblk > WIDE_MAXBLK || nblk > G) { nblk = 1; return true; }
            Wc = (((L + nblk - 1) / nblk) + 63) & ~63;          // (<= WcMax: a multiple of 64 that is at least L / nblk)
            const int nsl1 = nsl + 1;
            // per team: per block the kept entries of every row, the row's rank in the block's own ROW ORDER (windows of 512 positions sorted by
            // that count: the 64 rows of a slice of the block's stream then hold nearly equal numbers of entries — by position a (slice, block)
            // cell was as wide as the longest of 64 binomial counts, 1.43 x the entries with two blocks), the row at a rank, and where a row's
            // pulled pieces are (first piece, pieces | lane slot << 16)
            const size_t Lcap = (size_t)bmWords * 64;
            uint32_t* upw = upMeta + (size_t)wb.team * (size_t)bmWords * 1800;
            uint16_t* gCnt = reinterpret_cast<uint16_t*>(upw + (size_t)WIDE_MAXBLK * bmWords);
            uint16_t* gRank = gCnt + WIDE_MAXBLK * Lcap;
            uint16_t* gRow = gRank + WIDE_MAXBLK * Lcap;
            uint2* gPin = reinterpret_cast<uint2*>(gRow + WIDE_MAXBLK * Lcap);
            // which of the two stored copies of a pair the half copy keeps: (p, q) with p < q when p + q is odd, with p > q when it is even —
            // every row keeps about half of its entries whatever the positions of its neighbours (by position order alone the rows of a
            // slice keep very different shares, and a slice is as wide as its longest row: 0.92 of the full stream instead of 0.5)
            auto up_keeps = [](uint32_t rp_, uint32_t ci_) -> bool { return ((rp_ + ci_) & 1u) ? (rp_ < ci_) : (rp_ > ci_); };
            auto blk_of = [&](uint32_t ci) -> int { int j_ = 0; _Pragma("unroll") for (int z = 1; z < WIDE_MAXBLK; ++z) j_ += (z < nblk && ci >= (uint32_t)(z * Wc)) ? 1 : 0; return j_; };
            // (1) widths: a wave per slice counts, per lane, the entries the copy keeps per block
            for (int s_ = cgw; s_ < nsl; s_ += NWG) {
                const uint32_t ms = MEMW(s_), nst = CUMW(s_ + 1) - CUMW(s_);
                const unsigned long long* cp = reinterpret_cast<const unsigned long long*>(cols) + (size_t)ms * 64 + lane;
                const uint32_t rp = ((uint32_t)s_ << 6) + (uint32_t)lane;
                uint32_t cnt[WIDE_MAXBLK];
#pragma unroll
                for (int z = 0; z < WIDE_MAXBLK; ++z) cnt[z] = 0u;
                unsigned long long cN = nst > 0u ? cp[0] : ~0ull;
                for (uint32_t g = 0; g < nst; ++g) {
                    const unsigned long long cC = cN;
                    if (g + 1u < nst) cN = cp[(size_t)(g + 1u) * 64];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const uint32_t ci = (uint32_t)(cC >> (16 * h)) & 0xffffu;
                        if (ci < (uint32_t)L && up_keeps(rp, ci)) {
                            const int j_ = blk_of(ci);
#pragma unroll
                            for (int z = 0; z < WIDE_MAXBLK; ++z) cnt[z] += (z == j_) ? 1u : 0u;
                        }
                    }
                }
#pragma unroll
                for (int z = 0; z < WIDE_MAXBLK; ++z) if (z < nblk) gCnt[(size_t)z * Lcap + rp] = (uint16_t)min(cnt[z], 0xffffu);
            }
            if (!wide_sync<true>(sh, wb, ltid)) return false;
            // (1b) the blocks' row orders: a wave ranks a window of 512 positions of one block by (count descending, position ascending)
            {
                const int nwin = (nsl * 64 + 511) / 512;
                uint16_t* sc = reinterpret_cast<uint16_t*>(xl) + (size_t)w * 512;     // (the gathered vector's region: no stream yet)
                for (int t_ = cgw; t_ < nblk * nwin; t_ += NWG) {
                    const int j_ = t_ / nwin, base = (t_ - j_ * nwin) * 512, nrow = min(512, nsl * 64 - base);
                    uint32_t c_[8];
#pragma unroll
                    for (int i_ = 0; i_ < 8; ++i_) {
                        const int idx = lane + 64 * i_;
                        c_[i_] = idx < nrow ? (uint32_t)gCnt[(size_t)j_ * Lcap + base + idx] : 0u;
                        sc[idx] = (uint16_t)c_[i_];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    uint32_t rk_[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                    for (int m_ = 0; m_ < nrow; ++m_) {
                        const uint32_t cm = sc[m_];
#pragma unroll
                        for (int i_ = 0; i_ < 8; ++i_) rk_[i_] += (cm > c_[i_] || (cm == c_[i_] && m_ < lane + 64 * i_)) ? 1u : 0u;
                    }
#pragma unroll
                    for (int i_ = 0; i_ < 8; ++i_) {
                        const int idx = lane + 64 * i_;
                        if (idx < nrow) { gRank[(size_t)j_ * Lcap + base + idx] = (uint16_t)(base + rk_[i_]); gRow[(size_t)j_ * Lcap + base + rk_[i_]] = (uint16_t)(base + idx); }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            if (!wide_sync<true>(sh, wb, ltid)) return false;
            // (2) every workgroup: cumulative widths per block (LDS, in the region of the gathered vector: no stream yet), the deal
            uint32_t* lw = reinterpret_cast<uint32_t*>(xl);     // [block][nsl1]
            if (w < nblk) {
                uint32_t run = 0u;
                for (int p0 = 0; p0 < nsl; p0 += WAVE) {
                    // (a slice of the block's order is as wide as its first row: the windows are sorted by count, eight slices each)
                    const uint32_t wv = (p0 + lane < nsl) ? (((uint32_t)gCnt[(size_t)w * Lcap + gRow[(size_t)w * Lcap + (size_t)(p0 + lane) * 64]] + 3u) >> 2) : 0u;
                    const uint32_t inc = wave_incl_scan(wv);
                    if (p0 + lane < nsl) lw[w * nsl1 + p0 + lane] = run + inc - wv;
                    run += (uint32_t)__shfl((int)inc, 63);
                }
                if (lane == 0) lw[w * nsl1 + nsl] = run;
            }
            __syncthreads();
            if (ltid == 0) {
                unsigned long long Ttot = 0ull; int ne = 0;
                for (int j_ = 0; j_ < nblk; ++j_) { const uint32_t t_ = lw[j_ * nsl1 + nsl]; Ttot += t_; ne += t_ ? 1 : 0; }
                bool ok = Ttot > 0ull && Ttot <= (unsigned long long)Tfull && ne <= G;
                // (the pushed sums a pass writes and reads, 32 bytes per column and workgroup, against the 2560 bytes of a step it saves)
                ok = ok && (unsigned long long)G * (unsigned long long)Wc * 32ull * 2ull <= Ttot * 2560ull;
                int rem = G, cu = 0; unsigned long long remT = Ttot; uint32_t pb = 0u, sb = 0u;
                for (int j_ = 0; j_ < WIDE_MAXBLK; ++j_) {
                    const uint32_t t_ = j_ < nblk ? lw[j_ * nsl1 + nsl] : 0u;
                    sh.upT[j_] = t_; sh.upSB[j_] = sb; sb += t_; sh.upCU[j_] = cu; sh.upPB[j_] = pb; sh.upCS[j_] = 1u; sh.upG[j_] = 0;
                    if (!ok || t_ == 0u) continue;
                    --ne;
                    int g_ = (int)(((unsigned long long)rem * t_ + remT / 2ull) / remT);
                    g_ = max(1, min(g_, rem - ne));
                    sh.upG[j_] = g_; cu += g_; rem -= g_; remT -= t_;
                    const uint32_t cs_ = chunk_steps(t_, (uint32_t)g_ * WIDE_NW);
                    sh.upCS[j_] = cs_; pb += (t_ + cs_ - 1u) / cs_ + (uint32_t)nsl;
                }
                if (((int64_t)pb + 4) * 128 > (int64_t)partStride) ok = false;
                sh.sint[0] = ok ? 1 : 0; sh.sint[1] = (int)min(Ttot, 0x7fffffffull);
            }
            __syncthreads();
            const bool ok = sh.sint[0] != 0;
            upSteps = (uint32_t)sh.sint[1];
            __syncthreads();
            if (!ok) { nblk = 1; return true; }                 // (the same verdict in every workgroup of the team: the same words)
#pragma unroll
            for (int z = 0; z < WIDE_MAXBLK; ++z) if (sh.upG[z] > 0 && wb.tRank >= sh.upCU[z] && wb.tRank < sh.upCU[z] + sh.upG[z]) jb = z;
            GU = sh.upG[jb]; rkU = wb.tRank - sh.upCU[jb];
            // (3) the copy: every lane scatters the entries of its row that the copy keeps to the row's slot in every block's own order — slice
            // rank / 64, lane slot rank % 64 —, pads the slot to that slice's width, and notes where the row's pulled pieces will be
            for (int s_ = cgw; s_ < nsl; s_ += NWG) {
                const uint32_t ms = MEMW(s_), nst = CUMW(s_ + 1) - CUMW(s_);
                const unsigned long long* cp = reinterpret_cast<const unsigned long long*>(cols) + (size_t)ms * 64 + lane;
                const dbl2_t* vp = reinterpret_cast<const dbl2_t*>(vals) + (size_t)ms * 128 + lane;
                uint16_t* cq = reinterpret_cast<uint16_t*>(colsK); double* vq = valsK;
                const uint32_t rp = ((uint32_t)s_ << 6) + (uint32_t)lane;
                uint32_t cur[WIDE_MAXBLK], rkb[WIDE_MAXBLK];
#pragma unroll
                for (int z = 0; z < WIDE_MAXBLK; ++z) { cur[z] = 0u; rkb[z] = z < nblk ? (uint32_t)gRank[(size_t)z * Lcap + rp] : 0u; }
                // (a slot's entries start vl / 64 of the way into it, cyclically: the lanes of a wave do not push to the same few accumulators
                //  in the same instruction)
                auto put = [&](int j_, uint32_t e0_, uint32_t label, double v_) {
                    uint32_t rk_ = 0u;
#pragma unroll
                    for (int z = 0; z < WIDE_MAXBLK; ++z) if (z == j_) rk_ = rkb[z];
                    const uint32_t v_s = rk_ >> 6, vl = rk_ & 63u;
                    const uint32_t a0_ = lw[j_ * nsl1 + v_s], we_ = (lw[j_ * nsl1 + v_s + 1] - a0_) << 2;
                    uint32_t e_ = e0_ + (we_ * vl >> 6);
                    e_ -= e_ >= we_ ? we_ : 0u;
                    const size_t step = (size_t)sh.upSB[j_] + a0_ + (e_ >> 2);
                    const uint32_t h = e_ & 3u;
                    cq[(step * 64 + vl) * 4 + h] = (uint16_t)label;
                    vq[(step * 128 + (size_t)(h >> 1) * 64 + vl) * 2 + (h & 1u)] = v_;
                };
                unsigned long long cN = ~0ull; dbl2_t v0N = dbl2_t{0.0, 0.0}, v1N = dbl2_t{0.0, 0.0};
                if (nst > 0u) { cN = cp[0]; v0N = vp[0]; v1N = vp[64]; }
                for (uint32_t g = 0; g < nst; ++g) {
                    const unsigned long long cC = cN; const dbl2_t v0 = v0N, v1 = v1N;
                    if (g + 1u < nst) { cN = cp[(size_t)(g + 1u) * 64]; v0N = vp[(size_t)(2u * g + 2u) * 64]; v1N = vp[(size_t)(2u * g + 3u) * 64]; }
                    const double v4[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const uint32_t ci = (uint32_t)(cC >> (16 * h)) & 0xffffu;
                        if (ci < (uint32_t)L && up_keeps(rp, ci)) {
                            const int j_ = blk_of(ci);
                            uint32_t e_ = 0u;
#pragma unroll
                            for (int z = 0; z < WIDE_MAXBLK; ++z) if (z == j_) { e_ = cur[z]; cur[z] += 1u; }
                            put(j_, e_, ci - (uint32_t)(j_ * Wc), v4[h]);
                        }
                    }
                }
#pragma unroll
                for (int z = 0; z < WIDE_MAXBLK; ++z) if (z < nblk) {
                    const uint32_t v_s = rkb[z] >> 6, a0 = lw[z * nsl1 + v_s], e0 = lw[z * nsl1 + v_s + 1];
                    const uint32_t we = (e0 - a0) << 2;
                    for (uint32_t e_ = cur[z]; e_ < we; ++e_) put(z, e_, 0xffffu, 0.0);
                    uint32_t f_ = 0u, n_ = 0u;                   // the pieces of the slot's slice: chunks that overlap it
                    if (e0 > a0 && sh.upG[z] > 0) { const uint32_t cs_ = sh.upCS[z]; f_ = sh.upPB[z] + a0 / cs_ + v_s; n_ = (e0 - 1u) / cs_ - a0 / cs_ + 1u; }
                    gPin[(size_t)z * Lcap + rp] = make_uint2(f_, n_ | ((rkb[z] & 63u) << 16));
                }
            }
            __syncthreads();                                    // (every wave is done with the full matrix's widths in cw and with lw)
            for (int p = ltid; p <= nsl; p += WIDE_NT) cw[p] = lw[jb * nsl1 + p];
            __syncthreads();
            for (int p = ltid; p < 2 * Wc; p += WIDE_NT) reinterpret_cast<unsigned long long*>(xl + Wc)[p] = 0ull;
            if (!wide_sync<true>(sh, wb, ltid)) return false;   // the copy is complete (plain stores: release)
            if (ltid == 0) { sh.upI[0] = nblk; sh.upI[1] = Wc; sh.upI[2] = jb; sh.upI[3] = (int)upSteps; }
            T = sh.upT[jb]; cmode = 2; cNWG = GU * WIDE_NW; cgw = w * GU + rkU;
            geometry();
            upOn = true;
            }
            return true;
        };
        enum { PH_RESCALE, PH_INIT, PH_TRIAL };
        if (L > 0) {
            double r4[4] = {0.0, 0.0, 0.0, 0.0};
            double alpha = 1.0, nr = 0.0, s1cur = 0.0;
            int j = 0, k2 = 0;
            const double* xcur = xva; const unsigned long long* bmcur = bma;
            uint32_t mpcur = (uint32_t)L;                       // support bound of the vector in xcur
            double mxcur = 0.0;                                 // ... and its largest element
            auto normalise = [&]() -> bool {                    // u /= |u|, usum = sum u
                double r2[3] = {0.0, 0.0, 0.0};
                FORK(k) { r2[0] += u[k] * u[k]; r2[1] += u[k]; r2[2] = fmax(r2[2], u[k]); }
                if constexpr (HALF_ON) { if (!wide_reduce<3, 1>(r2, sh, slots, wb, ltid)) return false; }
                else { double q2[2] = {r2[0], r2[1]}; if (!wide_reduce<2>(q2, sh, slots, wb, ltid)) return false; r2[0] = q2[0]; r2[1] = q2[1]; }
                const double nr_ = sqrt(r2[0]);
                if (nr_ > 0.0) FORK(k) u[k] /= nr_;
                usum = (nr_ > 0.0) ? r2[1] / nr_ : r2[1];
                mxcur = (nr_ > 0.0) ? (r2[2] / nr_) * 1.0000001 : r2[2];    // (a bound: the division rounds)
                return true;
            };
            int phase = P.rescale_u0 ? PH_RESCALE : PH_INIT;
            if constexpr (HALF_ON) alive = build_upper();
            if (alive && phase == PH_INIT) alive = normalise();
            if (alive) {                                        // the first vector out; the barrier carries its largest element
                publish(xva, bma, u);
                if constexpr (HALF_ON) {
                    double m1[2] = {0.0, 0.0};
                    FORK(k) { m1[0] = fmax(m1[0], u[k]); m1[1] = fmax(m1[1], -u[k]); }
                    alive = wide_reduce<2, 2>(m1, sh, slots, wb, ltid);
                    mxcur = m1[0];
                    if (alive && upOn && m1[1] > 0.0) { upOn = false; full_stream(); }   // (a caller's start vector with negative elements: the pushed sums are unsigned)
                } else alive = wide_reduce<0>(dummy1, sh, slots, wb, ltid);
            }
            while (alive) {
                if (!(alive = stream(xcur, bmcur, mpcur, mxcur))) break;
                WMARK(upOn ? 7 : 2);                            // (timing build: the pull + push passes apart)
                if (!(alive = wide_reduce<0>(dummy1, sh, slots, wb, ltid))) break;
                WMARK(3);
                collect(Mn, Cn);
                if (phase == PH_RESCALE) {                      // u = normalize(M u0 + diag u0)
                    FORK(k) u[k] = in[k] ? Mn[k] + sd[k] * u[k] : 0.0;
                    if (!(alive = normalise())) break;
                    publish(xva, bma, u);
                    if (!(alive = wide_reduce<0>(dummy1, sh, slots, wb, ltid))) break;
                    mpcur = (uint32_t)L;
                    phase = PH_INIT;
                    continue;
                }
                bool new_outer = false;
                if (phase == PH_INIT) {
                    FORK(k) { Mu[k] = Mn[k]; Cu[k] = Cn[k]; }
                    if (!(alive = d_ratio(false, r4))) break;
                    d = (r4[1] > 0.0) ? r4[0] / r4[1] : 0.0;
                    i = 0;
                    if (i >= P.maxoliters) break;
                    new_outer = true;
                } else {                                        // PH_TRIAL: products of the trial vector
                    ++ls_trials;
                    const double unsum = (nr > 0.0) ? s1cur / nr : s1cur;
                    double r6[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // objective, |du|^2, sums of the two candidates, their support bounds, their largest elements
                    FORK(k) {
                        if (nr > 0.0) { tt[k] /= nr; Mn[k] /= nr; Cn[k] /= nr; }
                        const double up = tt[k];
                        const double g = (((sd[k] + d) * up - d * unsum) + Mn[k]) + Cn[k] * d;
                        r6[0] += up * g;
                        const double df = up - u[k]; r6[1] += df * df;
                    }
                    const bool can_back = k2 + 1 < P.maxlsiters, can_next = j + 1 < P.maxiniters;
                    if (can_back) { trial(u, Mu, Cu, usum, alpha * P.beta, tb, r6[2], r6[3], r6[6], r6[8]); publish(xva, bma, tb); }
                    if (can_next) { trial(tt, Mn, Cn, unsum, 1.0, ta, r6[4], r6[5], r6[7], r6[9]); publish(xvb, bmb, ta); }
                    WMARK(4);
                    if constexpr (HALF_ON) { if (!(alive = (wide_reduce<10, 4>(r6, sh, slots, wb, ltid)))) break; }
                    else {                                      // (the largest elements are the half copy's: eight values as before)
                        double r8[8] = {r6[0], r6[1], r6[2], r6[3], r6[4], r6[5], r6[6], r6[7]};
                        if (!(alive = (wide_reduce<8, 2>(r8, sh, slots, wb, ltid)))) break;
#pragma unroll
                        for (int z = 0; z < 8; ++z) r6[z] = r8[z];
                    }
                    WMARK(5);
                    const double Fnew = r6[0], deltaF = Fnew - F;
                    if (deltaF < -P.eps && can_back) {          // backtrack: the shorter step from the same base is already out
                        alpha *= P.beta; ++k2;
                        FORK(k) tt[k] = tb[k];
                        nr = sqrt(r6[2]); s1cur = r6[3]; xcur = xva; bmcur = bma; mpcur = (uint32_t)r6[6]; mxcur = r6[8];
                        continue;
                    }
                    // accept: the trial vector and its products become the current ones
                    const double du = sqrt(r6[1]);
                    F = Fnew; usum = unsum;
                    FORK(k) { u[k] = tt[k]; Mu[k] = Mn[k]; Cu[k] = Cn[k]; }
                    ++inner_iters; ++j;
                    if (!(du < P.tol_u || fabs(deltaF) < P.tol_F || !can_next)) {   // the inner loop goes on: its next trial is already out
                        alpha = 1.0; k2 = 0;
                        FORK(k) tt[k] = ta[k];
                        nr = sqrt(r6[4]); s1cur = r6[5]; xcur = xvb; bmcur = bmb; mpcur = (uint32_t)r6[7]; mxcur = r6[9];
                        continue;
                    }
                    if (!(alive = d_ratio(true, r4))) break;    // end of the inner loop: homotopy update of d
                    if (r4[1] > 0.0) d += r4[0] / r4[1]; else break;
                    ++i;
                    if (i >= P.maxoliters) break;
                    new_outer = true;
                }
                if (new_outer) {                                // first trial of an inner loop: nothing to overlap it with
                    F = r4[2] + d * r4[3];                      // u . gradF at the new d
                    alpha = 1.0; j = 0; k2 = 0;
                    double r2[4] = {0.0, 0.0, 0.0, 0.0};
                    trial(u, Mu, Cu, usum, alpha, tt, r2[0], r2[1], r2[2], r2[3]);
                    publish(xva, bma, tt);
                    WMARK(0);
                    if constexpr (HALF_ON) { if (!(alive = (wide_reduce<4, 2>(r2, sh, slots, wb, ltid)))) break; }
                    else {
                        double r3[3] = {r2[0], r2[1], r2[2]};
                        if (!(alive = (wide_reduce<3, 1>(r3, sh, slots, wb, ltid)))) break;
                        r2[0] = r3[0]; r2[1] = r3[1]; r2[2] = r3[2];
                    }
                    WMARK(1);
                    nr = sqrt(r2[0]); s1cur = r2[1]; xcur = xva; bmcur = bma; mpcur = (uint32_t)r2[2]; mxcur = r2[3];
                    phase = PH_TRIAL;
                }
            }
        }
        if (!alive) return;                                      // a bounded wait expired: give up (the pre-written records say so)
        if (i >= P.maxoliters) status |= ROMAN_ST_MAXITER;
        S.n_pass = n_pass; S.ls_trials = ls_trials; S.inner_iters = inner_iters;
        S.outer_iters = i; S.score = F; S.d_final = d;
#ifdef ROMAN_SOLVE_TIMING
        WMARK(6);
        if (wb.tRank == 0 && ltid == 0 && O.dbg) {              // 100 MHz ticks -> the host prints them as "cycles": x 10 ns
            unsigned long long* dg = O.dbg + (size_t)b * 16;
            for (int t = 0; t < 8; ++t) { dg[t] = wacc[t]; dg[8 + t] = wcnt[t]; }
            dg[8 + 5] = Tfull; dg[8 + 6] = HALF_ON ? (unsigned long long)sh.upI[3] : wcnt[6];            // (steps of the full stream / of the half copy, all blocks)
        }
#endif
#undef WMARK
        // final u by position for the shared tail (plain stores + one release); workgroup 0 selects and writes the pose
        FORK(k) if (in[k]) vU[rb + ((gw + k * NWG) << 6) + lane] = u[k];
        if (!wide_sync<true>(sh, wb, ltid)) return;
        if (wb.tRank == 0)
            finish_one(D, b, pd, feats, assoc, plp, lp, rowPosPool, permPool, O, L > 0 ? vU + rb : nullptr, vS0 + rb,
                       reinterpret_cast<int32_t*>(vS1 + rb), reinterpret_cast<int32_t*>(vS2 + rb), L, rb, lo, F, status, S, sh.red, sh.sint);
#undef CUMW
#undef FORK
    }
}

// k_skipped: result records of the problems that found no workspace (kind 2): ROMAN_ST_WORKSPACE, no associations,
// NaN pose.  One thread per problem.  Runs right before the solver kernels and also resets their problem queues.
// With `fbList` (the whole-device solver is part of the launch): the fallback problems are listed for it (count in
// wideBar line 21, zeroed by the host before this kernel) and their records PRE-WRITTEN as "not finished" (ROMAN_ST_INTERNAL,
// no associations, NaN pose): the solver overwrites a record when it finishes the problem, so a launch that gives up (a
// bounded wait expired) leaves a statement, not stale memory, for every problem it did not get to.
__global__ void __launch_bounds__(256) k_skipped(int B, const ProbDesc* __restrict__ probs, const ProbState* __restrict__ st, SolveOut O,
                                                 int* __restrict__ queue /* the solvers' problem queues (8 ints): cleared here */,
                                                 int32_t* __restrict__ fbList, unsigned* __restrict__ wideBar)
{
    if (blockIdx.x == 0 && (threadIdx.x < 8 || threadIdx.x == 10 || threadIdx.x == 11)) queue[threadIdx.x] = 0;   // ([10], [11]: the stream solver's continuation counters)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && fbList != nullptr && st[b].kind == 1) {
        fbList[atomicAdd(wideBar + 32 * 21, 1u)] = b;
        for (int t = 0; t < 16; ++t) O.T_out[(int64_t)b * 16 + t] = d_nan();
        O.n_assoc_out[b] = 0; O.status_out[b] = ROMAN_ST_INTERNAL; O.nSel[b] = 0;
        if (O.stats_out) { roman_stats_t S{}; S.n_assoc_in = probs[b].nA; S.n_live = st[b].L; O.stats_out[b] = S; }
    }
    if (b >= B || st[b].kind != 2) return;
    for (int t = 0; t < 16; ++t) O.T_out[(int64_t)b * 16 + t] = d_nan();
    O.n_assoc_out[b] = 0; O.status_out[b] = ROMAN_ST_WORKSPACE; O.nSel[b] = 0;
    if (O.stats_out) { roman_stats_t S{}; S.n_assoc_in = probs[b].nA; S.n_live = st[b].L; O.stats_out[b] = S; }
}

// ---------------------------------------------------------------------------------------------
// standalone pose kernel: T_align on given correspondences (one wave per problem)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_pose(int dim, const double* __restrict__ pts1,
                                             const double* __restrict__ pts2,
                                             const int64_t* __restrict__ corrOff,
                                             double* __restrict__ T_out, int32_t* __restrict__ status_out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t o = corrOff[b]; const int k = (int)(corrOff[b + 1] - o);
    if (k < dim) {
        if (lane == 0) { for (int t = 0; t < 16; ++t) T_out[(int64_t)b * 16 + t] = d_nan(); status_out[b] = ROMAN_ST_INSUFFICIENT; }
        return;
    }
    double T[16];
    wave_pose(T, dim, k, lane, [&](int t, double (&a)[3], double (&q)[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] = c < dim ? pts1[(o + t) * dim + c] : 0.0; q[c] = c < dim ? pts2[(o + t) * dim + c] : 0.0; }
    });
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) T_out[(int64_t)b * 16 + t] = T[t];
        status_out[b] = ROMAN_ST_OK;
    }
}


// elementwise math probe for tests
__global__ void k_debug_math(int kind, const double* __restrict__ a, const double* __restrict__ b,
                             int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r;
    switch (kind) {
    case 0: r = sqrt(x); break;
    case 1: r = exp(x); break;
    case 2: r = cbrt(x); break;
    case 3: r = x / y; break;
    case 4: r = pow(x, y); break;
    case 5: r = fx_exp(x); break;
    case 6: r = fx_cbrt(x); break;
    default: r = fma(x, y, x); break;
    }
    out[i] = r;
}

}  // namespace roman
