// roman_hip.hip — host side of libroman_hip.so: context, HBM workspace, launch sequence and the
// C ABI declared in include/roman_hip.h.  gfx950 only; no CPU fallback.
//
// A batch is a pure ENQUEUE: the launch sequence never waits for the GPU.  The sizes of the sparse
// pools (candidate bit matrices, matrix entries) depend on the data; they are allocated from estimates —
// exact where the invariant has no single scores (every association is live), otherwise from the ratios
// seen in earlier batches with the same parameters (read back lazily, never waited for), otherwise from a
// heuristic — and the device checks every problem against the capacity it was given: a problem that does
// not fit is skipped with ROMAN_ST_WORKSPACE and the recorded need sizes the next attempt.  The entry
// points that are synchronous anyway (host pointers, stepwise API) retry by themselves.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kernels.hip.h"

using namespace roman;

namespace {

thread_local std::string g_last_error;

// grow-only device buffer
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&p, want); }
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

constexpr int ROMAN_MAX_PIPELINE = 6;        // workspaces (batches in flight) a context can hold (3 serves the headline; calls of thousands of small
                                             // problems — each as long as its slowest problem — pack better at 6)

struct roman_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    int num_xcc = 8;                           // XCDs the runtime reports for this device / partition (hipDeviceAttributeNumberOfXccs)
    int wide_teams = -1;                       // team mode of the whole-device solver: -1 automatic, 0 never, 1 / 2 / 4 teams per XCD (roman_ctx_set_wide_teams)
    bool teams_launched = false;               // a launch since the last reset ran in team mode (a ROMAN_ST_INTERNAL record may be a team that could not hold its problem)
    size_t lds_max = 65536;
    bool coop_ok = false;                      // hipLaunchCooperativeKernel available (large-problem solver)
    unsigned long long spin_ticks = 400000000ull;  // bounded waits of the whole-device solver: 4 s in ticks of the device's wall clock
    hipEvent_t coopDone = nullptr;             // behind the most recent cooperative launch of this context
    bool coopIssued = false;
    std::string err;

    // Workspace: every device pool of one batch in flight, its stream and its profiling events.  With
    // roman_ctx_set_pipeline(ctx, 2) consecutive batch calls alternate between two of them (each on its own
    // internal stream), so the straggler tail of one batch's kernels overlaps the next batch's build.  Set 0 also
    // serves the stepwise API.
    struct Workspace {
        hipStream_t stream = nullptr;          // set 0 at depth 1: the context's stream; otherwise internal
        hipEvent_t done = nullptr;             // recorded after the last kernel of a batch call
        bool issued = false;
        // pools (see DESIGN.md "Data layout in HBM")
        DevBuf probs, state, totals, queue;
        DevBuf cosPool, cosDense, tabPool, qtabPool, sTmp, chunkCnt;
        DevBuf lp, li, lj, ls, ld, lza, lzb;                       // per live association, live order
        DevBuf plp, pli, plj, pls, pld, plza, plzb;                // the same in position order (stream layout)
        DevBuf rowCnt, rowPos, perm, sliceWidth, sliceBase, items, maskPool, prefPool, listPool, listOff;
        DevBuf vMu, vCu, vMun, vCun, gU, gUn, uOut, nodesOrig, nSel, widePart, wideSlots, wideBar, wideBm, wideY, wideUp, fbList;
        DevBuf cols16, cols32, vals, colsC, valsC, contSpill, contList;
        long long capMaskWords = 0, capNnz = 0, capList = 0;       // what the sparse pools hold (elements)
        // staging for the host-pointer entry points
        DevBuf hFeats, hAssoc, hU0, oAssoc, oN, oT, oStatus, oStats, hAux1, hAux2, hAux3;
        DevBuf oAll;                           // outputs of a host-output batch call as ONE block (T | stats | assoc | n | status): one copy brings it back
        // totals of the most recent batch on this workspace, copied back without waiting
        BatchTotals* pinnedTotals = nullptr;
        ProbDesc* pinnedProbs = nullptr; size_t pinnedProbsCap = 0;   // staging of the problem descriptors (truly asynchronous upload)
        hipEvent_t probsEvent = nullptr; bool probsPending = false;
        hipEvent_t totEvent = nullptr;
        bool totPending = false;
        double totMaskBound = 0.0, totSumA = 0.0, totMaxA = 0.0;   // the bounds the pending totals relate to
        unsigned totEpoch = 0;                                     // sizing-history epoch (parameter block) the pending totals were measured under
        // per-stage hipEvent pairs on `stream`
        hipEvent_t evA[ROMAN_STAGE_COUNT] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t evB[ROMAN_STAGE_COUNT] = {nullptr, nullptr, nullptr, nullptr};
        bool pending[ROMAN_STAGE_COUNT] = {false, false, false, false};
    } ws[ROMAN_MAX_PIPELINE];
    int cur = 0;                               // workspace the current call works on
    int pipeline = 1;                          // batches in flight (1..3)
    int next_ws = 0;
    hipStream_t istream[ROMAN_MAX_PIPELINE] = {};              // internal streams of the workspaces while pipelining
    int latest_ws = -1;                        // workspace of the most recent pipelined batch call
    hipEvent_t evIn = nullptr;                 // inputs ready on the caller's stream
    // roman_align_batch (host pointers): a batch of more than host_chunk problems is issued as calls of host_chunk problems with
    // host_depth of them in flight (roman_ctx_set_host_batching)
    void* hostOut = nullptr; size_t hostOutCap = 0;            // pinned landing block of the host-output entry points' single read-back
    int host_chunk = 2048, host_depth = 3;     // (config 4, 4096 pairs: 2 x 2048 take 37.7 ms, 8 x 512 41 ms — a call pays its launches and its own solver tail)

    // sizing history: largest observed need relative to what the host can bound before the launch
    struct Hist {
        bool valid = false;                    // at least one batch of this block has reported its totals
        bool tagged = false;                   // params / F below are set
        roman_params_t params; int32_t F = 0;  // the ratios belong to this parameter block
        double rMaxL = 0.0;                    // largest live set / largest association list
        double rMask = 0.0;                    // bit-matrix words / sum of nA * ceil(nA / 64)
        double rNnz = 0.0;                     // matrix slots / sum of nA
        double rList = 0.0;                    // candidate-list elements / sum of nA
        bool smallSeen = false;                // a stream-layout problem of at most SMALL_MAXL live associations has occurred
        bool largeSeen = false;                // ... one of more than SMALL_MAXL
        bool generalSeen = false;              // a problem was left to the general kernels (k_small did not finish it)
        bool cosSeen = false;                  // a batch of this block went through k_cos_sel ...
        double cosDenseFrac = 0.0;             // ... and this share of the latest such batch was left to the dense kernel (too many candidates)
        int cosSkipped = 0;                    // batches since that took the dense kernels for it (every 64th tries the screen again)
    } hist;
    long long cosScreenBatches = 0, cosDenseBatches = 0;     // batches with a cosine stage that took k_cos_sel / the dense kernels
    unsigned histEpoch = 1;                    // bumped whenever the history is reset for another parameter block
    long long skippedTotal = 0;                // problems reported ROMAN_ST_WORKSPACE so far (harvested totals)

    std::vector<std::pair<const void*, int>> ldsAttr;   // dynamic-LDS limits already set (per kernel function)

    bool profile = false;
    double prof_ms[ROMAN_STAGE_COUNT] = {0, 0, 0, 0};
    int64_t prof_n[ROMAN_STAGE_COUNT] = {0, 0, 0, 0};

    // state of the last single-problem call (stepwise API for the clipperpy shim)
    struct Last {
        bool scored = false, solved = false, dense = false, hascz = false;
        int kind = 1;                      // layout of the problem held by workspace 0 (ProbState.kind)
        DevParams D;
        ProbDesc pd;
        int32_t nA = 0, L = 0, nsel = 0;
        int64_t nnzCap = 0;
        std::vector<int32_t> assoc;        // (nA,2) host copy (explicit list) — empty for all-to-all
        std::vector<int32_t> nodes;        // selected nodes (original association indices)
        std::vector<double> u;             // length nA
        roman_stats_t stats;
        int32_t status = 0;
    } last;
};

#define WS (c->ws[c->cur])

// hipFuncAttributeMaxDynamicSharedMemorySize, raised only when a launch needs more than the function already has
static hipError_t dyn_lds(roman_ctx* c, const void* fn, size_t bytes)
{
    for (auto& e : c->ldsAttr) if (e.first == fn) {
        if ((size_t)e.second >= bytes) return hipSuccess;
        e.second = (int)bytes;
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    }
    c->ldsAttr.emplace_back(fn, (int)bytes);
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

namespace {

int fail(roman_ctx* c, int code, const char* fmt, ...)
{
    char buf[640];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_last_error = buf;
    return code;
}

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); \
    return fail((c), (e_ == hipErrorOutOfMemory) ? ROMAN_E_NOMEM : ROMAN_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

// The stepwise / host-pointer entry points run on workspace 0 and the context's own stream; work that
// pipelined batch calls still have in flight is drained first.
int use_ws0(roman_ctx* c)
{
    if (c->pipeline >= 2)
        for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) if (c->istream[k]) HIPCHK(c, hipStreamSynchronize(c->istream[k]));
    // a depth-1 batch may still be running on the context's (non-blocking) stream: the blocking copies of the stepwise
    // entry points (null stream) are not ordered against it otherwise
    if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cur = 0; c->ws[0].stream = c->stream;
    return ROMAN_OK;
}

// smallest x with sqrt(x) >= t  (so  sqrt(x) < t  <=>  x < result ; sqrt is monotone and correctly
// rounded on host and device)
double sqrt_threshold(double t)
{
    if (!(t > 0.0)) return 0.0;
    double x = t * t;
    while (std::sqrt(x) >= t) x = std::nextafter(x, 0.0);
    while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
    return x;
}

// host twins of col_pos / val_pos (kernels.hip.h)
inline size_t h_col_pos(bool quad, size_t sbase, uint32_t slot, uint32_t e) { return quad ? sbase + (size_t)(e >> 2) * 256 + slot * 4 + (e & 3u) : sbase + (size_t)e * 64 + slot; }
inline size_t h_val_pos(bool quad, size_t sbase, uint32_t slot, uint32_t e) { return quad ? sbase + (size_t)(e >> 1) * 128 + slot * 2 + (e & 1u) : sbase + (size_t)e * 64 + slot; }

int make_dev_params(roman_ctx* c, const roman_params_t* p, int32_t F, DevParams* D)
{
    if (!p) return fail(c, ROMAN_E_INVALID, "params is NULL");
    if (p->point_dim != 2 && p->point_dim != 3) return fail(c, ROMAN_E_INVALID, "point_dim must be 2 or 3 (got %d)", p->point_dim);
    if (p->invariant != ROMAN_INV_EUCLIDEAN && p->invariant != ROMAN_INV_ROMAN && p->invariant != ROMAN_INV_EUCLIDEAN_PRUNED) return fail(c, ROMAN_E_INVALID, "unknown invariant %d", p->invariant);
    if (p->ratio_feature_dim < 0 || p->ratio_feature_dim > ROMAN_MAX_RATIO_FEATURES) return fail(c, ROMAN_E_INVALID, "ratio_feature_dim out of range");
    if (p->cos_feature_dim < 0) return fail(c, ROMAN_E_INVALID, "cos_feature_dim < 0");
    if (p->drift_aware) return fail(c, ROMAN_E_UNSUPPORTED, "drift_aware is not defined by the reference call sites (always False)");
    if (p->invariant == ROMAN_INV_ROMAN && p->gravity_guided && p->point_dim != 3) return fail(c, ROMAN_E_UNSUPPORTED, "gravity_guided requires point_dim == 3");
    if (p->fusion_method < 0 || p->fusion_method > 2) return fail(c, ROMAN_E_INVALID, "unknown fusion_method");
    if (p->gravity_mode < 0 || p->gravity_mode > 2) return fail(c, ROMAN_E_INVALID, "unknown gravity_mode %d", p->gravity_mode);
    if (p->single_mode < 0 || p->single_mode > 3) return fail(c, ROMAN_E_INVALID, "unknown single_mode %d", p->single_mode);
    if (p->reserved != 0) return fail(c, ROMAN_E_INVALID, "roman_params_t.reserved must be 0");
    if (!(p->sigma > 0.0)) return fail(c, ROMAN_E_INVALID, "sigma must be > 0");
    const int need = (p->invariant != ROMAN_INV_EUCLIDEAN) ? p->point_dim + p->ratio_feature_dim + p->cos_feature_dim : p->point_dim;
    if (F < need) return fail(c, ROMAN_E_INVALID, "F=%d smaller than the %d features the invariant reads", F, need);
    memset(D, 0, sizeof(*D));
    D->p = *p;
    if (p->invariant == ROMAN_INV_EUCLIDEAN) { D->p.ratio_feature_dim = 0; D->p.cos_feature_dim = 0; D->p.gravity_guided = 0; D->p.gravity_mode = 0; D->p.single_mode = 0; }
    D->sig2 = p->sigma * p->sigma;
    D->sin_unc = std::sin(p->gravity_unc_ang_rad);
    D->x_eps = sqrt_threshold(p->epsilon);
    D->x_mindist = sqrt_threshold(p->mindist);
    if (p->invariant == ROMAN_INV_EUCLIDEAN_PRUNED) { D->p.gravity_guided = 0; D->p.gravity_mode = 0; D->p.single_mode = ROMAN_SINGLE_DIAG; }   // the pair score alone in M
    D->single = (p->invariant != ROMAN_INV_EUCLIDEAN) && (p->ratio_feature_dim > 0 || p->cos_feature_dim > 0);
    D->pruned = (p->invariant == ROMAN_INV_EUCLIDEAN_PRUNED && D->single) ? 1 : 0;
    D->gravity = (p->invariant == ROMAN_INV_ROMAN) && p->gravity_guided;
    D->gmode = D->gravity ? 1 + p->gravity_mode : 0;
    D->diag_one = D->single && (D->p.single_mode == ROMAN_SINGLE_OFFDIAG || D->pruned);
    D->keep_all = D->single && D->p.single_mode == ROMAN_SINGLE_DIAG_KEEP;
    D->F = F;
    D->stream_maxL = STREAM_MAXL;
    // k_count's prefilter: bins of epsilon / 32 (15 bits: 1023 epsilon of range, 614 m at the reference's 0.6 m; everything beyond shares the
    // last bin) — a gate-passing pair's entries are at most 32 + 1 bins apart (rounding); one bin of margin: pairs up to 1.09 epsilon apart
    // reach the exact gate (bins of epsilon / 8, the first version: 1.375 epsilon, a quarter more candidates)
    D->pre_invw = (p->epsilon > 0.0 && std::isfinite(p->epsilon)) ? 32.0 / p->epsilon : 0.0;
    D->pre_K = 34;
    D->allow_fallback = 1;
    { static const char* cooEnv = getenv("ROMAN_COO"); D->solve_flags = (cooEnv && cooEnv[0] == '0') ? 1 : 0; }   // ROMAN_COO=0: A/B switch of the one-wave solver's coordinate form
    {   // ROMAN_FILL_ROTATE=0: list order inside a row (A/B); =m: rotate a row's quads by m * row (default 5)
        static const char* rotEnv = getenv("ROMAN_FILL_ROTATE");
        const int m = rotEnv ? atoi(rotEnv) : 5;
        if (m <= 0) D->solve_flags |= 2; else D->solve_flags |= (m & 0xffff) << 8;
    }
    return ROMAN_OK;
}

void prof_flush(roman_ctx* c, int k, int s)
{
    roman_ctx::Workspace& W = c->ws[k];
    if (!W.pending[s]) return;
    float ms = 0.f;
    if (hipEventSynchronize(W.evB[s]) == hipSuccess && hipEventElapsedTime(&ms, W.evA[s], W.evB[s]) == hipSuccess) {
        c->prof_ms[s] += (double)ms; c->prof_n[s] += 1;
    }
    W.pending[s] = false;
}
struct StageTimer {
    roman_ctx* c; int s;
    StageTimer(roman_ctx* c_, int s_) : c(c_), s(s_) { if (c->profile) { prof_flush(c, c->cur, s); (void)hipEventRecord(WS.evA[s], WS.stream); } }
    void stop() { if (c->profile) { (void)hipEventRecord(WS.evB[s], WS.stream); WS.pending[s] = true; } }
};

// ROMAN_DEBUG=1: synchronise after every launch and say which one it was (locating a kernel that does not return)
int dbg_stage(roman_ctx* c, const char* what)
{
    static const bool dbg = getenv("ROMAN_DEBUG") != nullptr;
    if (!dbg) return ROMAN_OK;
    const hipError_t e = hipStreamSynchronize(WS.stream);
    fprintf(stderr, "[roman] %-14s %s\n", what, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
    return e == hipSuccess ? ROMAN_OK : fail(c, ROMAN_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
}
#define DBG(c, what) do { int rc_ = dbg_stage((c), (what)); if (rc_) return rc_; } while (0)

struct BatchIn {
    int32_t B; const double* feats; const int64_t* off1; const int32_t* n1; const int64_t* off2; const int32_t* n2;
    int32_t F; const int32_t* assoc; const int64_t* assoc_off;
};

// ---- sizing -----------------------------------------------------------------------------------------------------
// What the host can bound before the launch, and the estimates derived from it.
struct Sizing {
    double sumA = 0, maxA = 0, maskBound = 0;      // sum / max of the association list lengths, sum of nA * ceil(nA/64)
    int expectMaxL = 0;                            // estimate of the largest live set
    long long capMaskWords = 0, capNnz = 0, capList = 0;        // capacities to allocate (elements)
};

// Fold the totals a finished batch left in pinned memory into the history (never waits: only completed copies count).
void harvest_totals(roman_ctx* c, bool wait)
{
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) {
        roman_ctx::Workspace& W = c->ws[k];
        if (!W.totPending || !W.totEvent) continue;
        if (wait) { if (hipEventSynchronize(W.totEvent) != hipSuccess) { (void)hipGetLastError(); continue; } }
        else if (hipEventQuery(W.totEvent) != hipSuccess) { (void)hipGetLastError(); continue; }
        W.totPending = false;
        const BatchTotals& t = *W.pinnedTotals;
        c->skippedTotal += t.overflow;
        if (W.totEpoch != c->histEpoch) continue;               // measured under another parameter block: not this history's ratios
        roman_ctx::Hist& H = c->hist;
        if (W.totMaxA > 0) H.rMaxL = std::max(H.rMaxL, (double)t.maxL / W.totMaxA);
        if (W.totMaskBound > 0) H.rMask = std::max(H.rMask, (double)t.needMaskWords / W.totMaskBound);
        if (W.totSumA > 0) H.rNnz = std::max(H.rNnz, (double)t.needNnz / W.totSumA);
        if (W.totSumA > 0) H.rList = std::max(H.rList, (double)t.listTop / W.totSumA);
        if (t.minStreamL <= SMALL_MAXL) H.smallSeen = true;
        if (t.maxStreamL > SMALL_MAXL) H.largeSeen = true;
        if (t.nGeneral > 0) H.generalSeen = true;
        if (t.cosScreened > 0) { H.cosSeen = true; H.cosDenseFrac = (double)t.cosDense / t.cosScreened; }
        H.valid = true;
    }
}

void estimate_sizes(roman_ctx* c, const DevParams& D, const roman_params_t* params, int32_t F, const std::vector<ProbDesc>& hd, Sizing* S)
{
    roman_ctx::Hist& H = c->hist;
    if (!H.tagged || H.F != F || memcmp(&H.params, params, sizeof(roman_params_t)) != 0) {   // other parameters: other ratios
        H = roman_ctx::Hist{}; H.params = *params; H.F = F; H.tagged = true;
        ++c->histEpoch;                                         // totals still in flight belong to the old block: harvest_totals drops them
    }
    harvest_totals(c, false);
    const bool prunes = D.single && !D.keep_all;              // single scores can remove associations: L <= A, else L == A
    double heurMask = 0, heurNnz = 0; int heurMaxL = 0;
    for (const ProbDesc& d : hd) {
        const double nA = d.nA;
        S->sumA += nA; S->maxA = std::max(S->maxA, nA); S->maskBound += nA * std::ceil(nA / 64.0);
        // first-call heuristic: without single scores every association is live; with them a fraction is
        const double capL = prunes ? std::min(nA, std::max(4096.0, nA / 8.0)) : nA;
        heurMaxL = std::max(heurMaxL, (int)capL);
        heurMask += capL * std::ceil(capL / 64.0);
        heurNnz += capL * std::min(capL, 96.0);
    }
    if (H.valid) {
        S->expectMaxL = prunes ? (int)std::min(S->maxA, std::ceil(H.rMaxL * S->maxA * 1.15) + 64.0) : (int)S->maxA;
        S->capMaskWords = (long long)(prunes ? H.rMask * S->maskBound * 1.3 + 4096.0 : S->maskBound);
        S->capNnz = (long long)(H.rNnz * S->sumA * 1.3 + 65536.0);
        S->capList = (long long)(H.rList * S->sumA * 1.3 + 65536.0);
    } else {
        S->expectMaxL = heurMaxL; S->capMaskWords = (long long)heurMask; S->capNnz = (long long)heurNnz + 65536;
        S->capList = (long long)(2.5 * heurNnz) + 65536;
    }
    S->capMaskWords = std::max<long long>(S->capMaskWords, 64);
    // tests: force the first attempt of a batch to overflow (exercises the skip / retry path)
    const char* tcap = getenv("ROMAN_TEST_CAPNNZ");            // (read per call: a test sets it for some of its contexts)
    if (tcap && !H.valid) S->capNnz = atoll(tcap);
}

int may_fallback(const DevParams& D, const std::vector<ProbDesc>& hd);

struct BatchOut { int32_t kmax; int32_t* assoc_out; int32_t* n_assoc_out; double* T_out; int32_t* status_out; roman_stats_t* stats_out; };

// device pointers of a batch's outputs + the row pools the solvers' shared tail writes (sized by the association total)
int make_solve_out(roman_ctx* c, int B, int64_t sumA, const BatchOut& out, SolveOut* O)
{
    const size_t R1 = (size_t)std::max<int64_t>(sumA, 1);
    HIPCHK(c, WS.uOut.ensure(sizeof(double) * R1)); HIPCHK(c, WS.nodesOrig.ensure(sizeof(int32_t) * R1));
    HIPCHK(c, WS.nSel.ensure(sizeof(int32_t) * (size_t)B));
    O->assoc_out = out.assoc_out; O->n_assoc_out = out.n_assoc_out; O->T_out = out.T_out; O->status_out = out.status_out; O->stats_out = out.stats_out; O->kmax = out.kmax;
    O->nodesOrig = WS.nodesOrig.as<int32_t>(); O->nSel = WS.nSel.as<int32_t>(); O->uOut = WS.uOut.as<double>();
    O->dbg = nullptr;
    return ROMAN_OK;
}

// The cosines of a batch that are only used behind the gate cos > cosine_min: k_cos_sel (bf16 screen of all pairs, exact f64 for the
// candidates), then k_cos_deal for the problems it flagged (more candidates than its list holds).  `approx`: no candidates at all —
// the screen's own matrix (tests).
static hipError_t launch_cos_sel(roman_ctx* c, hipStream_t stream, const DevParams& D, int B, int maxN1, int maxN2, const ProbDesc* dP, const double* feats, double* cosPool,
                                 int32_t* dense, bool approx)
{
    hipError_t e = dyn_lds(c, reinterpret_cast<const void*>(k_cos_sel), (size_t)CSEL_LDS);
    if (e != hipSuccess) return e;
    const double thr = approx ? INFINITY : D.p.cosine_min - 0x1p-6;
    hipLaunchKernelGGL(k_cos_sel, dim3((unsigned)B), dim3(1024), (size_t)CSEL_LDS, stream, D, B, dP, feats, cosPool, dense, thr);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    constexpr int TD = 5;
    using CD = CosDeal<TD>;
    const int Gd = CD::tiles(maxN1) * CD::tiles(maxN2);
    e = dyn_lds(c, reinterpret_cast<const void*>(k_cos_deal<TD, 0>), (size_t)CD::LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_cos_deal<TD, 0>), dim3((unsigned)(B >= 8 ? Gd * ((B + 7) / 8) * 8 : Gd * B)), dim3(256), (size_t)CD::LDS, stream, D, B, Gd, dP, feats, cosPool, (const int32_t*)dense);
    return hipGetLastError();
}
// When: the score is the gated one (single scores on, not the pruned prefilter's raw products, cosine_max > cosine_min), maps of at most
// CSEL_MAXN objects, and a batch that gives most compute units a problem.  ROMAN_COS_SEL=0 never, =1 whenever the results allow it (tests).
static bool cos_sel_applies(const roman_ctx* c, const DevParams& D, int B, int maxN1, int maxN2, int64_t poolRows)
{
    if (!D.single || D.pruned || !(D.p.cosine_max > D.p.cosine_min) || !std::isfinite(D.p.cosine_min) || !std::isfinite(D.p.cosine_max)) return false;
    if (maxN1 > CSEL_MAXN || maxN2 > CSEL_MAXN || maxN1 <= 0 || maxN2 <= 0) return false;
    if (poolRows * (int64_t)D.F * 8 >= (1LL << 32)) return false;         // (the exact pass addresses a row by a 32-bit byte offset into the pool)
    const char* env = getenv("ROMAN_COS_SEL");
    if (env && env[0]) return env[0] == '1';
    // by itself: a batch that gives at least half the compute units a problem (one workgroup per problem; smaller batches keep the dense
    // kernels' many workgroups per problem), descriptors long enough for the two sweeps to pay, and no history of this parameter block
    // that says the screen rules out too little (a quarter of the latest screened batch left to the dense kernel: the dense kernel
    // directly, the screen again every 64th batch)
    if (2 * B < c->num_cu || D.p.cos_feature_dim < 64 || std::max(maxN1, maxN2) < 64) return false;
    roman_ctx::Hist& H = const_cast<roman_ctx*>(c)->hist;
    if (H.cosSeen && H.cosDenseFrac > 0.25) {
        if (++H.cosSkipped < 64) return false;
        H.cosSkipped = 0;
    }
    return true;
}

// ---- the launch sequence of one batch (score + solve), on workspace c->cur and its stream; never waits -------------

// Positions of the stream layout: rows of one degree at most for the sort-free placement (place_keys(), kernels.hip.h); beyond, the
// bitonic sort.  ROMAN_SORT_EQMAX=n sets it (0: always the sort — A/B, tests; read per call).
static int sort_eq_max()
{
    const char* e = getenv("ROMAN_SORT_EQMAX");
    return (e && e[0]) ? std::max(0, atoi(e)) : 512;
}

// Cosine matrices of B problems: k_cos_tile (64x64 tile per workgroup, operands through LDS) by default, k_cos_deal (80x80 tiles, blocks
// dealt evenly to the waves) for batches of mid-size maps, k_cos_wave (one wave per problem) for maps of at most 48 objects; ROMAN_COS=0 selects
// k_cos (32x32 tile per wave, operands from global memory), ROMAN_COS=16 / 32 the stage depth.
static hipError_t launch_cos(roman_ctx* c, hipStream_t stream, const DevParams& D, int B, int maxN1, int maxN2, const ProbDesc* dP, const double* feats, double* cosPool)
{
    static const char* env = getenv("ROMAN_COS");
    const int mode = env ? atoi(env) : 16;
    static const char* waveEnv = getenv("ROMAN_COS_WAVE");      // "0": never the one-wave-per-problem kernel (A/B)
    {   // a batch with at least two workgroups per compute unit of 5 x 5-block tiles: k_cos_deal (blocks dealt evenly to the waves).
        // ROMAN_COS_DEAL=0 never, =1 always (A/B, tests; read per call)
        const char* dealEnv = getenv("ROMAN_COS_DEAL");
        constexpr int TD = 5;
        using CD = CosDeal<TD>;
        const int Gd = CD::tiles(maxN1) * CD::tiles(maxN2);
        const bool deal = (dealEnv && dealEnv[0]) ? dealEnv[0] == '1' : (mode == 16 && (maxN1 > 16 * COSW_NB || maxN2 > 16 * COSW_NB) && (int64_t)Gd * B >= 2 * (int64_t)c->num_cu);
        if (deal) {
            const hipError_t e = dyn_lds(c, reinterpret_cast<const void*>(k_cos_deal<TD, 0>), (size_t)CD::LDS);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_cos_deal<TD, 0>), dim3((unsigned)(B >= 8 ? Gd * ((B + 7) / 8) * 8 : Gd * B)), dim3(256), (size_t)CD::LDS, stream, D, B, Gd, dP, feats, cosPool, (const int32_t*)nullptr);
            return hipGetLastError();
        }
    }
    if (mode != 0 && maxN1 <= 16 * COSW_NB && maxN2 <= 16 * COSW_NB && !(waveEnv && waveEnv[0] == '0')) {
        // the reference's demo scale: one wave per problem, no LDS, no barrier (k_cos_wave); up to two problems per compute unit (a
        // serial caller's one pair per call, a small batch): one wave per 16 x 16 block (k_cos_block: 31 against 80 us for one pair,
        // 75 against 127 for 256, 114 against 135 for 512, 196 against 152 for 1024 — tools/gpu_cos_block_sweep.py).
        // ROMAN_COS_BLOCK=0 never, =1 always (A/B, tests; read per call)
        const char* blockEnv = getenv("ROMAN_COS_BLOCK");
        const bool perBlock = (blockEnv && blockEnv[0]) ? blockEnv[0] == '1' : B <= 2 * c->num_cu;
        const int nbx = (maxN1 + 15) / 16, nby = (maxN2 + 15) / 16;
        if (perBlock) hipLaunchKernelGGL(k_cos_block, dim3((unsigned)((B * nbx * nby + 3) / 4)), dim3(256), 0, stream, D, B, nbx, nby, dP, feats, cosPool);
        else hipLaunchKernelGGL(k_cos_wave, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, D, B, dP, feats, cosPool);
        return hipGetLastError();
    }
    if (mode == 16) {
        // a few problems of larger maps (the single-pair call: 200 x 200 objects = 169 blocks): one wave per 16 x 16 block as well, eight
        // chunks of loads in flight per wave — the tile kernel below gives such a call sixteen workgroups with two stages in flight
        // (29 us; the contraction over the descriptor is serial either way).  Up to eight waves per compute unit; ROMAN_COS_BLOCK=0: never
        const char* blockEnv = getenv("ROMAN_COS_BLOCK");
        const int nbx = (maxN1 + 15) / 16, nby = (maxN2 + 15) / 16;
        if (!(blockEnv && blockEnv[0] == '0') && (int64_t)B * nbx * nby <= 8 * (int64_t)c->num_cu) {
            hipLaunchKernelGGL(k_cos_block, dim3((unsigned)((B * nbx * nby + 3) / 4)), dim3(256), 0, stream, D, B, nbx, nby, dP, feats, cosPool);
            return hipGetLastError();
        }
    }
    if (mode == 0) {
        const int tiles = ((maxN1 + COS_TILE - 1) / COS_TILE) * ((maxN2 + COS_TILE - 1) / COS_TILE), G = (tiles + 3) / 4;
        hipLaunchKernelGGL(k_cos, dim3((unsigned)(G * ((B + 7) / 8) * 8)), dim3(256), 0, stream, D, B, G, dP, feats, cosPool);
        return hipGetLastError();
    }
    auto tiles = [](int n) { return (((n + 15) >> 4) + 3) >> 2; };     // cos_tiles()
    const int G = tiles(maxN1) * tiles(maxN2);
    const int KC = mode == 32 ? 32 : 16;
    const size_t lds = (size_t)2 * 128 * (KC * 8 + 16);
    auto kf = KC == 16 ? k_cos_tile<16> : k_cos_tile<32>;
    const hipError_t e = dyn_lds(c, reinterpret_cast<const void*>(kf), lds);
    if (e != hipSuccess) return e;
    // one workgroup per tile; ROMAN_COS_WGS=n: n persistent workgroups per compute unit looping over the tiles (measured
    // slower at config 3: 376 against 324 us — the static deal balances worse than the dispatcher)
    static const char* wgEnv = getenv("ROMAN_COS_WGS");
    const int perCu = wgEnv ? atoi(wgEnv) : 0;
    const int total = G * ((B + 7) / 8) * 8;
    const int grid = perCu > 0 ? std::min(total, (c->num_cu & ~7) * perCu) : total;
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(256), lds, stream, D, B, G, dP, feats, cosPool);
    return hipGetLastError();
}

// smallOut != NULL (batch calls: score and solve in one go): problems of the reference's demo scale (<= SMALL_MAXL live
// associations) are finished by k_small right behind the live lists — tests, positions, values, solve, pose in one kernel —
// and pass the rest of the sequence by.  The stepwise entry points (roman_score, then roman_solve / the export calls) keep
// the general layout for every problem.
int enqueue_score(roman_ctx* c, const DevParams& Din, const roman_params_t* params, const BatchIn& in, std::vector<ProbDesc>& hd, DevParams* Dout,
                  const BatchOut* smallOut = nullptr, const double* smallU0 = nullptr)
{
    const int B = in.B;
    hd.assign(B, ProbDesc{});
    int64_t sumA = 0, sumCos = 0, sumTab = 0, sumQ4 = 0;
    int maxN12 = 0, maxN = 0, maxA = 0, maxN1 = 0, maxN2 = 0; int64_t maxTab = 0, poolRows = 0 /* rows of the feature pool the batch names */;
    for (int b = 0; b < B; ++b) {
        ProbDesc& d = hd[b];
        d.off1 = in.off1[b]; d.off2 = in.off2[b]; d.n1 = in.n1[b]; d.n2 = in.n2[b];
        if (d.n1 < 0 || d.n2 < 0) return fail(c, ROMAN_E_INVALID, "negative map size in problem %d", b);
        const int64_t na_list = in.assoc ? in.assoc_off[b + 1] - in.assoc_off[b] : 0;
        if (in.assoc && (na_list < 0 || na_list > 2147483647LL || in.assoc_off[b] < 0)) return fail(c, ROMAN_E_INVALID, "bad assoc_off at problem %d", b);
        if (in.assoc && na_list > 0) {
            d.assocOff = in.assoc_off[b];
            d.nA = (int32_t)na_list;
        } else {                                       // no list, or an EMPTY one: all-to-all (clipperpy replaces an empty A likewise)
            d.assocOff = -1;
            const int64_t na = (int64_t)d.n1 * d.n2;
            if (na > 2147483647LL) return fail(c, ROMAN_E_TOO_LARGE, "n1*n2 overflows int32 in problem %d", b);
            d.nA = (int32_t)na;
        }
        d.liveOff = sumA; d.cosOff = sumCos; d.tabOff = sumTab; d.normOff = 0;
        if (sumQ4 > 2147483647LL) return fail(c, ROMAN_E_TOO_LARGE, "the tables of this batch exceed the bin pool's 32-bit offsets; split it");
        d.qtabOff4 = (int32_t)sumQ4;
        sumQ4 += ((int64_t)d.n1 * ((d.n1 + 3) & ~3) + (int64_t)d.n2 * ((d.n2 + 3) & ~3)) / 4;
        maxA = std::max(maxA, d.nA);
        sumA += d.nA; sumCos += (int64_t)d.n1 * d.n2; sumTab += (int64_t)d.n1 * d.n1 + (int64_t)d.n2 * d.n2;
        maxN12 = std::max(maxN12, d.n1 + d.n2); maxN = std::max(maxN, std::max(d.n1, d.n2));
        maxN1 = std::max(maxN1, d.n1); maxN2 = std::max(maxN2, d.n2);
        maxTab = std::max(maxTab, (int64_t)d.n1 * d.n1 + (int64_t)d.n2 * d.n2);
        poolRows = std::max(poolRows, std::max(d.off1 + d.n1, d.off2 + d.n2));
    }
    if (sumA > 2000000000LL) return fail(c, ROMAN_E_TOO_LARGE, "batch has %lld associations; split it (limit 2e9 per call)", (long long)sumA);
    const size_t nA1 = (size_t)std::max<int64_t>(sumA, 1);
    DevParams D = Din;
    const bool cosOn = D.p.cos_feature_dim > 0;

    Sizing SZ;
    estimate_sizes(c, D, params, in.F, hd, &SZ);
    // a workspace never shrinks: keep what earlier (larger) batches made it hold
    SZ.capMaskWords = std::max(SZ.capMaskWords, WS.capMaskWords); SZ.capNnz = std::max(SZ.capNnz, WS.capNnz); SZ.capList = std::max(SZ.capList, WS.capList);
    // stream layout: as many live associations as the LDS tiles of this launch are sized for
    D.stream_maxL = std::min(STREAM_MAXL, std::max(64, (SZ.expectMaxL + 63) & ~63));
    { static const char* smEnv = getenv("ROMAN_STREAM_MAXL"); if (smEnv) D.stream_maxL = std::max(64, std::min(D.stream_maxL, atoi(smEnv))); }   // experiments: force the fallback layout for smaller live sets
    // The fallback layout's kernels (symmetric SELL fill, k_solve / k_solve_wide) are launched only when a problem can
    // need them: no history yet, a live set beyond the stream layout expected, or parameters only k_solve handles.  Else a
    // problem that turns out too large for the LDS tiles of this launch is skipped (ROMAN_ST_WORKSPACE) and, the
    // history corrected, takes them on its second run.
    D.allow_fallback = (!c->hist.valid || SZ.expectMaxL > D.stream_maxL || D.p.maxiniters < 1 || D.p.maxlsiters < 1) ? 1 : 0;
    {   // Which solver takes the fallback problems: few (large) ones -> k_solve_wide, every compute unit on one problem at a
        // time; many -> k_solve, one workgroup per problem.  Decided here because the fill writes 16-bit column labels for
        // the wide solver when every position fits (10 instead of 12 bytes per entry of the stream it is bound by).
        static const char* wideEnv = getenv("ROMAN_WIDE");      // "0": never, "1": whenever a fallback problem can exist
        const int mayFb = may_fallback(D, hd);
        // (many fallback problems: the teams take them when their live sets are KNOWN to be large — every association live (no single
        //  scores to remove any), or this parameter block's history says so; the first call of a gated method, whose live sets are a
        //  guess, keeps k_solve as its catch-all: no cooperative launch on the headline path)
        const bool bigLiveSets = !(D.single && !D.keep_all) || c->hist.valid;
        D.wide = (c->coop_ok && mayFb > 0 && D.p.maxiniters >= 1 && D.p.maxlsiters >= 1 && maxA <= (int64_t)WIDE_KW * c->num_cu * WIDE_NW * 64 &&
                  (wideEnv ? wideEnv[0] == '1' : (mayFb <= std::max(1, c->num_cu / 4) || (bigLiveSets && SZ.expectMaxL >= 4608)))) ? 1 : 0;   // (several: teams of compute units, one problem each;
                  // measured, solve stage: 128 x L = 3 600 k_solve 10.7 ms / teams 12.9; 128 x L = 4 900 29 / ~24; 128 x L = 6 400 41.6 / 31.3; 96 x L = 10 000 118 / 63)
        const char* i16Env = getenv("ROMAN_WIDE_IDX16");         // "0": 32-bit labels always (read per call: tests)
        D.idx16 = (D.wide && maxA <= 65534 && !(i16Env && i16Env[0] == '0')) ? 1 : 0;
    }
    *Dout = D;

    HIPCHK(c, WS.probs.ensure(sizeof(ProbDesc) * (size_t)B));
    HIPCHK(c, WS.state.ensure(sizeof(ProbState) * (size_t)B));
    HIPCHK(c, WS.totals.ensure(sizeof(BatchTotals)));
    HIPCHK(c, WS.queue.ensure(sizeof(int) * 16));              // [0..7]: the solvers' queues (k_skipped clears them), [8]: k_small's
    HIPCHK(c, WS.cosPool.ensure(sizeof(double) * (size_t)(cosOn ? std::max<int64_t>(sumCos, 1) : 1)));
    HIPCHK(c, WS.tabPool.ensure(sizeof(double) * (size_t)std::max<int64_t>(sumTab, 1)));
    HIPCHK(c, WS.sTmp.ensure(sizeof(double) * nA1));
    {
        DevBuf* i32s[] = {&WS.lp, &WS.li, &WS.lj, &WS.plp, &WS.pli, &WS.plj, &WS.rowCnt, &WS.rowPos, &WS.perm, &WS.sliceWidth, &WS.sliceBase, &WS.listOff};
        for (DevBuf* b_ : i32s) HIPCHK(c, b_->ensure(sizeof(int32_t) * nA1));
        DevBuf* f64s[] = {&WS.ls, &WS.ld, &WS.lza, &WS.lzb, &WS.pls, &WS.pld, &WS.plza, &WS.plzb};
        for (DevBuf* b_ : f64s) HIPCHK(c, b_->ensure(sizeof(double) * nA1));
    }
    HIPCHK(c, WS.maskPool.ensure(sizeof(unsigned long long) * (size_t)SZ.capMaskWords));
    HIPCHK(c, WS.listPool.ensure(sizeof(uint16_t) * (size_t)std::max<long long>(SZ.capList, 4)));
    HIPCHK(c, WS.prefPool.ensure(sizeof(uint32_t) * (size_t)SZ.capMaskWords));
    HIPCHK(c, WS.vals.ensure(sizeof(double) * (size_t)SZ.capNnz));
    HIPCHK(c, WS.cols16.ensure(sizeof(uint16_t) * (size_t)SZ.capNnz));
    HIPCHK(c, WS.cols32.ensure(sizeof(uint32_t) * (size_t)SZ.capNnz));
    WS.capMaskWords = SZ.capMaskWords; WS.capNnz = SZ.capNnz; WS.capList = SZ.capList;
    // work items: blocks of RPB consecutive live rows of one problem (more, smaller items for small batches)
    int RPB = 32;
    while (RPB < 128 && (int64_t)RPB * c->num_cu * 64 < sumA) RPB <<= 1;     // 128: ~17 items per problem balance the static item loop best (measured 32..1024)
    { const char* e_ = getenv("ROMAN_RPB"); if (e_ && e_[0]) { const int v_ = atoi(e_); if (v_ == 32 || v_ == 64 || v_ == 128) RPB = v_; } }   // (experiments: read per call)
    const size_t maxItems = (size_t)(sumA / RPB) + (size_t)B + 1;
    HIPCHK(c, WS.items.ensure(sizeof(ItemDesc) * maxItems));

    {   // descriptors: through pinned staging (a pageable source makes the runtime stage the copy itself, at several times the cost)
        if (WS.probsPending) { HIPCHK(c, hipEventSynchronize(WS.probsEvent)); WS.probsPending = false; }   // the previous upload has left the staging
        if (WS.pinnedProbsCap < (size_t)B) {
            if (WS.pinnedProbs) (void)hipHostFree(WS.pinnedProbs);
            WS.pinnedProbs = nullptr; WS.pinnedProbsCap = 0;
            const size_t cap = std::max<size_t>((size_t)B * 2, 64);
            HIPCHK(c, hipHostMalloc((void**)&WS.pinnedProbs, sizeof(ProbDesc) * cap, hipHostMallocDefault));
            WS.pinnedProbsCap = cap;
        }
        memcpy(WS.pinnedProbs, hd.data(), sizeof(ProbDesc) * (size_t)B);
        HIPCHK(c, hipMemcpyAsync(WS.probs.p, WS.pinnedProbs, sizeof(ProbDesc) * (size_t)B, hipMemcpyHostToDevice, WS.stream));
        HIPCHK(c, hipEventRecord(WS.probsEvent, WS.stream));
        WS.probsPending = true;
    }
    const ProbDesc* dP = WS.probs.as<ProbDesc>();
    ProbState* dS = WS.state.as<ProbState>();
    BatchTotals* dT = WS.totals.as<BatchTotals>();
    const LivePools LP{WS.lp.as<int32_t>(), WS.li.as<int32_t>(), WS.lj.as<int32_t>(), WS.ls.as<double>(), WS.ld.as<double>(), WS.lza.as<double>(), WS.lzb.as<double>()};
    const LivePools PP{WS.plp.as<int32_t>(), WS.pli.as<int32_t>(), WS.plj.as<int32_t>(), WS.pls.as<double>(), WS.pld.as<double>(), WS.plza.as<double>(), WS.plzb.as<double>()};

    StageTimer t0(c, ROMAN_STAGE_SINGLE);
    bool cosScreen = false;
    if (cosOn && maxN1 > 0 && maxN2 > 0) {
        cosScreen = cos_sel_applies(c, D, B, maxN1, maxN2, poolRows);
        ++(cosScreen ? c->cosScreenBatches : c->cosDenseBatches);
        if (cosScreen) {
            HIPCHK(c, WS.cosDense.ensure(sizeof(int32_t) * (size_t)B));
            HIPCHK(c, launch_cos_sel(c, WS.stream, D, B, maxN1, maxN2, dP, in.feats, WS.cosPool.as<double>(), WS.cosDense.as<int32_t>(), false));
        } else
        HIPCHK(c, launch_cos(c, WS.stream, D, B, maxN1, maxN2, dP, in.feats, WS.cosPool.as<double>()));
    DBG(c, "k_cos");
    }
    if (maxTab > 0) {
        const size_t tabLds = sizeof(double) * 3 * (size_t)std::max(maxN, 1);
        if (tabLds > c->lds_max) return fail(c, ROMAN_E_TOO_LARGE, "maps of %d objects exceed the LDS point staging of this build", maxN);
        HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(k_tables), tabLds));
        // bands: enough workgroups to fill the device twice, no more (every band stages the whole map's points)
        const int bands = std::max(1, std::min((maxN + 15) / 16, (2 * c->num_cu + 2 * B - 1) / (2 * B)));
        const int RBt = (((maxN + bands - 1) / bands) + 15) & ~15;
        const int thr = RBt >= 64 ? 1024 : 256;
        // (the same entries as 15-bit bins for k_count's prefilter: 2 more bytes per entry)
        uint16_t* qtab = nullptr;
        if (D.pre_invw > 0.0 && std::isfinite(D.pre_invw)) {
            HIPCHK(c, WS.qtabPool.ensure(sizeof(uint16_t) * 4 * (size_t)std::max<int64_t>(sumQ4, 1) + 64));
            qtab = WS.qtabPool.as<uint16_t>();
        }
        hipLaunchKernelGGL(k_tables, dim3((unsigned)((maxN + RBt - 1) / RBt), 2, B), dim3(thr), tabLds, WS.stream, D, dP, in.feats, WS.tabPool.as<double>(), RBt, qtab);
    DBG(c, "k_tables");
    }
    {   // single scores, then the ordered compaction of the live associations: chunks x problems
        const int maxChunks = std::max(1, (maxA + LIVE_CHUNK - 1) / LIVE_CHUNK);
        HIPCHK(c, WS.chunkCnt.ensure(sizeof(int32_t) * (size_t)B * (size_t)maxChunks * 4));      // live associations per wave segment
        hipLaunchKernelGGL(k_live<0>, dim3((unsigned)maxChunks, (unsigned)B), dim3(256), 0, WS.stream, D, dP, dS, in.feats, in.assoc, WS.cosPool.as<double>(), WS.sTmp.as<double>(), PP.lp /* scratch until k_upper */,
                           WS.chunkCnt.as<int32_t>(), maxChunks, LP.lp, LP.li, LP.lj, LP.ls, LP.ld, LP.lza, LP.lzb);
        hipLaunchKernelGGL(k_live<1>, dim3((unsigned)maxChunks, (unsigned)B), dim3(256), 0, WS.stream, D, dP, dS, in.feats, in.assoc, WS.cosPool.as<double>(), WS.sTmp.as<double>(), PP.lp /* scratch until k_upper */,
                           WS.chunkCnt.as<int32_t>(), maxChunks, LP.lp, LP.li, LP.lj, LP.ls, LP.ld, LP.lza, LP.lzb);
    DBG(c, "k_live");
    }
    t0.stop();

    // pair-test kernel LDS: a column tile (objects [+ z] of every live association) + per wave the table rows of
    // the NR rows it sweeps together (NR = 2 if that fits beside the whole column tile, else 1).  The tile is sized
    // for the EXPECTED largest live set; a problem that exceeds it reads its columns from memory instead.
    const int expL = std::max(SZ.expectMaxL, 1);
    const int ldsPerRow = ((2 * std::max(maxN, 1) + 1 + 1) & ~1) + 2;     // n1 + sentinel + n2 doubles
    const int colBytesC = D.gravity ? 20 : 4;                             // [z pair] + packed 16-bit table-slice indices
    const int Lneed = (expL + 255) & ~255;
    // (two rows per wave when their table slices leave room for the whole expected live set — or, for live sets beyond any
    //  tile, for a tile of at least 2048 columns: k_count sweeps larger live sets tile by tile)
    int NRc = ((size_t)16 * 2 * ldsPerRow * sizeof(double) + (size_t)std::min(Lneed, 2048) * colBytesC <= c->lds_max) ? 2 : 1;
    int wpb = 16;
    static const char* nrEnv = getenv("ROMAN_COUNT_NR");
    static const char* wpbEnv = getenv("ROMAN_COUNT_WPB");
    if (nrEnv) NRc = atoi(nrEnv) == 2 ? 2 : 1;
    if (wpbEnv) wpb = std::max(1, std::min(16, atoi(wpbEnv)));
    while (wpb > 1 && (size_t)wpb * NRc * ldsPerRow * sizeof(double) + 256 * colBytesC > c->lds_max) wpb >>= 1;
    if ((size_t)wpb * NRc * ldsPerRow * sizeof(double) + 256 * colBytesC > c->lds_max)
        return fail(c, ROMAN_E_TOO_LARGE, "maps of %d objects exceed the LDS table staging of this build", maxN);
    const int ldsPerWave = NRc * ldsPerRow;
    const size_t tabLds = (size_t)wpb * ldsPerWave * sizeof(double);
    int TCc = (int)std::min<size_t>((c->lds_max - tabLds) / colBytesC, 32768) & ~255;
    TCc = std::min(TCc, Lneed);
    const size_t pairLds = tabLds + (size_t)TCc * colBytesC;
    const int pairGrid = c->num_cu * std::max(1, std::min(2048 / (wpb * 64), (int)(c->lds_max / pairLds)));
    // The one-tile sweep with candidate generation in front of the exact gate (count_rows_pre, round 6): column tile without the z
    // pairs (the exact gate reads them from memory for the ~6 % of the columns that reach it) + per wave the table slices, their
    // 16-bit bin slices, the candidate queue and the rows' mask words.  Taken when the whole expected live set fits its tile;
    // ROMAN_COUNT_PRE=0 / 1 in the environment forces the plain sweep / the prefilter where it fits (A/B and tests: read per call).
    int preNR = 0, preWpb = 0, preTC = 0, preNO = 0; size_t preLds = 0;
    const int preRow = (2 * std::max(maxN, 1) + 1 + 3) & ~3;                 // entries of the packed bin table (n1 + sentinel + n2)
    {
        const char* preEnv = getenv("ROMAN_COUNT_PRE");
        const bool want = (preEnv ? preEnv[0] != '0' : true) && D.pre_invw > 0.0 && std::isfinite(D.pre_invw) && Lneed <= (preEnv ? 16000 : 4096) && maxN <= 32767;   // (live sets beyond the stream layout's: measured slower there — 64 x L = 10 000: 5.8 against 4.0 ms with the tiled plain sweep; forced on by ROMAN_COUNT_PRE=1 up to 16 000)
        if (want) {
            const int tc = Lneed;
            for (int wp = PRE_WAVES; wp >= 8; wp -= 4) {
                const size_t need = (size_t)(tc + PRE_COLPAD + 4) * 4 + (size_t)wp * (size_t)count_pre_wave_bytes(preRow, tc);
                if (need <= c->lds_max) { preNR = PRE_NR; preWpb = wp; preTC = tc; preLds = need; break; }
            }
            if (wpbEnv && preNR) {                               // (A/B: ROMAN_COUNT_WPB also caps the prefiltered sweep's waves)
                preWpb = std::min(preWpb, wpb);
                preLds = (size_t)(preTC + PRE_COLPAD + 4) * 4 + (size_t)preWpb * (size_t)count_pre_wave_bytes(preRow, preTC);
            }
            // ROMAN_COUNT_OBJ=1 (A/B and tests; read per call): the exact gate recomputes its two distances from the objects' coordinates in
            // LDS (32 bytes per object) instead of reading the tables and the heights from memory — four L2 gathers per candidate fewer,
            // bit-identical masks (146 parity tests either way), and SLOWER: 397 against 373 us per batch of 256 on one box, twice: the
            // sweep is bound by its LDS, which the eight extra reads per candidate load further; sixteen waves hide the L2 round trips.  Off.
            if (preNR) {
                const char* objEnv = getenv("ROMAN_COUNT_OBJ");
                const int no = (std::max(maxN, 1) + 1) & ~1;
                const size_t objBytes = (size_t)2 * no * 32 + 16;
                if (objEnv && objEnv[0] == '1' && preLds + (size_t)preTC * 4 + objBytes <= c->lds_max) { preNO = no; preLds += objBytes; }
            }
        }
    }

    StageTimer t1(c, ROMAN_STAGE_COUNT_PASS);                  // (the stage also holds k_small: at the demo scale it IS the rest of the alignment)
    {   // demo-scale problems: finished here, in one kernel (k_small); launched when such problems have been seen with this
        // parameter block (or, with no history yet, when the association lists are short enough to make them likely)
        static const char* fusedEnv = getenv("ROMAN_SMALL_FUSED");     // "0": never (A/B)
        const bool want = smallOut != nullptr && !(fusedEnv && fusedEnv[0] == '0') && D.p.maxiniters >= 1 && D.p.maxlsiters >= 1 && maxTab > 0 &&
                          (c->hist.valid ? c->hist.smallSeen : maxA <= 16 * SMALL_MAXL);
        if (want) {
            SolveOut O;
            int rc = make_solve_out(c, B, sumA, *smallOut, &O);
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(WS.queue.as<int>() + 8, 0, sizeof(int), WS.stream));
            const bool fast = D.single && (D.p.single_mode == ROMAN_SINGLE_BOTH || D.p.single_mode == ROMAN_SINGLE_OFFDIAG) && D.p.distance_weight == 1.0 &&
                              D.p.fusion_method != ROMAN_FUSE_ARITHMETIC_MEAN && D.p.fusion_method != ROMAN_FUSE_PRODUCT;      // (k_fill_list's own choice)
            constexpr size_t ldsSmall = small_lds_bytes();     // 3 vectors of 192 + reduction scratch + coordinate list + the problem's columns (~14 KB)
            const int grid = std::max(1, std::min(B, c->num_cu * 12));
            auto ks = fast ? k_small<true> : k_small<false>;
            hipLaunchKernelGGL(ks, dim3(grid), dim3(64), ldsSmall, WS.stream, D, B, dP, dS, in.feats, in.assoc, WS.tabPool.as<double>(),
                               LP.lp, LP.li, LP.lj, LP.ls, LP.ld, LP.lza, LP.lzb, WS.plp.as<int32_t>(), WS.pld.as<double>(), WS.rowPos.as<uint32_t>(),
                               smallU0, O, WS.queue.as<int>() + 8);
    DBG(c, "k_small");
            // Every problem of this parameter block has so far been finished by k_small: the general kernels (twelve launches
            // that would find nothing to do: ~0.25 ms per call of 4096 problems) are left out.  A problem k_small leaves behind
            // is then skipped like a workspace overflow (ROMAN_ST_WORKSPACE) and the history remembers that they are needed.
            static const char* onlyEnv = getenv("ROMAN_SMALL_ONLY");   // "0": never leave them out
            D.small_only = (c->hist.valid && !c->hist.generalSeen && !(onlyEnv && onlyEnv[0] == '0')) ? 1 : 0;
            Dout->small_only = D.small_only;
        }
    }
    hipLaunchKernelGGL(k_rowbase, dim3(1), dim3(B > 256 ? 1024 : 256), 0, WS.stream, B, RPB, SZ.capMaskWords, dS, dT, D.small_only, cosScreen ? (const int32_t*)WS.cosDense.as<int32_t>() : (const int32_t*)nullptr);
    DBG(c, "k_rowbase");
    if (!D.small_only) {
    hipLaunchKernelGGL(k_items, dim3(B), dim3(256), 0, WS.stream, RPB, dS, WS.items.as<ItemDesc>());
    DBG(c, "k_items");
    }
    if (sumA > 0 && !D.small_only) {
        // two instantiations over the same items: the one-tile sweep for live sets that fit the LDS column tile, and — launched
        // only when such problems can exist (fallback problems, or a tile capped by the LDS) — the tile-by-tile sweep for the rest
        const bool needTiled = D.allow_fallback || TCc < Lneed;
        int degGiven = 0;
        for (int tiled = 0; tiled < (needTiled ? 2 : 1); ++tiled) {
            const void* kc = nullptr;
#define ROMAN_KC(GM_) kc = tiled ? (NRc == 2 ? reinterpret_cast<const void*>(k_count<GM_, 2, true>) : reinterpret_cast<const void*>(k_count<GM_, 1, true>)) \
                                 : (NRc == 2 ? reinterpret_cast<const void*>(k_count<GM_, 2, false>) : reinterpret_cast<const void*>(k_count<GM_, 1, false>))
            switch (D.gmode) { case 1: ROMAN_KC(1); break; case 2: ROMAN_KC(2); break; case 3: ROMAN_KC(3); break; default: ROMAN_KC(0); break; }
#undef ROMAN_KC
            const bool usePre = !tiled && preNR > 0;
            if (usePre) {
#define ROMAN_KP(GM_) kc = reinterpret_cast<const void*>(k_count<GM_, 1, false, true>)
                switch (D.gmode) { case 1: ROMAN_KP(1); break; case 2: ROMAN_KP(2); break; case 3: ROMAN_KP(3); break; default: ROMAN_KP(0); break; }
#undef ROMAN_KP
            }
            const size_t ldsK = usePre ? preLds : pairLds;
            const int wpbK = usePre ? preWpb : wpb;
            const int gridK = usePre ? c->num_cu * std::max(1, std::min(2048 / (wpbK * 64), (int)(c->lds_max / ldsK))) : pairGrid;
            // (the prefiltered sweep of a batch with at least a problem per compute unit takes whole problems as work items: RPB = -B;
            //  ROMAN_COUNT_WHOLE=0 / 1 forces the row blocks / whole problems; + the rows' degrees in LDS then)
            const char* wholeEnv = getenv("ROMAN_COUNT_WHOLE");
            const bool wholeP = usePre && (wholeEnv ? wholeEnv[0] != '0' : B >= c->num_cu) && preLds + (size_t)preTC * 4 <= c->lds_max;
            if (wholeP) degGiven = 1;                            // k_count counts the degrees as the pairs pass: k_lists skips its degree sweep
            HIPCHK(c, dyn_lds(c, kc, ldsK + (wholeP ? (size_t)preTC * 4 : 0)));
            DevParams a_D = D; const ProbDesc* a_dP = dP; const ProbState* a_dS = dS; const BatchTotals* a_dT = dT; const ItemDesc* a_items = WS.items.as<ItemDesc>();
            const double* a_tab = WS.tabPool.as<double>(); const int32_t* a_li = LP.li; const int32_t* a_lj = LP.lj; const double* a_za = LP.lza; const double* a_zb = LP.lzb;
            uint32_t* a_rc = WS.rowCnt.as<uint32_t>(); unsigned long long* a_mask = WS.maskPool.as<unsigned long long>(); uint32_t* a_pref = WS.prefPool.as<uint32_t>();
            int a_TC = usePre ? preTC : TCc, a_lpw = usePre ? preRow : ldsPerWave;
            int a_RPB = wholeP ? -B : RPB;
            const uint16_t* a_qtab = WS.qtabPool.as<uint16_t>();
            const double* a_feats = in.feats; int a_NO = usePre ? preNO : 0;
            void* args[] = {&a_D, &a_dP, &a_dS, &a_dT, &a_items, &a_tab, &a_li, &a_lj, &a_za, &a_zb, &a_rc, &a_mask, &a_pref, &a_TC, &a_lpw, &a_RPB, &a_qtab, &a_feats, &a_NO};
            HIPCHK(c, hipLaunchKernel(kc, dim3(gridK), dim3(wpbK * 64), args, ldsK + (wholeP ? (size_t)preTC * 4 : 0), WS.stream));
        }
    DBG(c, "k_count");
        // Stream-layout problems (kind 0) go from the upper blocks straight to positions and kept-candidate lists in ONE kernel, one
        // workgroup per problem (k_lists, round 5); the symmetric matrix and its four kernels remain for the fallback layout's
        // problems — launched only when such problems can occur — and, with ROMAN_LISTS=0 in the environment, for everything (A/B).
        // (a handful of problems — the single-pair call — keep the four kernels: they spread ONE problem over the device, k_lists
        //  gives it one compute unit: 0.59 against 0.66 ms for B = 1)
        const char* listsEnv = getenv("ROMAN_LISTS");           // "0": never, "1": always (A/B and tests: read per call)
        const int fusedLists = listsEnv ? (listsEnv[0] == '0' ? 0 : 1) : (B >= std::max(8, c->num_cu / 8) ? 1 : 0);
        if (fusedLists) {
            // ROMAN_LISTS_LDS=n: n bytes of unused dynamic LDS on top of the kernel's ~60 KB (above 20 KB a compute unit holds ONE workgroup).
            // Probe of round 6: behind the prefiltered k_count with whole problems as work items k_lists takes 162-165 us in some
            // processes and 192-210 us in others (every launch of a process alike; identical instruction and byte counters,
            // SQ_WAIT_INST_ANY 122 M -> 190-220 M wave-cycles; behind the plain sweep or row-block work items always 160-169 us).  NOT the
            // cause, each measured with a throw-away build (tools/r6_lists_probe.sh is the per-process probe): two workgroups on one compute unit (this pad), which workgroup takes which
            // problem (rotations by 1, 4, 8, 128), dirty mask lines in L2 (non-temporal stores: -3 us).  Open.
            const char* llEnv = getenv("ROMAN_LISTS_LDS");
            const size_t listsPad = llEnv ? (size_t)std::max(0, atoi(llEnv)) : 0;
            if (listsPad) HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(k_lists), listsPad));
            hipLaunchKernelGGL(k_lists, dim3((unsigned)std::max(1, std::min(B, 2 * c->num_cu))), dim3(LISTS_NT), listsPad, WS.stream, B, dP, dS, dT,
                               WS.maskPool.as<unsigned long long>(), WS.listPool.as<uint16_t>(), WS.listOff.as<uint32_t>(),
                               WS.rowCnt.as<uint32_t>(), WS.perm.as<uint32_t>(), WS.rowPos.as<uint32_t>(), LP, PP, (long long)SZ.capList, sort_eq_max(), degGiven);
    DBG(c, "k_lists");
        }
        if (!fusedLists || D.allow_fallback) {
        {   // lower triangle of the bit matrices = transposed blocks of the upper triangle (grid for the expected size;
            // the kernel loops when a problem is larger)
            const int Wexp = (expL + 63) / 64;
            const int tasks = ((Wexp + 7) / 8) * ((std::max(Wexp - 1, 1) + 7) / 8);          // workgroups of 8 waves
            const int T = std::max(tasks, 1);
            hipLaunchKernelGGL(k_mirror, dim3((unsigned)(T * ((B + 7) / 8) * 8)), dim3(512), 0, WS.stream, B, T, dS, WS.maskPool.as<unsigned long long>(), fusedLists);
    DBG(c, "k_mirror");
        }
        hipLaunchKernelGGL(k_rowprefix, dim3(c->num_cu * 2), dim3(1024), 0, WS.stream, dP, dS, dT, WS.items.as<ItemDesc>(),
                           WS.maskPool.as<unsigned long long>(), WS.prefPool.as<uint32_t>(), WS.rowCnt.as<uint32_t>(), RPB, fusedLists);
    DBG(c, "k_rowprefix");
        // the stream layout's bitonic sort takes N/2 threads for N = 2^k >= L keys; the fallback layout's counting sort is written
        // for 1024 (at the reference's demo scale — 60 live associations — 4096 workgroups of 1024 threads were 99 us of a 1.4 ms call)
        int sortThr = 1024;
        if (!D.allow_fallback) { int N2 = 64; while (N2 < expL) N2 <<= 1; sortThr = std::max(64, std::min(1024, N2 / 2)); }
        hipLaunchKernelGGL(k_rowsort, dim3(B), dim3(sortThr), 0, WS.stream, dP, dS, dT, WS.rowCnt.as<uint32_t>(), WS.rowPos.as<uint32_t>(), WS.perm.as<uint32_t>(),
                           WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.listOff.as<uint32_t>(), SZ.capList, fusedLists, sort_eq_max());
    DBG(c, "k_rowsort");
        if (!fusedLists) {
        // small live sets (the reference's demo scale): a work item is a whole problem of a few dozen rows — more, smaller
        // workgroups keep more of them in flight (a row is a chain of dependent memory round trips)
        const int upThr = expL <= 256 ? 256 : 1024;
        hipLaunchKernelGGL(k_upper, dim3(c->num_cu * 2 * (1024 / upThr)), dim3(upThr), 0, WS.stream, dP, dS, dT, WS.items.as<ItemDesc>(),
                           WS.maskPool.as<unsigned long long>(), WS.listPool.as<uint16_t>(), WS.listOff.as<uint32_t>(),
                           WS.rowCnt.as<uint32_t>(), WS.perm.as<uint32_t>(), WS.rowPos.as<uint32_t>(), LP, PP, RPB);
    DBG(c, "k_upper");
        }
        }
        hipLaunchKernelGGL(k_slicegeom, dim3(B), dim3(64), 0, WS.stream, dP, dS, WS.rowCnt.as<uint32_t>(), WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>());
    DBG(c, "k_slicegeom");
    }
    // k_fill_list work items: groups of consecutive slices of one problem, about 3 per CU for the whole batch
    // (about 3 groups per CU, EVERY problem cut into the same number NG of groups: the static deal of the groups to the
    // workgroups then gives each one heavy (first slices) and two light groups; whole problems per workgroup or cuts
    // that differ between problems measured 0.46-0.63 ms against 0.38-0.40)
    static const char* ngEnv = getenv("ROMAN_FILL_NG");
    int NG = (int)std::max<int64_t>(1, std::min<int64_t>(FILLS_MAXSPI, (3 * (int64_t)c->num_cu + B / 2) / std::max(B, 1)));
    {   // a workgroup takes every (grid/8)-th group of its XCD's range: the residues it meets (which part of a problem
        // a group is) rotate through all NG values only if that stride is coprime to NG — NG = 4 on 256 CUs (stride 32)
        // hands some workgroups nothing but first, heavy groups (fill 0.59 ms against 0.38 with NG = 3)
        const int stride = std::max(1, (c->num_cu & ~7) / 8);
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        while (NG > 1 && gcd(stride % NG, NG) != 1) --NG;
    }
    if (ngEnv) NG = std::max(1, std::min(FILLS_MAXSPI, atoi(ngEnv)));
    const int Wmax = std::max(1, D.stream_maxL / 64);
    const int SPI = (Wmax + std::min(NG, Wmax) - 1) / std::min(NG, Wmax);       // slices per group at most (LDS capacity)
    if (!D.small_only) {
    hipLaunchKernelGGL(k_probscan, dim3(1), dim3(B > 256 ? 1024 : 256), 0, WS.stream, B, NG, SZ.capNnz, dS, dT);
    DBG(c, "k_probscan");
    }
    t1.stop();

    StageTimer t2(c, ROMAN_STAGE_FILL);
    if (sumA > 0 && !D.small_only) {
        // list fill (stream layout): column tile + the rows and slice tables of one group
        const int colBytesF = D.gravity ? 32 : 16;
        const int TCs = D.stream_maxL;
        // the objects' coordinates in LDS (32 bytes per object) instead of the z columns, when they fit: the fill then
        // recomputes the distances instead of gathering them from the tables
        static const char* objEnv = getenv("ROMAN_FILL_OBJ");
        const int NO = (std::max(maxN, 1) + 1) & ~1;
        const size_t groupLds = sizeof(uint32_t) * (size_t)(3 * SPI * 64 + 2 * (SPI + 1) + 2);
        const bool obj = (objEnv ? atoi(objEnv) != 0 : true) && (size_t)TCs * (16 + 2) + (size_t)64 * NO + groupLds <= c->lds_max;
        const size_t sliceLds = obj ? (size_t)TCs * (16 + 2) + (size_t)64 * NO + groupLds : (size_t)TCs * (colBytesF + 2) + groupLds;
        if (sliceLds > c->lds_max) return fail(c, ROMAN_E_TOO_LARGE, "internal: stream column tile does not fit the LDS");
        const bool fast = D.single && (D.p.single_mode == ROMAN_SINGLE_BOTH || D.p.single_mode == ROMAN_SINGLE_OFFDIAG) && D.p.distance_weight == 1.0 &&
                          D.p.fusion_method != ROMAN_FUSE_ARITHMETIC_MEAN && D.p.fusion_method != ROMAN_FUSE_PRODUCT;
        auto kf = D.gravity ? (fast ? k_fill_list<true, true, false> : k_fill_list<true, false, false>) : (fast ? k_fill_list<false, true, false> : k_fill_list<false, false, false>);
        if (obj) kf = D.gravity ? (fast ? k_fill_list<true, true, true> : k_fill_list<true, false, true>) : (fast ? k_fill_list<false, true, true> : k_fill_list<false, false, true>);
        HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(kf), sliceLds));
        const int fillThr = expL <= 256 ? 256 : 1024;          // (k_upper: small problems, small workgroups, more of them)
        const int fillWgs = std::max(1, std::min(1024 / fillThr, (int)(c->lds_max / sliceLds)));
        hipLaunchKernelGGL(kf, dim3((unsigned)((c->num_cu & ~7) * fillWgs)), dim3(fillThr), sliceLds, WS.stream,
                           D, B, dP, dS, dT, WS.tabPool.as<double>(), in.feats, NO, LP.li, LP.lj, LP.ls, LP.lza, LP.lzb,
                           WS.listPool.as<uint16_t>(), WS.listOff.as<uint32_t>(), WS.rowCnt.as<uint32_t>(), WS.perm.as<uint32_t>(), WS.rowPos.as<uint32_t>(),
                           WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.cols16.as<uint16_t>(), WS.vals.as<double>(), TCs, NG, SPI);
    DBG(c, "k_fill_list");
        // fallback layout (symmetric SELL-64, 32-bit indices) for the problems the stream layout does not take: only when
        // one can exist (the kernel would find no work otherwise)
        if (D.allow_fallback && (SZ.maxA > D.stream_maxL || D.p.maxiniters < 1 || D.p.maxlsiters < 1)) {
            const int colBytesG = D.gravity ? 40 : 24;
            const size_t ringLds = (size_t)16 * 3 * FILL_Q * sizeof(uint32_t) + (size_t)FILL_RPB * FILL_ROWBYTES;     // per-wave rings + the item's rows
            static_assert(FILL_RPB >= 128, "the work items of a batch are blocks of at most 128 rows");
            int TCf = (int)std::min<size_t>((c->lds_max - ringLds) / colBytesG, 32768) & ~63;
            TCf = std::min(TCf, std::max(64, (Lneed + 63) & ~63));      // (a multiple of 64, at least 64: k_fill cuts larger live sets into windows of TCf columns)
            const size_t fillLds = ringLds + (size_t)TCf * colBytesG;
            const int fillGrid = c->num_cu * std::max(1, std::min(2, (int)(c->lds_max / fillLds)));
            if (D.idx16) {                                      // 16-bit column labels for k_solve_wide
                auto kg = D.gravity ? k_fill<true, uint16_t, true> : k_fill<false, uint16_t, true>;
                HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(kg), fillLds));
                hipLaunchKernelGGL(kg, dim3(fillGrid), dim3(1024), fillLds, WS.stream, D, dP, dS, dT, WS.items.as<ItemDesc>(), WS.tabPool.as<double>(),
                                   LP.li, LP.lj, LP.ls, LP.lza, LP.lzb,
                                   WS.rowCnt.as<uint32_t>(), WS.maskPool.as<unsigned long long>(), WS.prefPool.as<uint32_t>(), WS.rowPos.as<uint32_t>(),
                                   WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.cols16.as<uint16_t>(), WS.vals.as<double>(), TCf, RPB, (const uint32_t*)WS.perm.as<uint32_t>());
            } else {
                auto kg = D.gravity ? k_fill<true, uint32_t, true> : k_fill<false, uint32_t, true>;
                HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(kg), fillLds));
                hipLaunchKernelGGL(kg, dim3(fillGrid), dim3(1024), fillLds, WS.stream, D, dP, dS, dT, WS.items.as<ItemDesc>(), WS.tabPool.as<double>(),
                                   LP.li, LP.lj, LP.ls, LP.lza, LP.lzb,
                                   WS.rowCnt.as<uint32_t>(), WS.maskPool.as<unsigned long long>(), WS.prefPool.as<uint32_t>(), WS.rowPos.as<uint32_t>(),
                                   WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.cols32.as<uint32_t>(), WS.vals.as<double>(), TCf, RPB, (const uint32_t*)WS.perm.as<uint32_t>());
            }
    DBG(c, "k_fill");
        }
    }
    t2.stop();
    // the totals travel back on their own: whoever sizes a later batch picks them up once they have arrived
    HIPCHK(c, hipMemcpyAsync(WS.pinnedTotals, dT, sizeof(BatchTotals), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipEventRecord(WS.totEvent, WS.stream));
    WS.totPending = true; WS.totMaskBound = SZ.maskBound; WS.totSumA = SZ.sumA; WS.totMaxA = SZ.maxA; WS.totEpoch = c->histEpoch;
    HIPCHK(c, hipGetLastError());
    return ROMAN_OK;
}

// Solver + rounding + pose on the matrices held by the workspace.  `feats` may be NULL (dense
// matrix problems have no points: the pose is skipped).  mayFallback: a problem of the fallback kind can exist.
int enqueue_solve(roman_ctx* c, const DevParams& D, int B, int64_t sumA, int64_t maxA /* longest association list */, const double* feats, const int32_t* assoc,
                  const double* u0, bool hascz, int mayFallback /* problems that can be of the fallback kind */, const BatchOut& out)
{
    const size_t R1 = (size_t)std::max<int64_t>(sumA, 1);
    SolveOut O;
    { int rc_ = make_solve_out(c, B, sumA, out, &O); if (rc_) return rc_; }
#ifdef ROMAN_SOLVE_TIMING
    HIPCHK(c, WS.hAux3.ensure(sizeof(unsigned long long) * 16 * (size_t)B));
    HIPCHK(c, hipMemsetAsync(WS.hAux3.p, 0, sizeof(unsigned long long) * 16 * (size_t)B, WS.stream));
    O.dbg = WS.hAux3.as<unsigned long long>();
#endif
    // stream solver: LDS = three vectors of Lc elements (Lc: whole slices of stream_maxL + the 64 dummy elements the
    // inert padding entries point at) + reduction scratch + slice table
    constexpr int NW = ROMAN_SOLVE_WAVES;
    const int Lc = STREAM_MAXL + 64;                            // (fixed: the kernel addresses the three vectors at compile-time distances)
    const size_t ldsUp = (size_t)3 * 8 * Lc + sizeof(double) * red_doubles(NW) + sizeof(uint32_t) * (ST_MAXSL + 2) + sizeof(int) * 8 + 16;
    const int wgPerCu = std::max(1, std::min((int)(c->lds_max / ldsUp), 2048 / (NW * 64)));
    const int gridUp = std::max(1, std::min(B, c->num_cu * wgPerCu));

    // Bounded launches (SolveCont, kernels.hip.h): a problem still iterating after `cap` passes of the launch is suspended and a second
    // launch resumes the suspended ones, all at once, one workgroup each.  Built for the round-5 review's item 2 (a call with more
    // problems than workgroups hands them out from a queue, and a 400-pass problem claimed late was thought to hold the tail), correct
    // bit for bit (tests/test_gpu_batch.py::test_bounded_solver_launches_...), and MEASURED WITHOUT EFFECT (round 6, one box, budget 64
    // against none, alternating): a rank's 512-pair share of the config-4 grid 4.4 ... 6.5 ms either way, the slowest rank 6.5 ms, one
    // shot of the 4096-pair grid 32-33 ms, headline 157-160 k alignments/s.  The tail IS the long problem's own passes — 364 x ~10 us on
    // its one compute unit, whenever they start —, not its place in the queue.  OFF unless ROMAN_SOLVE_CAP=n asks for a budget of n
    // passes (read per call).
    SolveCont cont{}; SolveCont contResume{};
    {
        const char* capEnv = getenv("ROMAN_SOLVE_CAP");
        const int cap = capEnv ? std::max(0, atoi(capEnv)) : 0;
        if (cap > 0 && !D.small_only) {
            const int slots = std::max(64, std::min(B, 1024));
            const int maxL = std::max(64, (D.stream_maxL + 63) & ~63);
            const int slotDoubles = 16 + 3 * maxL;
            HIPCHK(c, WS.contSpill.ensure(sizeof(double) * (size_t)slots * (size_t)slotDoubles));
            HIPCHK(c, WS.contList.ensure(sizeof(int32_t) * (size_t)slots));
            cont.spill = WS.contSpill.as<double>(); cont.list = WS.contList.as<int32_t>(); cont.counters = WS.queue.as<int>() + 10;
            cont.cap = cap; cont.slots = slots; cont.slotDoubles = slotDoubles; cont.maxL = maxL; cont.resume = 0;
            contResume = cont; contResume.cap = 0; contResume.resume = 1;
        }
    }

    StageTimer t3(c, ROMAN_STAGE_SOLVE);
    const bool coopPlanned = mayFallback && D.wide != 0 && c->coop_ok;
    if (coopPlanned) {                                          // the whole-device solver's barrier words, problem queue and list (k_skipped fills the list)
        HIPCHK(c, WS.wideBar.ensure(sizeof(unsigned) * WIDE_BAR_WORDS));
        HIPCHK(c, WS.fbList.ensure(sizeof(int32_t) * (size_t)B));
        HIPCHK(c, hipMemsetAsync(WS.wideBar.p, 0, sizeof(unsigned) * WIDE_BAR_WORDS, WS.stream));      // counters, generations, queue and the abort flag start at 0 in every launch
    }
    hipLaunchKernelGGL(k_skipped, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, WS.stream, B, WS.probs.as<ProbDesc>(), WS.state.as<ProbState>(), O, WS.queue.as<int>(),
                       coopPlanned ? WS.fbList.as<int32_t>() : (int32_t*)nullptr, coopPlanned ? WS.wideBar.as<unsigned>() : (unsigned*)nullptr);
    DBG(c, "k_skipped");
    if (D.small_only) { t3.stop(); HIPCHK(c, hipGetLastError()); return ROMAN_OK; }   // every problem was k_small's (or is reported skipped)
    // Small problems (<= SMALL_MAXL live associations: the reference's demo scale) take the one-wave-per-problem
    // instantiation; it is launched when such problems have been seen with this parameter block (or, with no history
    // yet, when the association lists are short enough to make them likely).  The general launch takes the rest — and
    // everything when the small one is not part of the batch.
    static const char* smallEnv = getenv("ROMAN_SMALL");        // "0": never
    const bool small = !(smallEnv && smallEnv[0] == '0') && D.p.maxiniters >= 1 && D.p.maxlsiters >= 1 &&
                       (c->hist.valid ? c->hist.smallSeen : maxA <= 16 * SMALL_MAXL);
    const int Lc1 = SMALL_MAXL + 64;
    const size_t ldsUp1 = (size_t)3 * 8 * Lc1 + sizeof(double) * red_doubles(1) + sizeof(uint32_t) * (ST_MAXSL + 2) + sizeof(int) * 8 + 16 + (size_t)COO_CAP * 12;
    const int gridUp1 = std::max(1, std::min(B, c->num_cu * 24));
    // The general instantiation keeps a thread's six vector elements and three quads of the stream in flight in 189 registers:
    // one workgroup per compute unit.  ROMAN_SOLVE_LEAN=1 selects the 128-register instantiation instead (the multiplied vector
    // not held across the stream, two quads in flight, the rest of the state spilled by the compiler around the stream loop):
    // two workgroups — two problems, or a problem and another batch's kernels — per compute unit.  MEASURED SLOWER (round 4,
    // config 3: 1.34 against 1.09 ms per isolated launch, 119-121 k against 123-125 k alignments/s with three batches in
    // flight; B = 1: 0.52 against 0.47 ms): the second workgroup does not buy back what the thinner stream loses.  Off by default.
    static const char* leanEnv = getenv("ROMAN_SOLVE_LEAN");    // "1": use it
    const bool lean = leanEnv && leanEnv[0] == '1' && D.stream_maxL <= LEAN_MAXL;
    // A handful of problems (the single-pair call whose latency bench.py reports): the wide passes are bound by how many bytes
    // ONE compute unit keeps in flight, not by the memory system — twice the quads in flight per lane (ROMAN_SOLVE_DEEP=0/1 forces).
    // MEASURED (round 4, config 2, B = 1): p50 0.678 ms with six quads in flight against 0.667 with three — no gain: off unless forced.
    // Round 5 (four quads, the rebuilt stream loop): 0.953-0.961 against 0.941-0.950 ms per isolated launch of 256 problems, p50 0.588 / 0.586.
    static const char* deepEnv = getenv("ROMAN_SOLVE_DEEP");
    const bool deep = deepEnv && deepEnv[0] == '1';
#define ROMAN_LAUNCH_UP(CZ_)                                                                                                  \
    do {                                                                                                                      \
        auto kup = cont.cap > 0 ? k_solve_up<NW, CZ_, STREAM_MAXL, ST_D, false, true>     /* (the budgeted instantiation: plain stream depth only) */ \
                                : (lean ? k_solve_up<NW, CZ_, LEAN_MAXL, LEAN_D, true> : (deep ? k_solve_up<NW, CZ_, STREAM_MAXL, DEEP_D> : k_solve_up<NW, CZ_, STREAM_MAXL>)); \
        HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(kup), ldsUp)); \
        hipLaunchKernelGGL(kup, dim3(gridUp), dim3(NW * 64), ldsUp, WS.stream, D, B, WS.probs.as<ProbDesc>(), WS.state.as<ProbState>(), feats, assoc, \
                           WS.plp.as<int32_t>(), WS.lp.as<int32_t>(), WS.rowPos.as<uint32_t>(), WS.pld.as<double>(), WS.sliceBase.as<uint32_t>(), \
                           WS.cols16.as<uint16_t>(), WS.vals.as<double>(), u0, O, WS.queue.as<int>(), Lc, small ? SMALL_MAXL + 1 : 0, STREAM_MAXL, (small && c->hist.valid && !c->hist.largeSeen) ? 64 : 1, cont); \
        if (cont.cap > 0)         /* the suspended problems, one workgroup each, to the end */                                  \
            hipLaunchKernelGGL(kup, dim3(std::min(cont.slots, gridUp)), dim3(NW * 64), ldsUp, WS.stream, D, B, WS.probs.as<ProbDesc>(), WS.state.as<ProbState>(), feats, assoc, \
                               WS.plp.as<int32_t>(), WS.lp.as<int32_t>(), WS.rowPos.as<uint32_t>(), WS.pld.as<double>(), WS.sliceBase.as<uint32_t>(), \
                               WS.cols16.as<uint16_t>(), WS.vals.as<double>(), u0, O, WS.queue.as<int>(), Lc, small ? SMALL_MAXL + 1 : 0, STREAM_MAXL, 1, contResume); \
        if (small) {                                                                                                          \
            HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(k_solve_up<1, CZ_, SMALL_MAXL>), ldsUp1));                       \
            hipLaunchKernelGGL((k_solve_up<1, CZ_, SMALL_MAXL>), dim3(gridUp1), dim3(64), ldsUp1, WS.stream, D, B, WS.probs.as<ProbDesc>(), WS.state.as<ProbState>(), feats, assoc, \
                               WS.plp.as<int32_t>(), WS.lp.as<int32_t>(), WS.rowPos.as<uint32_t>(), WS.pld.as<double>(), WS.sliceBase.as<uint32_t>(), \
                               WS.cols16.as<uint16_t>(), WS.vals.as<double>(), u0, O, WS.queue.as<int>() + 1, Lc1, 0, SMALL_MAXL, 1, SolveCont{}); \
        }                                                                                                                     \
    } while (0)
    if (hascz) ROMAN_LAUNCH_UP(true); else ROMAN_LAUNCH_UP(false);
#undef ROMAN_LAUNCH_UP
    DBG(c, "k_solve_up");
    if (mayFallback) {
        // fallback solver (symmetric SELL-64, 32-bit indices): u and u' in LDS when they fit, everything in HBM otherwise
        HIPCHK(c, WS.vMu.ensure(sizeof(double) * R1)); HIPCHK(c, WS.vCu.ensure(sizeof(double) * R1));
        HIPCHK(c, WS.vMun.ensure(sizeof(double) * R1)); HIPCHK(c, WS.vCun.ensure(sizeof(double) * R1));
        HIPCHK(c, WS.gU.ensure(sizeof(double) * R1)); HIPCHK(c, WS.gUn.ensure(sizeof(double) * R1));
        // Few (large) problems: all compute units solve them together, one after the other (k_solve_wide, cooperative
        // launch so that every workgroup is resident); many: one workgroup per problem, u and u' in LDS when they fit.
        bool coop = D.wide != 0 && c->coop_ok;                  // decided when the batch was scored (enqueue_score / roman_set_matrix_data)
        if (coop) {
            const int G = c->num_cu;
            // Several fallback problems: TEAMS — the workgroups of an XCD (or of half an XCD) solve one problem each, with a
            // one-level barrier among themselves; the whole device on one problem at a time otherwise (one huge problem, or
            // live sets beyond a team's registers: a thread owns WIDE_KW elements).
            int a_teams = 0;
            // The kernel forms its teams from a census of the XCD every workgroup really runs on; the HOST sizes a team's share of
            // the buffers from the XCD count the runtime reports (8 on an MI355X in SPX mode; a CPX partition has 1).  Should the
            // two ever disagree — a team larger than its share of the partials, or too small for a live set — the kernel leaves
            // the problem its pre-written ROMAN_ST_INTERNAL record, and the host-pointer entry points run those problems again
            // with the whole device per problem (roman_ctx_set_wide_teams(ctx, 0) does the same for a device-pointer caller).
            const int perXcd = std::max(1, (G + c->num_xcc - 1) / c->num_xcc);
            {
                static const char* marginEnv = getenv("ROMAN_WIDE_MARGIN");
                const int margin = marginEnv ? atoi(marginEnv) : 2;
                auto cap = [&](int sub) { return (int64_t)WIDE_KW * std::max(1, perXcd / sub - margin) * WIDE_NW * 64; };   // (a margin of two workgroups against uneven placement)
                // more, smaller teams while there are problems for them and the live sets fit their registers: a pass of a team is a stream the
                // whole device's bandwidth bounds however it is shared out, plus two barriers and a collect whose latencies only other teams can
                // hide (64 problems of L = 4 900: 24.7 / 16.9 / 12.4 / 12.1 ms with 1 / 2 / 3 / 4 teams per XCD)
                if (mayFallback >= 2 && maxA <= cap(1)) {
                    a_teams = 1;
                    for (int s_ = 2; s_ <= 4; ++s_) if (mayFallback > 6 * s_ && maxA <= cap(s_)) a_teams = s_;
                    // three teams per XCD WITHOUT the reserve when that is what the live sets need (L = 10 000 = 100 x 100 objects on teams of
                    // 10-11 units): a cooperative launch of one workgroup per unit fills every XCD evenly on this part; should a team come out
                    // smaller after all, its problems keep their ROMAN_ST_INTERNAL records and run again on the whole device (align_chunked)
                    if (a_teams == 2 && mayFallback > 18 && G == c->num_cu && maxA <= (int64_t)WIDE_KW * (perXcd / 3) * WIDE_NW * 64) a_teams = 3;
                }
                const char* teamEnv = getenv("ROMAN_WIDE_TEAMS");          // experiments / tests: 0 never, 1 / 2 / 4 teams per XCD whenever the live sets fit (read per call)
                const int forced = teamEnv ? atoi(teamEnv) : c->wide_teams;
                if (teamEnv || c->wide_teams >= 0) a_teams = (forced >= 1 && forced <= 4 && mayFallback >= 1 && maxA <= cap(forced)) ? forced : 0;
            }
            if (a_teams) c->teams_launched = true;
            const int nTeams = a_teams ? 8 * a_teams : 1;       // (teams are numbered XCC_ID * teams-per-XCD + sub-team: 8 XCC ids whatever the partition)
            const int NWGt = a_teams ? std::min(G, 2 * ((perXcd + a_teams - 1) / a_teams) + 8) * WIDE_NW : G * WIDE_NW;     // waves a team can have at most
            // Pull + push passes over a half copy of the matrix — every pair stored once, rows in a per-block order — (kernels.hip.h, build_upper;
            // an instantiation of its own, k_solve_wide<uint16_t, true>): taken where it was measured faster than the plain kernel — TEAMS on live
            // sets of at least 8 000 associations (break-even ~7 700: L = 7 225 19.1 -> 19.9, L = 8 100 25.8 -> 24.9, L = 9 025 30.0 -> 27.7; 64 x L = 10 000: 39.7 -> 36.6 ms of solve, 96 x: 55.8 -> 52.6, 24 x: 15.9 -> 14.4; L = 8 100:
            // 26.6 -> 25.8) —, not below (L = 4 900: 10.0 -> 12.2, L = 6 400: 27.5 -> 31.3: its passes outside the copy cost more registers and a
            // longer collect) and not with the whole device on one problem (n = m = 200: 22.8 -> 22.4).  ROMAN_WIDE_UPPER=0 / 1 forces either.
            int a_ucfg = (D.idx16 && a_teams > 0 && maxA >= 8000) ? 1 : 0;
            { const char* e_ = getenv("ROMAN_WIDE_UPPER"); if (e_ && e_[0]) a_ucfg = (D.idx16 && e_[0] != '0') ? 1 : 0; }
            long long a_partStride = (long long)(((size_t)NWGt * WIDE_MAXCH + (size_t)(a_ucfg ? WIDE_MAXBLK : 1) * ((size_t)(maxA + 63) / 64) + 4) * 64 * 2);   // pieces: chunks + slices (per column block of the half copy)
            HIPCHK(c, WS.widePart.ensure(sizeof(double) * (size_t)a_partStride * (size_t)nTeams));
            HIPCHK(c, WS.wideSlots.ensure(sizeof(double) * 2 * (size_t)G * WIDE_NRED * (size_t)nTeams));
            DevParams Dv = D; int Bv = B;
            const ProbDesc* a_probs = WS.probs.as<ProbDesc>(); ProbState* a_state = WS.state.as<ProbState>();
            const double* a_feats = feats; const int32_t* a_assoc = assoc;
            const int32_t* a_lp = WS.lp.as<int32_t>(); const double* a_ld = WS.ld.as<double>();
            const uint32_t* a_perm = WS.perm.as<uint32_t>(); const uint32_t* a_rpos = WS.rowPos.as<uint32_t>(); const uint32_t* a_sb = WS.sliceBase.as<uint32_t>();
            const void* a_cols = D.idx16 ? (const void*)WS.cols16.as<uint16_t>() : (const void*)WS.cols32.as<uint32_t>(); const double* a_vals = WS.vals.as<double>();
            double* a_vU = WS.gU.as<double>(); double* a_vX = WS.gUn.as<double>(); double* a_vX2 = WS.vCun.as<double>();
            double* a_s0 = WS.vMu.as<double>(); double* a_s1 = WS.vCu.as<double>(); double* a_s2 = WS.vMun.as<double>();
            int32_t* a_plp = WS.plp.as<int32_t>();
            const double* a_u0 = u0; SolveOut a_O = O;
            double* a_part = WS.widePart.as<double>(); double* a_slots = WS.wideSlots.as<double>(); unsigned* a_bar = WS.wideBar.as<unsigned>();
            // dynamic LDS: the support bit map of the gathered vector + its leading part (everything the static part leaves of the 160 KB)
            int a_bmw = (int)((maxA + 63) / 64) + 1;
            HIPCHK(c, WS.wideBm.ensure(sizeof(unsigned long long) * 3 * (size_t)a_bmw * (size_t)nTeams));      // two bit maps + the new slice widths of a compaction
            unsigned long long* a_bm = WS.wideBm.as<unsigned long long>();
            // LDS: three bit maps (the multiplied vector's support, the compacted copy's columns, the window's union), the
            // cumulative slice widths of the stream in use, the multiplied vector's leading part
            const int64_t wideFixed = (int64_t)wide_fixed_lds(a_bmw);
            int a_xcap = (int)(((int64_t)c->lds_max - 4096 - wideFixed) / (int64_t)sizeof(double)) & ~63;
            if (a_xcap < 0) a_xcap = 0;
            const size_t wideLds = sizeof(double) * (size_t)a_xcap + (size_t)wideFixed;
            // Column compaction (kernels.hip.h, k_solve_wide): a mirror of the matrix pools holds the compacted copy.
            // ROMAN_WIDE_COMPACT=0 turns it off; =0xWWTTCC sets passes per window / threshold (x/256) / compactions allowed per problem.
            int a_ccfg = 6 | (128 << 8) | (4 << 16);
            { const char* e_ = getenv("ROMAN_WIDE_COMPACT"); if (e_ && e_[0]) a_ccfg = (int)strtol(e_, nullptr, 0); }
            // the pushed sums of the half copy's column blocks (one slot per workgroup of a team) and its slice widths
            int a_ySlots = NWGt / WIDE_NW;
            int a_ycap = a_ucfg ? ((a_xcap / 3) & ~63) : 0;
            if (a_ucfg && a_ycap < 64) a_ucfg = 0;
            if (a_ucfg) {
                HIPCHK(c, WS.wideY.ensure(sizeof(unsigned long long) * 2 * (size_t)a_ycap * (size_t)a_ySlots * (size_t)nTeams));
                HIPCHK(c, WS.wideUp.ensure(sizeof(uint32_t) * 1800 * (size_t)a_bmw * (size_t)nTeams));      // (kernels.hip.h, build_upper: widths, per-block counts / ranks / rows, piece records)
            }
            unsigned long long* a_yPart = WS.wideY.as<unsigned long long>(); uint32_t* a_upMeta = WS.wideUp.as<uint32_t>();
            if ((a_ccfg & 0xff) || a_ucfg) {
                HIPCHK(c, WS.valsC.ensure(sizeof(double) * (size_t)WS.capNnz));
                HIPCHK(c, WS.colsC.ensure((D.idx16 ? sizeof(uint16_t) : sizeof(uint32_t)) * (size_t)WS.capNnz));
            }
            void* a_colsC = WS.colsC.p; double* a_valsC = WS.valsC.as<double>();
            const void* wideFn = D.idx16 ? (a_ucfg ? reinterpret_cast<const void*>(k_solve_wide<uint16_t, true>) : reinterpret_cast<const void*>(k_solve_wide<uint16_t, false>))
                                         : reinterpret_cast<const void*>(k_solve_wide<uint32_t, false>);
            HIPCHK(c, dyn_lds(c, wideFn, wideLds));
            static const char* tuneEnv = getenv("ROMAN_WIDE_TUNE");
            int a_tune = tuneEnv ? (int)strtol(tuneEnv, nullptr, 0) : 0;
            unsigned long long a_ticks = c->spin_ticks;        // 4 s of the device's wall clock (test hook ROMAN_WIDE_SPIN_MS: shorter)
            const int32_t* a_fb = WS.fbList.as<int32_t>();
            void* args[] = {&Dv, &Bv, &a_probs, &a_state, &a_feats, &a_assoc, &a_lp, &a_ld, &a_perm, &a_rpos, &a_sb, &a_cols, &a_vals,
                            &a_vU, &a_vX, &a_vX2, &a_s0, &a_s1, &a_s2, &a_plp, &a_u0, &a_O, &a_part, &a_slots, &a_bar, &a_bm, &a_bmw, &a_xcap, &a_tune, &a_ticks,
                            &a_teams, &a_fb, &a_partStride, &a_colsC, &a_valsC, &a_ccfg, &a_ucfg, &a_yPart, &a_ySlots, &a_ycap, &a_upMeta};
            // Two whole-device kernels must never be resident together (each would hold compute units while waiting at a
            // grid barrier for workgroups the other one keeps out): with batches in flight on several streams, a
            // launch waits for the previous one of this context.
            if (!c->coopDone) HIPCHK(c, hipEventCreateWithFlags(&c->coopDone, hipEventDisableTiming));
            if (c->coopIssued) HIPCHK(c, hipStreamWaitEvent(WS.stream, c->coopDone, 0));
            const hipError_t e = hipLaunchCooperativeKernel(wideFn, dim3((unsigned)G), dim3(WIDE_NT), args, wideLds, WS.stream);
            if (e == hipSuccess) { HIPCHK(c, hipEventRecord(c->coopDone, WS.stream)); c->coopIssued = true; }
            if (e != hipSuccess) {                              // not available here: the one-workgroup solver does the same work
                (void)hipGetLastError();
                fprintf(stderr, "[roman_hip] cooperative launch failed (%s); large problems use the single-workgroup solver\n", hipGetErrorString(e));
                c->coop_ok = false; coop = false;
                if (D.idx16) return fail(c, ROMAN_E_HIP, "cooperative launch of the large-problem solver failed (%s); run the call again (it now takes the single-workgroup solver)", hipGetErrorString(e));
            }
    DBG(c, "k_solve_wide");
        }
        if (!coop) {
        const size_t fixed = 72 * sizeof(double) + 4 * sizeof(int);
        const int Lcap = (int)(((c->lds_max - fixed) / (2 * sizeof(double))) & ~(size_t)1);
        const size_t lds = (size_t)2 * sizeof(double) * (size_t)Lcap + fixed;
        const int grid = std::max(1, std::min(B, c->num_cu));
        HIPCHK(c, dyn_lds(c, reinterpret_cast<const void*>(k_solve<uint32_t, 1>), lds));
        hipLaunchKernelGGL((k_solve<uint32_t, 1>), dim3(grid), dim3(1024), lds, WS.stream, D, B, WS.probs.as<ProbDesc>(), WS.state.as<ProbState>(), feats, assoc,
                           WS.lp.as<int32_t>(), WS.ld.as<double>(), WS.perm.as<uint32_t>(), WS.rowPos.as<uint32_t>(), WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.cols32.as<uint32_t>(), WS.vals.as<double>(),
                           WS.vMu.as<double>(), WS.vCu.as<double>(), WS.vMun.as<double>(), WS.vCun.as<double>(), WS.gU.as<double>(), WS.gUn.as<double>(),
                           WS.plp.as<int32_t>(), WS.pld.as<double>(), u0, O, WS.queue.as<int>() + 4, Lcap);
        }
    DBG(c, "k_solve");
    }
    t3.stop();
    HIPCHK(c, hipGetLastError());
#ifdef ROMAN_SOLVE_TIMING
    {
        std::vector<unsigned long long> h((size_t)B * 16);
        HIPCHK(c, hipMemcpyAsync(h.data(), WS.hAux3.p, sizeof(unsigned long long) * 16 * (size_t)B, hipMemcpyDeviceToHost, WS.stream));
        HIPCHK(c, hipStreamSynchronize(WS.stream));
        double acc[16] = {0};
        for (int b = 0; b < B; ++b) for (int t = 0; t < 16; ++t) acc[t] += (double)h[(size_t)b * 16 + t];
        // (k_solve_up: cycles of the phases named here; k_solve_wide: 10 ns ticks of trial+publish, barrier 1, stream, barrier 2,
        //  collect+objective, barrier 3, everything else — in slots 0..6)
        const char* nm[8] = {"stream", "spmv-barrier", "decode", "elementwise", "stream-narrow", "setup+tail", "publish", "red-sums"};
        fprintf(stderr, "[solve timing] B=%d cycles/problem:", B);
        double tot_ = 0; for (int t = 0; t < 8; ++t) tot_ += acc[t] / B;
        for (int t = 0; t < 8; ++t) fprintf(stderr, " %s %.0f (n=%.1f)", nm[t], acc[t] / B, acc[8 + t] / B);
        fprintf(stderr, " total %.0f\n", tot_);
        if (B > 1) {   // the slowest problems
            std::vector<std::pair<double, int>> tt;
            for (int b = 0; b < B; ++b) { double t_ = 0; for (int t = 0; t < 8; ++t) t_ += (double)h[(size_t)b * 16 + t]; tt.push_back({t_, b}); }
            std::sort(tt.begin(), tt.end());
            for (int r = 0; r < 3; ++r) {
                const int b = tt[(size_t)(B - 1 - r)].second;
                fprintf(stderr, "   slow #%d (b=%d, %.0f cyc):", r, b, tt[(size_t)(B - 1 - r)].first);
                for (int t = 0; t < 8; ++t) fprintf(stderr, " %s %.0f (n=%.0f)", nm[t], (double)h[(size_t)b * 16 + t], (double)h[(size_t)b * 16 + 8 + t]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "   median problem: %.0f cyc\n", tt[(size_t)B / 2].first);
        }
    }
#endif
    return ROMAN_OK;
}

// how many problems of the batch can be of the fallback kind (0: none, the fallback solver is not launched)
int may_fallback(const DevParams& D, const std::vector<ProbDesc>& hd)
{
    if (!D.allow_fallback) return 0;
    if (D.p.maxiniters < 1 || D.p.maxlsiters < 1) return (int)hd.size();
    int n = 0;
    for (const ProbDesc& d : hd) if (d.nA > D.stream_maxL) ++n;
    return n;
}

int ensure_events(roman_ctx* c)
{
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k)
        for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) {
            if (!c->ws[k].evA[s]) HIPCHK(c, hipEventCreate(&c->ws[k].evA[s]));
            if (!c->ws[k].evB[s]) HIPCHK(c, hipEventCreate(&c->ws[k].evB[s]));
        }
    return ROMAN_OK;
}

// one batch through the stages, on workspace c->cur and its stream (pure enqueue)
int run_batch(roman_ctx* c, const DevParams& D0, const roman_params_t* params, const BatchIn& in, const double* u0, const BatchOut& out)
{
    std::vector<ProbDesc> hd;
    DevParams D;
    int rc = enqueue_score(c, D0, params, in, hd, &D, &out, u0);
    if (rc) return rc;
    int64_t sumA = 0, maxA = 0; for (const ProbDesc& d : hd) { sumA += d.nA; maxA = std::max<int64_t>(maxA, d.nA); }
    return enqueue_solve(c, D, in.B, sumA, maxA, in.feats, in.assoc, u0, false, may_fallback(D, hd), out);
}

// After a synchronous run: did every problem fit its workspace?  (reads the totals the batch copied back; the stream
// has been synchronised by the caller).  On overflow the history now holds the need: the caller runs the batch again.
bool batch_overflowed(roman_ctx* c)
{
    harvest_totals(c, true);
    static const bool dbg = getenv("ROMAN_DEBUG") != nullptr;
    const BatchTotals& t = *WS.pinnedTotals;
    if (dbg) fprintf(stderr, "[roman] batch totals: R=%d maxL=%d items=%d maskWords need %lld cap %lld, nnz need %lld cap %lld, list need %llu cap %lld, overflow=%d sliceGroups=%d\n",
                     t.R, t.maxL, t.items, (long long)t.needMaskWords, WS.capMaskWords, (long long)t.needNnz, WS.capNnz, t.listTop, WS.capList, t.overflow, t.sliceGroups);
    return t.overflow > 0;
}

// run a B=1 solve on the matrices held by the context and pull the solution to the host
int solve_last(roman_ctx* c, const double* u0_host)
{
    roman_ctx::Last& Lst = c->last;
    const int32_t nA = Lst.nA;
    const size_t nA1 = (size_t)std::max(nA, 1);
    const double* dU0 = nullptr;
    if (u0_host) {
        HIPCHK(c, WS.hU0.ensure(sizeof(double) * nA1));
        HIPCHK(c, hipMemcpyAsync(WS.hU0.p, u0_host, sizeof(double) * (size_t)nA, hipMemcpyHostToDevice, WS.stream));
        dU0 = WS.hU0.as<double>();
    }
    const int32_t kmax = std::max(nA, 1);
    HIPCHK(c, WS.oAssoc.ensure(sizeof(int32_t) * 2 * (size_t)kmax)); HIPCHK(c, WS.oN.ensure(sizeof(int32_t)));
    HIPCHK(c, WS.oT.ensure(sizeof(double) * 16)); HIPCHK(c, WS.oStatus.ensure(sizeof(int32_t))); HIPCHK(c, WS.oStats.ensure(sizeof(roman_stats_t)));
    const double* feats = Lst.dense ? nullptr : WS.hFeats.as<double>();
    const int32_t* assoc = (Lst.pd.assocOff >= 0) ? WS.hAssoc.as<int32_t>() : nullptr;
    const BatchOut out{kmax, WS.oAssoc.as<int32_t>(), WS.oN.as<int32_t>(), WS.oT.as<double>(), WS.oStatus.as<int32_t>(), WS.oStats.as<roman_stats_t>()};
    int rc = enqueue_solve(c, Lst.D, 1, nA, nA, feats, assoc, dU0, Lst.hascz, Lst.kind == 1 ? 1 : 0, out);
    if (rc) return rc;
    int32_t nsel = 0;
    HIPCHK(c, hipMemcpyAsync(&nsel, WS.nSel.p, sizeof(int32_t), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipMemcpyAsync(&Lst.stats, WS.oStats.p, sizeof(roman_stats_t), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipMemcpyAsync(&Lst.status, WS.oStatus.p, sizeof(int32_t), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    if (Lst.status & ROMAN_ST_INTERNAL) return fail(c, ROMAN_E_INTERNAL, "the solve reported ROMAN_ST_INTERNAL: a bounded wait of the whole-device solver expired");
    Lst.nsel = nsel;
    Lst.nodes.assign((size_t)std::max(nsel, 0), 0);
    const int L = Lst.L;
    std::vector<double> ul((size_t)std::max(L, 1)); std::vector<int32_t> lpv((size_t)std::max(L, 1));
    if (nsel > 0) HIPCHK(c, hipMemcpyAsync(Lst.nodes.data(), WS.nodesOrig.p, sizeof(int32_t) * (size_t)nsel, hipMemcpyDeviceToHost, WS.stream));
    if (L > 0) {
        // u comes back indexed like the solver's vectors: by position (stream layout) or by live index (fallback);
        // the matching association-index pool maps either to the caller's association order
        HIPCHK(c, hipMemcpyAsync(ul.data(), WS.uOut.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, WS.stream));
        HIPCHK(c, hipMemcpyAsync(lpv.data(), Lst.kind == 0 ? WS.plp.p : WS.lp.p, sizeof(int32_t) * (size_t)L, hipMemcpyDeviceToHost, WS.stream));
    }
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    Lst.u.assign((size_t)nA, 0.0);
    for (int k = 0; k < L; ++k) Lst.u[(size_t)lpv[k]] = ul[k];
    Lst.solved = true;
    return ROMAN_OK;
}

// Download the matrix of the last single problem and decode it into per-row lists over LIVE indices, both triangles,
// ascending columns: rs/rl index into cols/vals (row-contiguous on return; bit 31 of a column = "C_pq == 0"); inert
// slots are dropped.  lp / diag: association index and diagonal value per live index.
int fetch_last_csr(const roman_ctx* cc, std::vector<uint32_t>& rs, std::vector<uint32_t>& rl, std::vector<uint32_t>& cols,
                   std::vector<double>& vals, std::vector<int32_t>& lp, std::vector<double>& ls)
{
    roman_ctx* c = const_cast<roman_ctx*>(cc);
    const roman_ctx::Last& Lst = c->last;
    const int L = Lst.L; const int64_t cap = Lst.nnzCap;
    const size_t L1 = (size_t)std::max(L, 1), cap1 = (size_t)std::max<int64_t>(cap, 1);
    std::vector<uint32_t> jcnt(L1, 0), jpos(L1, 0), jperm(L1, 0), jsw(L1, 0), jsb(L1, 0), jcols(cap1, 0); std::vector<double> jvals(cap1, 0.0);
    rs.assign(L1, 0); rl.assign(L1, 0); lp.assign(L1, 0); ls.assign(L1, 0.0);
    HIPCHK(c, hipSetDevice(c->device));
    roman_ctx::Workspace& W0 = c->ws[0];
    const bool up = Lst.kind == 0;
    if (L > 0) {
        const int nsl = (L + 63) / 64;
        HIPCHK(c, hipMemcpy(jcnt.data(), W0.rowCnt.p, sizeof(uint32_t) * (size_t)L, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(jpos.data(), W0.rowPos.p, sizeof(uint32_t) * (size_t)L, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(jperm.data(), W0.perm.p, sizeof(uint32_t) * (size_t)L, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(jsw.data(), W0.sliceWidth.p, sizeof(uint32_t) * (size_t)nsl, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(jsb.data(), W0.sliceBase.p, sizeof(uint32_t) * (size_t)nsl, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(lp.data(), W0.lp.p, sizeof(int32_t) * (size_t)L, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(ls.data(), W0.ld.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost));   // diagonal values
    }
    if (cap > 0) {
        HIPCHK(c, hipMemcpy(jvals.data(), W0.vals.p, sizeof(double) * (size_t)cap, hipMemcpyDeviceToHost));
        if (up) {
            std::vector<uint16_t> c16((size_t)cap);
            HIPCHK(c, hipMemcpy(c16.data(), W0.cols16.p, sizeof(uint16_t) * (size_t)cap, hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < cap; ++k) jcols[(size_t)k] = (c16[(size_t)k] & 0x8000u) ? (0x80000000u | (c16[(size_t)k] & 0x7fffu)) : c16[(size_t)k];
        } else if (Lst.D.idx16) {                               // fallback layout with 16-bit labels: 0xffff = inert
            std::vector<uint16_t> c16((size_t)cap);
            HIPCHK(c, hipMemcpy(c16.data(), W0.cols16.p, sizeof(uint16_t) * (size_t)cap, hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < cap; ++k) jcols[(size_t)k] = (c16[(size_t)k] == 0xffffu) ? 0xffffffffu : (uint32_t)c16[(size_t)k];
        } else {
            HIPCHK(c, hipMemcpy(jcols.data(), W0.cols32.p, sizeof(uint32_t) * (size_t)cap, hipMemcpyDeviceToHost));
        }
    }
    std::vector<std::vector<std::pair<uint32_t, double>>> rows(L1);
    if (up) {
        // stream layout: row p (POSITION) holds its strict-upper entries (p, q), q a position; rowCnt[p] slots are in use
        for (int p = 0; p < L; ++p) {
            const uint32_t sl = (uint32_t)p >> 6, slot = (uint32_t)p & 63u;
            const uint32_t kp = jperm[(size_t)p];
            for (uint32_t e = 0; e < ((jcnt[(size_t)p] + 3u) & ~3u); ++e) {                         // (every slot of the row's quads: the fill may rotate them)
                const uint32_t cq = jcols[h_col_pos(true, jsb[sl], slot, e)]; const double v = jvals[h_val_pos(true, jsb[sl], slot, e)];
                const uint32_t q = cq & 0x7fffffffu;
                if (q >= (uint32_t)L) continue;                                                      // inert slot (dummy column L + slot)
                const uint32_t kq = jperm[(size_t)q], flag = cq & 0x80000000u;
                rows[kp].push_back({kq | flag, v}); rows[kq].push_back({kp | flag, v});
            }
        }
    } else {
        for (int k = 0; k < L; ++k) {
            const uint32_t pos = jpos[(size_t)k], sl = pos >> 6, slot = pos & 63u;
            for (uint32_t e = 0; e < jcnt[(size_t)k]; ++e) {
                const uint32_t cq = jcols[h_col_pos(true, jsb[sl], slot, e)]; const double v = jvals[h_val_pos(true, jsb[sl], slot, e)];
                if (cq == 0xffffffffu || (v == 0.0 && (cq & 0x80000000u) && (cq & 0x7fffffffu) == pos)) continue;   // inert slot (16-bit: no column; 32-bit: the row's own position, flagged)
                rows[(size_t)k].push_back({jperm[(size_t)(cq & 0x7fffffffu)] | (cq & 0x80000000u), v});     // column labels are positions
            }
        }
    }
    cols.clear(); vals.clear();
    for (int k = 0; k < L; ++k) {
        auto& r = rows[(size_t)k];
        std::sort(r.begin(), r.end(), [](const std::pair<uint32_t, double>& a, const std::pair<uint32_t, double>& b) { return (a.first & 0x7fffffffu) < (b.first & 0x7fffffffu); });
        rs[(size_t)k] = (uint32_t)cols.size();
        for (auto& e : r) { cols.push_back(e.first); vals.push_back(e.second); }
        rl[(size_t)k] = (uint32_t)r.size();
    }
    if (cols.empty()) { cols.push_back(0); vals.push_back(0.0); }
    return ROMAN_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* roman_version(void) { return "roman_hip 0.2.0 gfx950 (HIP, wave64, f64)"; }

int roman_params_default(roman_params_t* p)
{
    if (!p) return fail(nullptr, ROMAN_E_INVALID, "params is NULL");
    memset(p, 0, sizeof(*p));
    p->invariant = ROMAN_INV_ROMAN; p->point_dim = 3; p->fusion_method = ROMAN_FUSE_GEOMETRIC_MEAN; p->rescale_u0 = 1;
    p->sigma = 0.4; p->epsilon = 0.6; p->mindist = 0.2;
    p->distance_weight = p->ratio_weight = p->cosine_weight = 1.0;
    p->cosine_min = 0.5; p->cosine_max = 0.7; p->gravity_unc_ang_rad = 0.0872665;
    p->tol_u = 1e-8; p->tol_F = 1e-9; p->beta = 0.25; p->eps = 1e-9; p->affinityeps = 1e-4;
    p->maxiniters = 200; p->maxoliters = 1000; p->maxlsiters = 99;
    return ROMAN_OK;
}

const char* roman_last_error(const roman_ctx_t* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int roman_ctx_create(roman_ctx_t** out, int device, void* stream)
{
    if (!out) return fail(nullptr, ROMAN_E_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(nullptr, ROMAN_E_NO_DEVICE, "no HIP device visible (libroman_hip has no CPU fallback)"); }
    if (device < 0 || device >= ndev) return fail(nullptr, ROMAN_E_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, ROMAN_E_HIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, ROMAN_E_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(nullptr, ROMAN_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    roman_ctx* c = new (std::nothrow) roman_ctx();
    if (!c) return fail(nullptr, ROMAN_E_NOMEM, "out of host memory");
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // XCDs of this device (8 on an MI355X in SPX mode, 1 in CPX mode): the team mode of k_solve_wide sizes a team's buffers from it
        int nx = 0;
        if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, device) != hipSuccess || nx < 1 || nx > 8) { (void)hipGetLastError(); nx = 8; }
        if (const char* e = getenv("ROMAN_NUM_XCC")) { const int v = atoi(e); if (v >= 1 && v <= 64) nx = v; }   // test hook: a host-side count that does not match the device's
        c->num_xcc = nx;
    }
    c->lds_max = prop.sharedMemPerBlock >= 163840 ? (size_t)(160 * 1024 - 256) : (size_t)prop.sharedMemPerBlock;
    { int coopAttr = 0; c->coop_ok = hipDeviceGetAttribute(&coopAttr, hipDeviceAttributeCooperativeLaunch, device) == hipSuccess && coopAttr != 0; (void)hipGetLastError(); }
    if (c->coop_ok) {
        // The whole-device solver is chosen when a batch is SCORED (16-bit labels are written for it): make sure here, once,
        // that one 512-thread workgroup with its largest dynamic LDS fits a compute unit for both instantiations — a
        // cooperative launch of num_cu workgroups then cannot fail for lack of residency after the fill has already run.
        const size_t wideLdsMax = c->lds_max > 4096 ? c->lds_max - 4096 : 0;
        const void* fns[3] = {reinterpret_cast<const void*>(k_solve_wide<uint16_t, false>), reinterpret_cast<const void*>(k_solve_wide<uint32_t, false>),
                              reinterpret_cast<const void*>(k_solve_wide<uint16_t, true>)};
        for (const void* fn : fns) {
            int nb = 0;
            if (dyn_lds(c, fn, wideLdsMax) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, WIDE_NT, wideLdsMax) != hipSuccess || nb < 1) { (void)hipGetLastError(); c->coop_ok = false; }
        }
    }
    {   // wall_clock64() ticks at the rate the runtime reports (kHz; 100 MHz on this part)
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
        double ms = 4000.0;
        if (const char* e = getenv("ROMAN_WIDE_SPIN_MS")) { const double v = atof(e); if (v >= 0.0) ms = v; }   // test hook (0: the first unsuccessful poll of a wait is a timeout)
        c->spin_ticks = (unsigned long long)((double)khz * ms);
    }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, ROMAN_E_HIP, "hipStreamCreate failed"); }
        c->own_stream = true;
    }
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) c->ws[k].stream = c->stream;
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) {
        if (hipHostMalloc((void**)&c->ws[k].pinnedTotals, sizeof(BatchTotals), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->ws[k].totEvent, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ws[k].probsEvent, hipEventDisableTiming) != hipSuccess) {
            roman_ctx_destroy(c); return fail(nullptr, ROMAN_E_NOMEM, "hipHostMalloc / hipEventCreate failed");
        }
        memset(c->ws[k].pinnedTotals, 0, sizeof(BatchTotals));
    }
    *out = c;
    return ROMAN_OK;
}

int roman_ctx_destroy(roman_ctx_t* c)
{
    if (!c) return ROMAN_OK;
    (void)hipSetDevice(c->device);
    (void)roman_ctx_sync(c);
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) {
        roman_ctx::Workspace& W = c->ws[k];
        DevBuf* all[] = {&W.probs, &W.state, &W.totals, &W.queue, &W.cosPool, &W.cosDense, &W.tabPool, &W.qtabPool, &W.sTmp, &W.chunkCnt,
                         &W.lp, &W.li, &W.lj, &W.ls, &W.ld, &W.lza, &W.lzb, &W.plp, &W.pli, &W.plj, &W.pls, &W.pld, &W.plza, &W.plzb,
                         &W.rowCnt, &W.rowPos, &W.perm, &W.sliceWidth, &W.sliceBase, &W.items, &W.maskPool, &W.prefPool, &W.listPool, &W.listOff,
                         &W.vMu, &W.vCu, &W.vMun, &W.vCun, &W.gU, &W.gUn, &W.uOut, &W.nodesOrig, &W.nSel, &W.widePart, &W.wideSlots, &W.wideBar, &W.wideBm, &W.wideY, &W.wideUp, &W.fbList, &W.cols16, &W.cols32, &W.vals, &W.colsC, &W.valsC, &W.contSpill, &W.contList,
                         &W.hFeats, &W.hAssoc, &W.hU0, &W.oAssoc, &W.oN, &W.oT, &W.oStatus, &W.oStats, &W.hAux1, &W.hAux2, &W.hAux3, &W.oAll};
        for (DevBuf* b : all) b->release();
        if (W.pinnedTotals) (void)hipHostFree(W.pinnedTotals);
        if (W.totEvent) (void)hipEventDestroy(W.totEvent);
        if (W.pinnedProbs) (void)hipHostFree(W.pinnedProbs);
        if (W.probsEvent) (void)hipEventDestroy(W.probsEvent);
        for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) { if (W.evA[s]) (void)hipEventDestroy(W.evA[s]); if (W.evB[s]) (void)hipEventDestroy(W.evB[s]); }
        if (W.done) (void)hipEventDestroy(W.done);
    }
    if (c->hostOut) (void)hipHostFree(c->hostOut);
    if (c->evIn) (void)hipEventDestroy(c->evIn);
    if (c->coopDone) (void)hipEventDestroy(c->coopDone);
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) if (c->istream[k]) (void)hipStreamDestroy(c->istream[k]);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ROMAN_OK;
}

// Batches in flight.  depth 1 (default): every call runs on the context's stream.  depth 2 or 3: batch calls
// (roman_align_batch_dev) rotate over that many workspaces, each with an internal stream that starts after
// the work already queued on the context's stream; their results are complete after roman_ctx_sync (or a
// device-wide synchronisation), NOT after synchronising the context's stream alone.
int roman_ctx_set_pipeline(roman_ctx_t* c, int depth)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (depth < 1 || depth > ROMAN_MAX_PIPELINE) return fail(c, ROMAN_E_INVALID, "pipeline depth must be 1..%d", ROMAN_MAX_PIPELINE);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = roman_ctx_sync(c);
    if (rc) return rc;
    if (depth >= 2) {
        for (int k = 0; k < depth; ++k) {
            if (!c->istream[k]) HIPCHK(c, hipStreamCreateWithFlags(&c->istream[k], hipStreamNonBlocking));
            if (!c->ws[k].done) HIPCHK(c, hipEventCreateWithFlags(&c->ws[k].done, hipEventDisableTiming));
        }
        if (!c->evIn) HIPCHK(c, hipEventCreateWithFlags(&c->evIn, hipEventDisableTiming));
    }
    c->pipeline = depth; c->next_ws = 0; c->cur = 0; c->latest_ws = -1;
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) { c->ws[k].issued = false; c->ws[k].stream = c->stream; }
    return ROMAN_OK;
}

// Make the context's (caller's) stream wait — without blocking the host — for the pipelined batches issued so
// far: all of them (skip_latest == 0) or all but the most recent one (skip_latest != 0), so that work queued on
// the caller's stream afterwards (e.g. the RCCL all_gather of batch k-1's records) sees their results while the
// latest batch keeps running.
int roman_ctx_join_on(roman_ctx_t* c, int skip_latest, void* stream)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t target = stream ? (hipStream_t)stream : c->stream;
    if (c->pipeline < 2) {                                      // everything runs on the context's stream: another stream waits for what is queued there
        if (target != c->stream) {
            if (!c->evIn) HIPCHK(c, hipEventCreateWithFlags(&c->evIn, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->evIn, c->stream));
            HIPCHK(c, hipStreamWaitEvent(target, c->evIn, 0));
        }
        return ROMAN_OK;
    }
    for (int k = 0; k < c->pipeline; ++k) {
        if (skip_latest && k == c->latest_ws) continue;
        if (c->ws[k].done && c->ws[k].issued) HIPCHK(c, hipStreamWaitEvent(target, c->ws[k].done, 0));
    }
    return ROMAN_OK;
}

int roman_ctx_join(roman_ctx_t* c, int skip_latest) { return roman_ctx_join_on(c, skip_latest, nullptr); }

int roman_ctx_sync(roman_ctx_t* c)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) if (c->istream[k]) HIPCHK(c, hipStreamSynchronize(c->istream[k]));
    if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    return ROMAN_OK;
}

int roman_ctx_skipped(roman_ctx_t* c, int wait, int64_t* n)
{
    if (!c || !n) return fail(c, ROMAN_E_INVALID, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (wait) { int rc = roman_ctx_sync(c); if (rc) return rc; }
    harvest_totals(c, wait != 0);
    *n = (int64_t)c->skippedTotal;
    return ROMAN_OK;
}

// --- instrumentation -----------------------------------------------------------------------------
int roman_profile_enable(roman_ctx_t* c, int on)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    if (on) { int rc = ensure_events(c); if (rc) return rc; }
    else for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) prof_flush(c, k, s);
    c->profile = on != 0;
    return ROMAN_OK;
}
int roman_profile_reset(roman_ctx_t* c)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) prof_flush(c, k, s);
    for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) { c->prof_ms[s] = 0.0; c->prof_n[s] = 0; }
    return ROMAN_OK;
}
int roman_profile_get(roman_ctx_t* c, double ms[ROMAN_STAGE_COUNT], int64_t launches[ROMAN_STAGE_COUNT])
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    for (int k = 0; k < ROMAN_MAX_PIPELINE; ++k) for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) prof_flush(c, k, s);
    for (int s = 0; s < ROMAN_STAGE_COUNT; ++s) { if (ms) ms[s] = c->prof_ms[s]; if (launches) launches[s] = c->prof_n[s]; }
    return ROMAN_OK;
}

// --- the batched hot path --------------------------------------------------------------------------
int roman_align_batch_dev(roman_ctx_t* c, const roman_params_t* params, int32_t B,
                          const double* feats, const int64_t* off1, const int32_t* n1,
                          const int64_t* off2, const int32_t* n2, int32_t F,
                          const int32_t* assoc, const int64_t* assoc_off, const double* u0,
                          int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out,
                          double* T_out, int32_t* status_out, roman_stats_t* stats_out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (B < 0) return fail(c, ROMAN_E_INVALID, "B < 0");
    if (B == 0) return ROMAN_OK;
    if (!off1 || !n1 || !off2 || !n2 || !assoc_out || !n_assoc_out || !T_out || !status_out || kmax < 0)
        return fail(c, ROMAN_E_INVALID, "NULL metadata/output pointer or kmax < 0");
    if (assoc && !assoc_off) return fail(c, ROMAN_E_INVALID, "assoc given without assoc_off");
    HIPCHK(c, hipSetDevice(c->device));
    DevParams D;
    int rc = make_dev_params(c, params, F, &D);
    if (rc) return rc;
    if (!feats) {
        bool any = false;
        for (int b = 0; b < B; ++b) any = any || (n1[b] > 0 || n2[b] > 0);
        if (any) return fail(c, ROMAN_E_INVALID, "feats is NULL");
    }
    const BatchIn in{B, feats, off1, n1, off2, n2, F, assoc, assoc_off};
    const BatchOut out{kmax, assoc_out, n_assoc_out, T_out, status_out, stats_out};
    if (c->pipeline >= 2) {
        // next workspace: its internal stream starts behind the work already queued on the caller's stream
        const int k = c->next_ws;
        c->next_ws = (c->next_ws + 1) % c->pipeline; c->latest_ws = k;
        HIPCHK(c, hipEventRecord(c->evIn, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->istream[k], c->evIn, 0));
        c->cur = k; c->ws[k].stream = c->istream[k];
        rc = run_batch(c, D, params, in, u0, out);
        if (!rc) {
            HIPCHK(c, hipEventRecord(c->ws[k].done, c->ws[k].stream));
            c->ws[k].issued = true;
        }
        c->cur = 0;
        c->last.scored = false; c->last.solved = false;        // workspace 0 is reused: the stepwise problem it held is gone
        return rc;
    }
    c->cur = 0; WS.stream = c->stream;
    rc = run_batch(c, D, params, in, u0, out);
    if (rc) return rc;
    c->last.scored = false; c->last.solved = false;
    return ROMAN_OK;
}

int roman_ctx_cosine_screen_stats(roman_ctx_t* c, int64_t* screened_batches, int64_t* dense_batches, double* latest_fallback_share)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    harvest_totals(c, false);
    if (screened_batches) *screened_batches = c->cosScreenBatches;
    if (dense_batches) *dense_batches = c->cosDenseBatches;
    if (latest_fallback_share) *latest_fallback_share = c->hist.cosSeen ? c->hist.cosDenseFrac : 0.0;
    return ROMAN_OK;
}

int roman_ctx_has_history(roman_ctx_t* c, const roman_params_t* params, int32_t F, int32_t* yes)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (!params || !yes) return fail(c, ROMAN_E_INVALID, "NULL argument");
    harvest_totals(c, false);
    const roman_ctx::Hist& H = c->hist;
    *yes = (H.valid && H.tagged && H.F == F && memcmp(&H.params, params, sizeof(roman_params_t)) == 0) ? 1 : 0;
    return ROMAN_OK;
}

int roman_ctx_set_wide_teams(roman_ctx_t* c, int teams_per_xcd)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (!(teams_per_xcd == -1 || teams_per_xcd == 0 || teams_per_xcd == 1 || teams_per_xcd == 2 || teams_per_xcd == 4))
        return fail(c, ROMAN_E_INVALID, "teams per XCD must be -1 (automatic), 0, 1, 2 or 4");
    c->wide_teams = teams_per_xcd;
    return ROMAN_OK;
}

int roman_ctx_set_host_batching(roman_ctx_t* c, int chunk, int depth)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (chunk < 1 || depth < 1 || depth > ROMAN_MAX_PIPELINE) return fail(c, ROMAN_E_INVALID, "chunk must be >= 1 and depth 1..%d", ROMAN_MAX_PIPELINE);
    c->host_chunk = chunk; c->host_depth = depth;
    return ROMAN_OK;
}

namespace {
// Attempts a problem gets before the host-pointer entry points give up on its workspace: the speculative sizes can be corrected
// once per pool (live set beyond the launch's LDS tiles / fallback kernels not launched, bit matrix, matrix slots, lists), and a
// skip the library chose itself (small_only: the general kernels left out) is one more.
constexpr int MAX_ATTEMPTS = 5;

// The host-pointer batch in CALLS of c->host_chunk problems, c->host_depth of them in flight: what the pipelined loop of a
// device-pointer caller does, for callers that hold host arrays — the straggler tail of one call's solver
// overlaps the next calls' affinity builds.  Inputs are already on the device; every call writes its own rows of the output
// arrays.  Problems a call skipped for workspace (ROMAN_ST_WORKSPACE) are issued again, those only, in runs of consecutive
// problems.  With no sizing history for this parameter block the first call runs alone and is waited for: the calls queued
// behind it size their pools from what it needed instead of repeating its guess.
int align_chunked(roman_ctx* c, const roman_params_t* params, const BatchIn& in, const double* dU0, const BatchOut& out)
{
    const int B = in.B, chunk = c->host_chunk;
    const int saved = c->pipeline;
    int rc = roman_ctx_set_pipeline(c, c->host_depth);
    if (rc) return rc;
    c->teams_launched = false;
    std::vector<int64_t> uoff;                                  // start of problem b's slice of u0
    if (dU0) {
        uoff.assign((size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) {
            const int64_t na = in.assoc ? in.assoc_off[b + 1] - in.assoc_off[b] : 0;
            uoff[b + 1] = uoff[b] + (na > 0 ? na : (int64_t)in.n1[b] * in.n2[b]);
        }
    }
    auto issue = [&](int lo, int hi) -> int {
        return roman_align_batch_dev(c, params, hi - lo, in.feats, in.off1 + lo, in.n1 + lo, in.off2 + lo, in.n2 + lo, in.F,
                                     in.assoc, in.assoc ? in.assoc_off + lo : nullptr, dU0 ? dU0 + uoff[lo] : nullptr, out.kmax,
                                     out.assoc_out + (size_t)lo * (size_t)out.kmax * 2, out.n_assoc_out + lo, out.T_out + (size_t)lo * 16, out.status_out + lo,
                                     out.stats_out ? out.stats_out + lo : nullptr);
    };
    auto restore = [&](int code) -> int { const int r2 = roman_ctx_set_pipeline(c, saved); c->cur = 0; c->ws[0].stream = c->stream; return code ? code : r2; };
    int lo = 0;
    const roman_ctx::Hist& H = c->hist;
    if (!(H.valid && H.tagged && H.F == in.F && memcmp(&H.params, params, sizeof(roman_params_t)) == 0)) {
        rc = issue(0, std::min(B, chunk));
        if (rc) return restore(rc);
        rc = roman_ctx_sync(c);
        if (rc) return restore(rc);
        harvest_totals(c, true);
        lo = std::min(B, chunk);
    }
    for (; lo < B; lo += chunk) { rc = issue(lo, std::min(B, lo + chunk)); if (rc) return restore(rc); }
    std::vector<int32_t> st((size_t)B);
    const int teams_saved = c->wide_teams;
    bool no_teams_tried = false;
    auto restore2 = [&](int code) -> int { c->wide_teams = teams_saved; return restore(code); };
    for (int attempt = 1; ; ++attempt) {
        rc = roman_ctx_sync(c);
        if (rc) return restore2(rc);
        harvest_totals(c, true);
        if (hipMemcpy(st.data(), out.status_out, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return restore2(fail(c, ROMAN_E_HIP, "status read-back failed")); }
        int nskip = 0, nint = 0;
        for (int b = 0; b < B; ++b) { nskip += (st[b] & ROMAN_ST_WORKSPACE) ? 1 : 0; nint += (st[b] & ROMAN_ST_INTERNAL) ? 1 : 0; }
        int again = ROMAN_ST_WORKSPACE;
        if (nint && c->teams_launched && !no_teams_tried) {
            // a team of the whole-device solver that could not hold its problem leaves ROMAN_ST_INTERNAL like an expired wait does:
            // those problems once more, the whole device on one problem at a time
            no_teams_tried = true; c->wide_teams = 0; again |= ROMAN_ST_INTERNAL;
        } else if (!nskip) break;                              // nothing left to issue again (a remaining ROMAN_ST_INTERNAL is final: the caller reports it)
        // (a final ROMAN_ST_INTERNAL does not end the loop while skipped problems remain: they are still solvable)
        if (attempt >= MAX_ATTEMPTS) {
            if (!nskip) break;                                  // only the teams-off retry was pending: the ROMAN_ST_INTERNAL records speak for themselves
            return restore2(fail(c, ROMAN_E_NOMEM, "the sparse workspace of %d problem(s) still does not fit after %d attempts", nskip, attempt));
        }
        for (int b = 0; b < B; ) {                              // runs of consecutive problems to issue again, at most a chunk long
            if (!(st[b] & again)) { ++b; continue; }
            int e = b + 1;
            while (e < B && e - b < chunk && (st[e] & again)) ++e;
            rc = issue(b, e);
            if (rc) return restore2(rc);
            b = e;
        }
    }
    return restore2(ROMAN_OK);
}
}  // namespace

namespace {
// The synchronous part shared by the two host-output entry points: inputs are on the device (`in`, dU0), the outputs of every call
// land in ONE device block (T | stats | assoc | n | status) and come back with ONE copy through a pinned landing block — a single
// pair's whole result is 1.8 KB, and five separate read-backs cost more than its build kernels.
int align_to_host(roman_ctx* c, const DevParams& D, const roman_params_t* params, const BatchIn& in, const double* dU0,
                  int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out, double* T_out, int32_t* status_out, roman_stats_t* stats_out)
{
    const int32_t B = in.B;
    int rc = ROMAN_OK;
    bool copied = false;
    const size_t kb = (size_t)B * (size_t)std::max(kmax, 1);
    const size_t oT = 0, oS = oT + sizeof(double) * 16 * (size_t)B, oA = oS + sizeof(roman_stats_t) * (size_t)B,
                 oNn = oA + sizeof(int32_t) * 2 * kb, oSt = oNn + sizeof(int32_t) * (size_t)B, total = oSt + sizeof(int32_t) * (size_t)B;
    static_assert(sizeof(roman_stats_t) % 8 == 0, "the blocks behind the statistics stay 8-byte aligned");
    HIPCHK(c, WS.oAll.ensure(total));
    if (c->hostOutCap < total) {
        if (c->hostOut) { (void)hipHostFree(c->hostOut); c->hostOut = nullptr; c->hostOutCap = 0; }
        const size_t want = total + total / 4 + 4096;
        HIPCHK(c, hipHostMalloc(&c->hostOut, want, hipHostMallocDefault));
        c->hostOutCap = want;
    }
    char* const dev = WS.oAll.as<char>();
    const BatchOut out{kmax, reinterpret_cast<int32_t*>(dev + oA), reinterpret_cast<int32_t*>(dev + oNn), reinterpret_cast<double*>(dev + oT),
                       reinterpret_cast<int32_t*>(dev + oSt), reinterpret_cast<roman_stats_t*>(dev + oS)};
    if (B > c->host_chunk && c->host_depth >= 2) {
        // many problems: calls of host_chunk problems, host_depth of them in flight, skipped problems issued again (align_chunked)
        rc = align_chunked(c, params, in, dU0, out);
        if (rc) return rc;
    } else
    // this entry point is synchronous anyway: when a problem did not fit the speculatively sized pools, run again
    // with the need the first attempt recorded
    {
        const int teams_saved = c->wide_teams;
        for (int attempt = 0; ; ++attempt) {
            c->teams_launched = false;
            rc = run_batch(c, D, params, in, dU0, out);
            if (rc) { c->wide_teams = teams_saved; return rc; }
            // the read-back rides behind the batch on the same stream: ONE wait covers both (a retry below overwrites the landing block)
            if (hipMemcpyAsync(c->hostOut, dev, total, hipMemcpyDeviceToHost, WS.stream) != hipSuccess) { (void)hipGetLastError(); c->wide_teams = teams_saved; return fail(c, ROMAN_E_HIP, "result read-back failed"); }
            if (hipStreamSynchronize(WS.stream) != hipSuccess) { (void)hipGetLastError(); c->wide_teams = teams_saved; return fail(c, ROMAN_E_HIP, "hipStreamSynchronize failed"); }
            copied = true;
            if (batch_overflowed(c)) {
                if (attempt + 1 >= MAX_ATTEMPTS) { c->wide_teams = teams_saved; return fail(c, ROMAN_E_NOMEM, "the sparse workspace still does not fit after %d attempts", attempt + 1); }
                continue;
            }
            if (c->teams_launched && c->wide_teams != 0) {      // a team that could not hold its problem leaves ROMAN_ST_INTERNAL: once more, the whole device per problem
                const int32_t* stv = reinterpret_cast<const int32_t*>(static_cast<const char*>(c->hostOut) + oSt);
                bool anyInt = false;
                for (int b = 0; b < B; ++b) anyInt = anyInt || (stv[b] & ROMAN_ST_INTERNAL);
                if (anyInt && attempt + 1 < MAX_ATTEMPTS) { c->wide_teams = 0; continue; }
            }
            break;
        }
        c->wide_teams = teams_saved;
    }
    if (!copied) {
        HIPCHK(c, hipMemcpyAsync(c->hostOut, dev, total, hipMemcpyDeviceToHost, WS.stream));
        HIPCHK(c, hipStreamSynchronize(WS.stream));
    }
    {
        const char* h = static_cast<const char*>(c->hostOut);
        if (kmax > 0) memcpy(assoc_out, h + oA, sizeof(int32_t) * 2 * (size_t)B * (size_t)kmax);
        memcpy(n_assoc_out, h + oNn, sizeof(int32_t) * (size_t)B);
        memcpy(T_out, h + oT, sizeof(double) * 16 * (size_t)B);
        memcpy(status_out, h + oSt, sizeof(int32_t) * (size_t)B);
        if (stats_out) memcpy(stats_out, h + oS, sizeof(roman_stats_t) * (size_t)B);
    }
    c->last.scored = false; c->last.solved = false;
    {   // a problem the whole-device solver gave up on has no result: an error, not a quiet "0 associations"
        int nint = 0, first = -1;
        for (int b = 0; b < B; ++b) if (status_out[b] & ROMAN_ST_INTERNAL) { if (first < 0) first = b; ++nint; }
        if (nint) return fail(c, ROMAN_E_INTERNAL, "%d problem(s) reported ROMAN_ST_INTERNAL (first: %d): a bounded wait of the whole-device solver expired", nint, first);
    }
    return ROMAN_OK;
}
}  // namespace

int roman_align_batch(roman_ctx_t* c, const roman_params_t* params, int32_t B,
                      const double* feats, int64_t n_objects,
                      const int64_t* off1, const int32_t* n1, const int64_t* off2, const int32_t* n2, int32_t F,
                      const int32_t* assoc, const int64_t* assoc_off, const double* u0,
                      int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out,
                      double* T_out, int32_t* status_out, roman_stats_t* stats_out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (B < 0 || n_objects < 0 || F < 0) return fail(c, ROMAN_E_INVALID, "negative size");
    if (B == 0) return ROMAN_OK;
    if (!off1 || !n1 || !off2 || !n2 || !assoc_out || !n_assoc_out || !T_out || !status_out || kmax < 0)
        return fail(c, ROMAN_E_INVALID, "NULL metadata/output pointer or kmax < 0");
    if (assoc && !assoc_off) return fail(c, ROMAN_E_INVALID, "assoc given without assoc_off");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    DevParams D;
    int rc = make_dev_params(c, params, F, &D);
    if (rc) return rc;
    int64_t sumA = 0;
    if (assoc && assoc_off[0] != 0) return fail(c, ROMAN_E_INVALID, "assoc_off[0] must be 0");
    for (int b = 0; b < B; ++b) {
        if (n1[b] < 0 || n2[b] < 0 || off1[b] < 0 || off2[b] < 0 || off1[b] + n1[b] > n_objects || off2[b] + n2[b] > n_objects)
            return fail(c, ROMAN_E_INVALID, "problem %d reads objects outside feats[0..%lld)", b, (long long)n_objects);
        if (assoc) {
            if (assoc_off[b + 1] < assoc_off[b]) return fail(c, ROMAN_E_INVALID, "assoc_off is not non-decreasing at problem %d", b);
            for (int64_t k = assoc_off[b]; k < assoc_off[b + 1]; ++k)
                if (assoc[2 * k] < 0 || assoc[2 * k] >= n1[b] || assoc[2 * k + 1] < 0 || assoc[2 * k + 1] >= n2[b])
                    return fail(c, ROMAN_E_INVALID, "problem %d: association %lld = (%d,%d) out of range", b, (long long)(k - assoc_off[b]), assoc[2 * k], assoc[2 * k + 1]);
        }
        const int64_t na = assoc ? (assoc_off[b + 1] - assoc_off[b]) : 0;
        sumA += na > 0 ? na : (int64_t)n1[b] * n2[b];      // an empty list means all-to-all
    }
    const size_t fbytes = sizeof(double) * (size_t)std::max<int64_t>(n_objects * F, 1);
    HIPCHK(c, WS.hFeats.ensure(fbytes));
    if (n_objects * F > 0) HIPCHK(c, hipMemcpyAsync(WS.hFeats.p, feats, sizeof(double) * (size_t)(n_objects * F), hipMemcpyHostToDevice, WS.stream));
    const int32_t* dA = nullptr;
    if (assoc) {
        const int64_t rows = assoc_off[B];
        HIPCHK(c, WS.hAssoc.ensure(sizeof(int32_t) * 2 * (size_t)std::max<int64_t>(rows, 1)));
        if (rows > 0) HIPCHK(c, hipMemcpyAsync(WS.hAssoc.p, assoc, sizeof(int32_t) * 2 * (size_t)rows, hipMemcpyHostToDevice, WS.stream));
        dA = WS.hAssoc.as<int32_t>();
    }
    const double* dU0 = nullptr;
    if (u0) {
        HIPCHK(c, WS.hU0.ensure(sizeof(double) * (size_t)std::max<int64_t>(sumA, 1)));
        if (sumA > 0) HIPCHK(c, hipMemcpyAsync(WS.hU0.p, u0, sizeof(double) * (size_t)sumA, hipMemcpyHostToDevice, WS.stream));
        dU0 = WS.hU0.as<double>();
    }
    const BatchIn in{B, WS.hFeats.as<double>(), off1, n1, off2, n2, F, dA, assoc_off};
    return align_to_host(c, D, params, in, dU0, kmax, assoc_out, n_assoc_out, T_out, status_out, stats_out);
}

/* roman_align_batch_resident: inputs in HBM (as roman_align_batch_dev), results on the HOST (as roman_align_batch). */
int roman_align_batch_resident(roman_ctx_t* c, const roman_params_t* params, int32_t B,
                               const double* feats, const int64_t* off1, const int32_t* n1,
                               const int64_t* off2, const int32_t* n2, int32_t F,
                               const int32_t* assoc, const int64_t* assoc_off, const double* u0,
                               int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out,
                               double* T_out, int32_t* status_out, roman_stats_t* stats_out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (B < 0 || F < 0) return fail(c, ROMAN_E_INVALID, "negative size");
    if (B == 0) return ROMAN_OK;
    if (!off1 || !n1 || !off2 || !n2 || !assoc_out || !n_assoc_out || !T_out || !status_out || kmax < 0)
        return fail(c, ROMAN_E_INVALID, "NULL metadata/output pointer or kmax < 0");
    if (assoc && !assoc_off) return fail(c, ROMAN_E_INVALID, "assoc given without assoc_off");
    if (assoc && assoc_off[0] != 0) return fail(c, ROMAN_E_INVALID, "assoc_off[0] must be 0");
    bool any = false;
    for (int b = 0; b < B; ++b) {
        if (n1[b] < 0 || n2[b] < 0 || off1[b] < 0 || off2[b] < 0) return fail(c, ROMAN_E_INVALID, "problem %d: negative size or offset", b);
        if (assoc && assoc_off[b + 1] < assoc_off[b]) return fail(c, ROMAN_E_INVALID, "assoc_off is not non-decreasing at problem %d", b);
        any = any || n1[b] > 0 || n2[b] > 0;
    }
    if (!feats && any) return fail(c, ROMAN_E_INVALID, "feats is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    DevParams D;
    int rc = make_dev_params(c, params, F, &D);
    if (rc) return rc;
    const BatchIn in{B, feats, off1, n1, off2, n2, F, assoc, assoc_off};
    return align_to_host(c, D, params, in, u0, kmax, assoc_out, n_assoc_out, T_out, status_out, stats_out);
}

// --- the deal of a batch over ranks (pure host function; roman_amd.align.distributed.deal_by_cost states the same) ------------
int roman_deal_problems(int32_t B, const int32_t* n1, const int32_t* n2, const int64_t* assoc_off,
                        int32_t world, int32_t rank, int32_t* idx_out, int32_t* n_out)
{
    if (B < 0 || world < 1 || rank < 0 || rank >= world || !n_out || (B > 0 && (!n1 || !n2 || !idx_out)))
        return fail(nullptr, ROMAN_E_INVALID, "bad arguments");
    std::vector<int64_t> work((size_t)B);
    for (int b = 0; b < B; ++b) {
        int64_t a = assoc_off ? assoc_off[b + 1] - assoc_off[b] : 0;
        if (a <= 0) a = (int64_t)n1[b] * n2[b];                 // no list, or an empty one: all-to-all
        work[(size_t)b] = a * a;
    }
    std::vector<int32_t> order((size_t)B);
    for (int b = 0; b < B; ++b) order[(size_t)b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return work[(size_t)x] > work[(size_t)y]; });
    std::vector<int64_t> load((size_t)world, 0), count((size_t)world, 0);
    std::vector<int32_t> mine;
    for (int32_t b : order) {
        int r = 0;
        for (int t = 1; t < world; ++t)
            if (load[(size_t)t] < load[(size_t)r] || (load[(size_t)t] == load[(size_t)r] && count[(size_t)t] < count[(size_t)r])) r = t;
        load[(size_t)r] += std::max<int64_t>(work[(size_t)b], 1); count[(size_t)r] += 1;
        if (r == rank) mine.push_back(b);
    }
    std::sort(mine.begin(), mine.end());
    for (size_t t = 0; t < mine.size(); ++t) idx_out[t] = mine[t];
    *n_out = (int32_t)mine.size();
    return ROMAN_OK;
}

// --- stepwise surface for the clipperpy-compatible shim -----------------------------------------------
int roman_create_all_to_all(int32_t n1, int32_t n2, int32_t* out)
{
    if (n1 < 0 || n2 < 0 || (!out && (int64_t)n1 * n2 > 0)) return fail(nullptr, ROMAN_E_INVALID, "bad arguments");
    for (int32_t i = 0; i < n1; ++i) for (int32_t j = 0; j < n2; ++j) { out[2 * ((int64_t)i * n2 + j)] = i; out[2 * ((int64_t)i * n2 + j) + 1] = j; }
    return ROMAN_OK;
}

int roman_score(roman_ctx_t* c, const roman_params_t* params, const double* D1, int32_t n1,
                const double* D2, int32_t n2, int32_t F, const int32_t* assoc, int32_t n_assoc)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (n1 < 0 || n2 < 0 || F < 0 || (assoc && n_assoc < 0)) return fail(c, ROMAN_E_INVALID, "negative size");
    if ((n1 > 0 && !D1) || (n2 > 0 && !D2)) return fail(c, ROMAN_E_INVALID, "NULL feature matrix");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    roman_ctx::Last& Lst = c->last;
    Lst.scored = false; Lst.solved = false; Lst.dense = false; Lst.hascz = false;
    DevParams D0;
    int rc = make_dev_params(c, params, F, &D0);
    if (rc) return rc;
    const int64_t nobj = (int64_t)n1 + n2;
    HIPCHK(c, WS.hFeats.ensure(sizeof(double) * (size_t)std::max<int64_t>(nobj * F, 1)));
    if ((int64_t)n1 * F > 0) HIPCHK(c, hipMemcpyAsync(WS.hFeats.p, D1, sizeof(double) * (size_t)n1 * F, hipMemcpyHostToDevice, WS.stream));
    if ((int64_t)n2 * F > 0) HIPCHK(c, hipMemcpyAsync(WS.hFeats.as<double>() + (size_t)n1 * F, D2, sizeof(double) * (size_t)n2 * F, hipMemcpyHostToDevice, WS.stream));
    const int32_t* dA = nullptr;
    if (assoc && n_assoc == 0) assoc = nullptr;        // clipperpy: an empty A is replaced by the all-to-all list
    int64_t aoff[2] = {0, n_assoc};
    Lst.assoc.clear();
    if (assoc) {
        for (int32_t k = 0; k < n_assoc; ++k)
            if (assoc[2 * k] < 0 || assoc[2 * k] >= n1 || assoc[2 * k + 1] < 0 || assoc[2 * k + 1] >= n2)
                return fail(c, ROMAN_E_INVALID, "association %d = (%d,%d) out of range", k, assoc[2 * k], assoc[2 * k + 1]);
        Lst.assoc.assign(assoc, assoc + 2 * (size_t)n_assoc);
        HIPCHK(c, WS.hAssoc.ensure(sizeof(int32_t) * 2 * (size_t)std::max(n_assoc, 1)));
        if (n_assoc > 0) HIPCHK(c, hipMemcpyAsync(WS.hAssoc.p, assoc, sizeof(int32_t) * 2 * (size_t)n_assoc, hipMemcpyHostToDevice, WS.stream));
        dA = WS.hAssoc.as<int32_t>();
    }
    const int64_t o1 = 0, o2 = n1;
    const BatchIn in{1, WS.hFeats.as<double>(), &o1, &n1, &o2, &n2, F, dA, aoff};
    std::vector<ProbDesc> hd;
    ProbState ps{};
    for (int attempt = 0; ; ++attempt) {
        rc = enqueue_score(c, D0, params, in, hd, &Lst.D);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(&ps, WS.state.p, sizeof(ProbState), hipMemcpyDeviceToHost, WS.stream));
        HIPCHK(c, hipStreamSynchronize(WS.stream));
        if (!batch_overflowed(c)) break;
        if (attempt + 1 >= MAX_ATTEMPTS) return fail(c, ROMAN_E_NOMEM, "the sparse workspace still does not fit after %d attempts", attempt + 1);
    }
    Lst.pd = hd[0]; Lst.nA = hd[0].nA; Lst.L = ps.L; Lst.kind = ps.kind; Lst.nnzCap = ps.nnzCap; Lst.scored = true;
    return ROMAN_OK;
}

int roman_set_matrix_data(roman_ctx_t* c, const roman_params_t* params, const double* M, const double* Cm, int32_t n)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!M || !Cm))) return fail(c, ROMAN_E_INVALID, "bad matrix arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    roman_ctx::Last& Lst = c->last;
    Lst.scored = false; Lst.solved = false; Lst.dense = true; Lst.assoc.clear();
    roman_params_t p = *params; p.invariant = ROMAN_INV_EUCLIDEAN;          // implicit identity diagonal
    int rc = make_dev_params(c, &p, p.point_dim, &Lst.D);
    if (rc) return rc;
    // The weights decide the solver.  The stream solver adds FIXED-POINT terms rint(v x 2^s) (kernels.hip.h, "Exact
    // accumulation"): a term stays below 2^50 and a row's sum below 2^62 only for 0 <= v <= 1 (what every scored matrix
    // holds by construction), and its accumulators are decoded as unsigned.  A caller's dense M may hold anything: the
    // strict upper triangle is scanned on the device first, and a matrix with a weight outside [0, 1] (or not finite) takes
    // the plain-double solvers of the fallback layout instead.
    const size_t n1_ = (size_t)std::max(n, 1);
    HIPCHK(c, WS.hAux1.ensure(sizeof(double) * n1_ * n1_)); HIPCHK(c, WS.hAux2.ensure(sizeof(double) * n1_ * n1_)); HIPCHK(c, WS.hAux3.ensure(sizeof(int) * 4));
    HIPCHK(c, hipMemsetAsync(WS.hAux3.p, 0, sizeof(int) * 4, WS.stream));
    int rangeFlag = 0;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(WS.hAux1.p, M, sizeof(double) * n1_ * n1_, hipMemcpyHostToDevice, WS.stream));
        HIPCHK(c, hipMemcpyAsync(WS.hAux2.p, Cm, sizeof(double) * n1_ * n1_, hipMemcpyHostToDevice, WS.stream));
        hipLaunchKernelGGL(k_dense_range, dim3((unsigned)std::min(c->num_cu * 8, (n + 3) / 4)), dim3(256), 0, WS.stream, n, WS.hAux1.as<double>(), WS.hAux3.as<int>());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&rangeFlag, WS.hAux3.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, WS.stream));
        HIPCHK(c, hipStreamSynchronize(WS.stream));
    }
    const bool iter = Lst.D.p.maxiniters >= 1 && Lst.D.p.maxlsiters >= 1;
    const bool up = n <= STREAM_MAXL && iter && rangeFlag == 0;
    // few large problems: the whole-device solver; a problem the range check keeps off the stream solver is small enough for
    // one workgroup (k_solve: u and u' in LDS)
    Lst.D.wide = (!up && n > STREAM_MAXL && c->coop_ok && iter && n <= (int64_t)WIDE_KW * c->num_cu * WIDE_NW * 64) ? 1 : 0;
    Lst.D.idx16 = 0;                                           // dense problems may carry C flags: 32-bit labels
    Lst.D.stream_maxL = up ? std::max(64, (n + 63) & ~63) : 64;
    // The conversion runs on the device (kernels.hip.h, "Dense problems"): upload M and C, candidate bit matrix, then the
    // scored path's own layout kernels.  Two small read-backs size the matrix pools (the slot total is known only after the
    // sort) and fetch the C-flag / capacity status.
    const int W = (n + 63) / 64;
    const int RPB = 128;
    const size_t maskWords = std::max<size_t>((size_t)n * (size_t)W, 1);
    const long long capList = up ? (long long)n * (n - 1) / 2 + 4LL * n + 4 : 4;
    HIPCHK(c, WS.probs.ensure(sizeof(ProbDesc))); HIPCHK(c, WS.state.ensure(sizeof(ProbState))); HIPCHK(c, WS.totals.ensure(sizeof(BatchTotals)));
    HIPCHK(c, WS.queue.ensure(sizeof(int) * 16));
    {
        DevBuf* i32s[] = {&WS.lp, &WS.plp, &WS.rowCnt, &WS.rowPos, &WS.perm, &WS.sliceWidth, &WS.sliceBase, &WS.listOff};
        for (DevBuf* b_ : i32s) HIPCHK(c, b_->ensure(sizeof(int32_t) * n1_));
        DevBuf* f64s[] = {&WS.ls, &WS.ld, &WS.pld};
        for (DevBuf* b_ : f64s) HIPCHK(c, b_->ensure(sizeof(double) * n1_));
    }
    HIPCHK(c, WS.maskPool.ensure(sizeof(unsigned long long) * maskWords)); HIPCHK(c, WS.prefPool.ensure(sizeof(uint32_t) * maskWords));
    HIPCHK(c, WS.listPool.ensure(sizeof(uint16_t) * (size_t)capList));
    HIPCHK(c, WS.items.ensure(sizeof(ItemDesc) * ((size_t)n / RPB + 2)));
    WS.capMaskWords = std::max<long long>(WS.capMaskWords, (long long)maskWords); WS.capList = std::max<long long>(WS.capList, capList);
    ProbDesc pd{}; pd.off1 = 0; pd.off2 = 0; pd.assocOff = -1; pd.liveOff = 0; pd.n1 = n; pd.n2 = 1; pd.nA = n;
    ProbState ps{}; ps.L = n; ps.kind = up ? 0 : 1;
    HIPCHK(c, hipMemcpyAsync(WS.probs.p, &pd, sizeof(pd), hipMemcpyHostToDevice, WS.stream));
    HIPCHK(c, hipMemcpyAsync(WS.state.p, &ps, sizeof(ps), hipMemcpyHostToDevice, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));                  // (pd, ps live on this frame)
    const ProbDesc* dP = WS.probs.as<ProbDesc>(); ProbState* dS = WS.state.as<ProbState>(); BatchTotals* dT = WS.totals.as<BatchTotals>();
    const double* dM = WS.hAux1.as<double>(); const double* dC = WS.hAux2.as<double>();
    if (n > 0) {
        std::vector<int32_t> ident(n1_); std::vector<double> ones(n1_, 1.0);
        for (int k = 0; k < n; ++k) ident[(size_t)k] = k;
        HIPCHK(c, hipMemcpy(WS.lp.p, ident.data(), sizeof(int32_t) * n1_, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(WS.plp.p, ident.data(), sizeof(int32_t) * n1_, hipMemcpyHostToDevice));      // (stream layout: k_upper overwrites it with the permutation)
        HIPCHK(c, hipMemcpy(WS.ls.p, ones.data(), sizeof(double) * n1_, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(WS.ld.p, ones.data(), sizeof(double) * n1_, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(WS.pld.p, ones.data(), sizeof(double) * n1_, hipMemcpyHostToDevice));
    }
    const LivePools LP{WS.lp.as<int32_t>(), nullptr, nullptr, WS.ls.as<double>(), WS.ld.as<double>(), nullptr, nullptr};
    const LivePools PP{WS.plp.as<int32_t>(), nullptr, nullptr, nullptr, WS.pld.as<double>(), nullptr, nullptr};
    hipLaunchKernelGGL(k_rowbase, dim3(1), dim3(256), 0, WS.stream, 1, RPB, (long long)maskWords, dS, dT, 0, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_items, dim3(1), dim3(256), 0, WS.stream, RPB, dS, WS.items.as<ItemDesc>());
    if (n > 0) {
        hipLaunchKernelGGL(k_dense_mask, dim3((unsigned)std::min(c->num_cu * 8, (n + 3) / 4)), dim3(256), 0, WS.stream, n, dM, dC,
                           WS.maskPool.as<unsigned long long>(), WS.hAux3.as<int>());
        hipLaunchKernelGGL(k_rowprefix, dim3(c->num_cu * 2), dim3(1024), 0, WS.stream, dP, dS, dT, WS.items.as<ItemDesc>(),
                           WS.maskPool.as<unsigned long long>(), WS.prefPool.as<uint32_t>(), WS.rowCnt.as<uint32_t>(), RPB, 0);
        hipLaunchKernelGGL(k_rowsort, dim3(1), dim3(1024), 0, WS.stream, dP, dS, dT, WS.rowCnt.as<uint32_t>(), WS.rowPos.as<uint32_t>(), WS.perm.as<uint32_t>(),
                           WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.listOff.as<uint32_t>(), capList, 0, sort_eq_max());
        if (up) {
            hipLaunchKernelGGL(k_upper, dim3(c->num_cu * 2), dim3(1024), 0, WS.stream, dP, dS, dT, WS.items.as<ItemDesc>(),
                               WS.maskPool.as<unsigned long long>(), WS.listPool.as<uint16_t>(), WS.listOff.as<uint32_t>(),
                               WS.rowCnt.as<uint32_t>(), WS.perm.as<uint32_t>(), WS.rowPos.as<uint32_t>(), LP, PP, RPB);
            hipLaunchKernelGGL(k_slicegeom, dim3(1), dim3(64), 0, WS.stream, dP, dS, WS.rowCnt.as<uint32_t>(), WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>());
        }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&ps, WS.state.p, sizeof(ps), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    if (ps.kind == 2) return fail(c, ROMAN_E_NOMEM, "internal: dense problem does not fit its own workspace");
    const uint64_t total = ps.nnzCap;
    if (total > 4000000000ull) return fail(c, ROMAN_E_TOO_LARGE, "dense matrix has too many non-zeros");
    const size_t nnz1 = (size_t)std::max<uint64_t>(total, 1);
    HIPCHK(c, WS.vals.ensure(sizeof(double) * nnz1)); HIPCHK(c, WS.cols16.ensure(sizeof(uint16_t) * nnz1)); HIPCHK(c, WS.cols32.ensure(sizeof(uint32_t) * nnz1));
    WS.capNnz = std::max<long long>(WS.capNnz, (long long)nnz1);
    hipLaunchKernelGGL(k_probscan, dim3(1), dim3(256), 0, WS.stream, 1, 1, (long long)nnz1, dS, dT);
    if (total > 0) {
        const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)c->num_cu * 16);
        auto kf = up ? k_dense_fill<0> : k_dense_fill<1>;
        hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, WS.stream, n, dM, dC, dS, WS.rowCnt.as<uint32_t>(), WS.rowPos.as<uint32_t>(), WS.perm.as<uint32_t>(),
                           WS.sliceWidth.as<uint32_t>(), WS.sliceBase.as<uint32_t>(), WS.listPool.as<uint16_t>(), WS.listOff.as<uint32_t>(),
                           WS.maskPool.as<unsigned long long>(), WS.prefPool.as<uint32_t>(), WS.cols16.as<uint16_t>(), WS.cols32.as<uint32_t>(), WS.vals.as<double>());
    }
    HIPCHK(c, hipGetLastError());
    int flags[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(&ps, WS.state.p, sizeof(ps), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipMemcpyAsync(flags, WS.hAux3.p, sizeof(flags), hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    if (ps.kind == 2) return fail(c, ROMAN_E_NOMEM, "internal: dense problem does not fit its own workspace");
    Lst.hascz = flags[0] != 0;
    Lst.pd = pd; Lst.nA = n; Lst.L = n; Lst.kind = ps.kind; Lst.nnzCap = (int64_t)total;
    Lst.scored = true;
    return ROMAN_OK;
}

int roman_solve(roman_ctx_t* c, const double* u0)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (!c->last.scored) return fail(c, ROMAN_E_INVALID, "roman_solve: no matrices (call roman_score or roman_set_matrix_data first)");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    return solve_last(c, u0);
}

int roman_num_associations(const roman_ctx_t* c, int32_t* n)
{
    if (!c || !n) return fail(nullptr, ROMAN_E_INVALID, "NULL argument");
    if (!c->last.scored) return fail(const_cast<roman_ctx*>(c), ROMAN_E_INVALID, "nothing scored yet");
    *n = c->last.nA; return ROMAN_OK;
}
int roman_num_selected(const roman_ctx_t* c, int32_t* n)
{
    if (!c || !n) return fail(nullptr, ROMAN_E_INVALID, "NULL argument");
    if (!c->last.solved) return fail(const_cast<roman_ctx*>(c), ROMAN_E_INVALID, "nothing solved yet");
    *n = c->last.nsel; return ROMAN_OK;
}
int roman_get_selected_associations(const roman_ctx_t* c, int32_t* out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    const roman_ctx::Last& L = c->last;
    if (!L.solved) return fail(const_cast<roman_ctx*>(c), ROMAN_E_INVALID, "nothing solved yet");
    for (int32_t t = 0; t < L.nsel; ++t) {
        const int32_t p = L.nodes[(size_t)t];
        if (!L.assoc.empty()) { out[2 * t] = L.assoc[2 * (size_t)p]; out[2 * t + 1] = L.assoc[2 * (size_t)p + 1]; }
        else if (L.dense) { out[2 * t] = p; out[2 * t + 1] = p; }
        else { out[2 * t] = p / L.pd.n2; out[2 * t + 1] = p % L.pd.n2; }
    }
    return ROMAN_OK;
}
int roman_get_solution(const roman_ctx_t* c, int32_t* nodes, double* u, double* score, roman_stats_t* stats)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    const roman_ctx::Last& L = c->last;
    if (!L.solved) return fail(const_cast<roman_ctx*>(c), ROMAN_E_INVALID, "nothing solved yet");
    if (nodes) for (int32_t t = 0; t < L.nsel; ++t) nodes[t] = L.nodes[(size_t)t];
    if (u) for (int32_t p = 0; p < L.nA; ++p) u[p] = L.u[(size_t)p];
    if (score) *score = L.stats.score;
    if (stats) *stats = L.stats;
    return ROMAN_OK;
}

int roman_get_upper_csr(const roman_ctx_t* c, int64_t* nnz, int64_t* rowptr, int32_t* cols_out, double* vals_out, double* diag)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    const roman_ctx::Last& Lst = c->last;
    if (!Lst.scored) return fail(const_cast<roman_ctx*>(c), ROMAN_E_INVALID, "nothing scored yet");
    std::vector<uint32_t> rs, rl, cols; std::vector<double> vals, ls; std::vector<int32_t> lp;
    int rc = fetch_last_csr(c, rs, rl, cols, vals, lp, ls);
    if (rc) return rc;
    const int L = Lst.L, nA = Lst.nA;
    int64_t cnt = 0;
    std::vector<int64_t> rp((size_t)nA + 1, 0);
    for (int k = 0; k < L; ++k) {
        int64_t rc_ = 0;
        for (uint32_t e = 0; e < rl[(size_t)k]; ++e) { const uint32_t q = cols[(size_t)rs[(size_t)k] + e] & 0x7fffffffu; if ((int)q > k) ++rc_; }
        rp[(size_t)lp[(size_t)k] + 1] = rc_; cnt += rc_;
    }
    for (int p = 0; p < nA; ++p) rp[(size_t)p + 1] += rp[(size_t)p];
    if (nnz) *nnz = cnt;
    if (rowptr) for (int p = 0; p <= nA; ++p) rowptr[p] = rp[(size_t)p];
    if (cols_out || vals_out) {
        for (int k = 0; k < L; ++k) {
            int64_t w = rp[(size_t)lp[(size_t)k]];
            for (uint32_t e = 0; e < rl[(size_t)k]; ++e) {
                const uint32_t q = cols[(size_t)rs[(size_t)k] + e] & 0x7fffffffu;
                if ((int)q > k) { if (cols_out) cols_out[w] = lp[(size_t)q]; if (vals_out) vals_out[w] = vals[(size_t)rs[(size_t)k] + e]; ++w; }
            }
        }
    }
    if (diag) {
        const bool single = Lst.D.single && !Lst.dense;
        for (int p = 0; p < nA; ++p) diag[p] = single ? 0.0 : 1.0;
        if (single) for (int k = 0; k < L; ++k) diag[(size_t)lp[(size_t)k]] = ls[(size_t)k];
    }
    return ROMAN_OK;
}

int roman_get_dense_matrices(const roman_ctx_t* c, double* M, double* Cm)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    const roman_ctx::Last& Lst = c->last;
    if (!Lst.scored) return fail(const_cast<roman_ctx*>(c), ROMAN_E_INVALID, "nothing scored yet");
    std::vector<uint32_t> rs, rl, cols; std::vector<double> vals, ls; std::vector<int32_t> lp;
    int rc = fetch_last_csr(c, rs, rl, cols, vals, lp, ls);
    if (rc) return rc;
    const int L = Lst.L; const int64_t nA = Lst.nA;
    if (M) memset(M, 0, sizeof(double) * (size_t)(nA * nA));
    if (Cm) memset(Cm, 0, sizeof(double) * (size_t)(nA * nA));
    const bool single = Lst.D.single && !Lst.dense;
    for (int64_t p = 0; p < nA; ++p) { if (M) M[p * nA + p] = single ? 0.0 : 1.0; if (Cm) Cm[p * nA + p] = 1.0; }
    for (int k = 0; k < L; ++k) {
        const int64_t p = lp[(size_t)k];
        if (single && M) M[p * nA + p] = ls[(size_t)k];
        for (uint32_t e = 0; e < rl[(size_t)k]; ++e) {
            const uint32_t cq = cols[(size_t)rs[(size_t)k] + e];
            const int64_t q = lp[(size_t)(cq & 0x7fffffffu)];
            if (M) M[p * nA + q] = vals[(size_t)rs[(size_t)k] + e];
            if (Cm) Cm[p * nA + q] = (cq & 0x80000000u) ? 0.0 : 1.0;
        }
    }
    return ROMAN_OK;
}

// --- pose from given correspondences -----------------------------------------------------------------
int roman_pose_batch(roman_ctx_t* c, int32_t dim, int32_t B, const double* pts1, const double* pts2,
                     const int64_t* corr_off, double* T_out, int32_t* status_out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (dim != 2 && dim != 3) return fail(c, ROMAN_E_INVALID, "dim must be 2 or 3");
    if (B < 0 || !corr_off || !T_out || !status_out) return fail(c, ROMAN_E_INVALID, "bad arguments");
    if (B == 0) return ROMAN_OK;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    const int64_t K = corr_off[B];
    if (K < 0 || (K > 0 && (!pts1 || !pts2))) return fail(c, ROMAN_E_INVALID, "bad correspondence arrays");
    HIPCHK(c, WS.hAux1.ensure(sizeof(double) * (size_t)std::max<int64_t>(K * dim, 1)));
    HIPCHK(c, WS.hAux2.ensure(sizeof(double) * (size_t)std::max<int64_t>(K * dim, 1)));
    HIPCHK(c, WS.hAux3.ensure(sizeof(int64_t) * ((size_t)B + 1)));
    HIPCHK(c, WS.oT.ensure(sizeof(double) * 16 * (size_t)B)); HIPCHK(c, WS.oStatus.ensure(sizeof(int32_t) * (size_t)B));
    if (K > 0) {
        HIPCHK(c, hipMemcpyAsync(WS.hAux1.p, pts1, sizeof(double) * (size_t)(K * dim), hipMemcpyHostToDevice, WS.stream));
        HIPCHK(c, hipMemcpyAsync(WS.hAux2.p, pts2, sizeof(double) * (size_t)(K * dim), hipMemcpyHostToDevice, WS.stream));
    }
    HIPCHK(c, hipMemcpyAsync(WS.hAux3.p, corr_off, sizeof(int64_t) * ((size_t)B + 1), hipMemcpyHostToDevice, WS.stream));
    hipLaunchKernelGGL(k_pose, dim3(B), dim3(64), 0, WS.stream, dim, WS.hAux1.as<double>(), WS.hAux2.as<double>(), WS.hAux3.as<int64_t>(),
                       WS.oT.as<double>(), WS.oStatus.as<int32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(T_out, WS.oT.p, sizeof(double) * 16 * (size_t)B, hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipMemcpyAsync(status_out, WS.oStatus.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    return ROMAN_OK;
}

// --- diagnostics -------------------------------------------------------------------------------------------
int roman_debug_math(roman_ctx_t* c, int kind, const double* in1, const double* in2, int64_t n, double* out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!in1 || !out))) return fail(c, ROMAN_E_INVALID, "bad arguments");
    if (n == 0) return ROMAN_OK;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    HIPCHK(c, WS.hAux1.ensure(sizeof(double) * (size_t)n)); HIPCHK(c, WS.hAux2.ensure(sizeof(double) * (size_t)n)); HIPCHK(c, WS.hAux3.ensure(sizeof(double) * (size_t)n));
    HIPCHK(c, hipMemcpyAsync(WS.hAux1.p, in1, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, WS.stream));
    if (in2) HIPCHK(c, hipMemcpyAsync(WS.hAux2.p, in2, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, WS.stream));
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, WS.stream, kind, WS.hAux1.as<double>(),
                       in2 ? WS.hAux2.as<double>() : (const double*)nullptr, n, WS.hAux3.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, WS.hAux3.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    return ROMAN_OK;
}

int roman_debug_cosine(roman_ctx_t* c, const roman_params_t* params, const double* D1, int32_t n1,
                       const double* D2, int32_t n2, int32_t F, double* out)
{
    if (!c) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    if (n1 <= 0 || n2 <= 0 || !D1 || !D2 || !out) return fail(c, ROMAN_E_INVALID, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = use_ws0(c); if (rc0) return rc0; }
    DevParams D;
    int rc = make_dev_params(c, params, F, &D);
    if (rc) return rc;
    if (D.p.cos_feature_dim <= 0) return fail(c, ROMAN_E_INVALID, "cos_feature_dim is 0");
    HIPCHK(c, WS.hFeats.ensure(sizeof(double) * (size_t)(n1 + n2) * F));
    HIPCHK(c, hipMemcpyAsync(WS.hFeats.p, D1, sizeof(double) * (size_t)n1 * F, hipMemcpyHostToDevice, WS.stream));
    HIPCHK(c, hipMemcpyAsync(WS.hFeats.as<double>() + (size_t)n1 * F, D2, sizeof(double) * (size_t)n2 * F, hipMemcpyHostToDevice, WS.stream));
    ProbDesc pd{}; pd.off1 = 0; pd.off2 = n1; pd.assocOff = -1; pd.n1 = n1; pd.n2 = n2; pd.nA = n1 * n2;
    HIPCHK(c, WS.probs.ensure(sizeof(ProbDesc)));
    HIPCHK(c, WS.cosPool.ensure(sizeof(double) * (size_t)n1 * n2));
    HIPCHK(c, hipMemcpyAsync(WS.probs.p, &pd, sizeof(pd), hipMemcpyHostToDevice, WS.stream));
    // ROMAN_COS_SEL=approx / gated (tests): k_cos_sel's screen matrix / k_cos_sel's matrix (exact where cos >= cosine_min - 2^-6, the screen elsewhere)
    const char* selEnv = getenv("ROMAN_COS_SEL");
    if (selEnv && (!strcmp(selEnv, "approx") || !strcmp(selEnv, "gated")) && n1 <= CSEL_MAXN && n2 <= CSEL_MAXN) {
        HIPCHK(c, WS.cosDense.ensure(sizeof(int32_t)));
        HIPCHK(c, launch_cos_sel(c, WS.stream, D, 1, n1, n2, WS.probs.as<ProbDesc>(), WS.hFeats.as<double>(), WS.cosPool.as<double>(), WS.cosDense.as<int32_t>(), selEnv[0] == 'a'));
    } else
    HIPCHK(c, launch_cos(c, WS.stream, D, 1, n1, n2, WS.probs.as<ProbDesc>(), WS.hFeats.as<double>(), WS.cosPool.as<double>()));
    HIPCHK(c, hipMemcpyAsync(out, WS.cosPool.p, sizeof(double) * (size_t)n1 * n2, hipMemcpyDeviceToHost, WS.stream));
    HIPCHK(c, hipStreamSynchronize(WS.stream));
    c->last.scored = false; c->last.solved = false;
    return ROMAN_OK;
}

int roman_debug_live(const roman_ctx_t* cc, int32_t* n_live, int32_t* idx, double* score)
{
    if (!cc) return fail(nullptr, ROMAN_E_INVALID, "ctx is NULL");
    roman_ctx* c = const_cast<roman_ctx*>(cc);
    if (!c->last.scored) return fail(c, ROMAN_E_INVALID, "nothing scored yet");
    const int L = c->last.L;
    if (n_live) *n_live = L;
    HIPCHK(c, hipSetDevice(c->device));
    if (idx && L > 0) HIPCHK(c, hipMemcpy(idx, WS.lp.p, sizeof(int32_t) * (size_t)L, hipMemcpyDeviceToHost));
    if (score && L > 0) HIPCHK(c, hipMemcpy(score, WS.ls.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost));
    return ROMAN_OK;
}

}  // extern "C"
