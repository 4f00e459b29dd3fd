/*
 * roman_hip.h — C ABI of libroman_hip.so, the MI355X (gfx950) replacement for the
 * `clipperpy` native module that mit-acl/roman's `roman.align` hot path binds.
 *
 * Every entry point names the reference interface it replaces ([REF file:line] is relative
 * to the reference checkout).  Plain pointers and sizes only: no torch / numpy / C++ types
 * cross this boundary.  All functions return 0 on success or a negative ROMAN_E_* code; they
 * never throw.  The library fails (ROMAN_E_NO_DEVICE) when no HIP device is present — there
 * is no CPU fallback behind this ABI.
 *
 * Data conventions (same as the reference's calls into clipperpy):
 *   - A "feature matrix" is what [REF roman/align/roman_registration.py:91-95] hands to
 *     `score_pairwise_and_single_consistency` as `map_cl.T`: F x n float64, one COLUMN per
 *     object, column-major == object-major: object o's F features are the F contiguous
 *     doubles at feats[o*F .. o*F+F).  Row layout inside a column is
 *     [x y (z)] ++ ratio features (ratio_feature_dim) ++ cosine features (cos_feature_dim)
 *     [REF roman/align/roman_registration.py:98-108].
 *   - An association list is (A,2) int32 row-major; column 0 indexes map 1, column 1 map 2
 *     [REF roman/align/object_registration.py:110-111].  A NULL list means all-to-all in
 *     the order of clipperpy.utils.create_all_to_all: row i*n2+j = (i,j)
 *     [REF roman/align/object_registration.py:41].
 *   - A pose is a row-major (dim+1)x(dim+1) float64 matrix mapping map 2 into map 1
 *     [REF roman/align/object_registration.py:88-129]; always stored in 16 doubles.
 */
#ifndef ROMAN_HIP_H
#define ROMAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported from libroman_hip.so (the library is built with -fvisibility=hidden) */
#if defined(__GNUC__)
#define ROMAN_API __attribute__((visibility("default")))
#else
#define ROMAN_API
#endif

#define ROMAN_MAX_RATIO_FEATURES 8

/* error codes */
#define ROMAN_OK                0
#define ROMAN_E_INVALID        -1   /* bad argument (NULL pointer, dim not 2/3, ...)            */
#define ROMAN_E_NO_DEVICE      -2   /* no HIP device / wrong architecture                        */
#define ROMAN_E_HIP            -3   /* a HIP runtime call failed (see roman_last_error)          */
#define ROMAN_E_NOMEM          -4   /* device or host allocation failed                          */
#define ROMAN_E_UNSUPPORTED    -5   /* parameter combination the reference does not define       */
#define ROMAN_E_TOO_LARGE      -6   /* problem exceeds the index width of this build             */
#define ROMAN_E_INTERNAL       -7   /* a problem came back with ROMAN_ST_INTERNAL (host-pointer / stepwise entry points; the
                                       outputs have been copied: status_out says which problems have no result)           */

/* per-problem status written to status_out[] (bit flags) */
#define ROMAN_ST_OK                  0
#define ROMAN_ST_EMPTY_MAP           1  /* n1==0 or n2==0: register() returns (1,0) array [REF object_registration.py:23-24] */
#define ROMAN_ST_INSUFFICIENT        2  /* fewer than `dim` associations: T_align raises InsufficientAssociationsException [REF object_registration.py:107-108] */
#define ROMAN_ST_MAXITER             4  /* solver stopped on maxoliters                          */
#define ROMAN_ST_ASSOC_TRUNCATED     8  /* more selected associations than kmax (output clipped) */
#define ROMAN_ST_TIE_FALLBACK       16  /* top-omega boundary tie: sequential heap emulation ran */
#define ROMAN_ST_INTERNAL           64  /* internal error (a bounded device-side wait of the large-problem solver expired): the
                                           problem has no result; never expected, reported instead of hanging the device */
#define ROMAN_ST_WORKSPACE          32  /* the problem was SKIPPED: roman_align_batch_dev sizes its device pools before the
                                           sizes of the sparse matrices are known (from earlier batches; the call never
                                           waits for the GPU) and this problem did not fit.  The library has recorded the
                                           need: issue the problem again.  The host-pointer and stepwise entry points
                                           retry by themselves and never report this flag. */

/* which invariant scores association pairs */
#define ROMAN_INV_EUCLIDEAN  0  /* clipperpy.invariants.EuclideanDistance + clipperpy.CLIPPER
                                   [REF roman/align/dist_reg_with_pruning.py:48-57]            */
#define ROMAN_INV_ROMAN      1  /* clipperpy.invariants.ROMAN + clipperpy.CLIPPERPairwiseAndSingle
                                   [REF roman/align/roman_registration.py:82-86]               */
#define ROMAN_INV_EUCLIDEAN_PRUNED 2  /* DistRegWithPruning with its NumPy prefilter moved onto the device
                                   [REF roman/align/dist_reg_with_pruning.py:71-97]: EuclideanDistance + clipperpy.CLIPPER on the
                                   associations (i,j) of the all-to-all list that survive
                                       NOT (<desc_i, desc_j> < cosine_min)                       (raw dot product, :75-80)
                                       NOT (min(f_i, f_j) / max(f_i, f_j) < ratio_epsilon[f])    for each of the ratio features (:83-90)
                                   and on ALL of them when none survives (the reference then hands clipperpy an empty list, which
                                   means all-to-all, :94-96).  Feature row: [x y (z)] ++ ratio features (the reference uses volume,
                                   linearity, planarity, scattering) ++ descriptor.  Same result as ROMAN_INV_EUCLIDEAN on the
                                   explicit pruned list (the dot product is accumulated in the order of the f64 matrix core: a
                                   product within rounding of cosine_min may fall on the other side than NumPy's BLAS puts it). */

/* clipperpy.invariants.ROMAN.{GEOMETRIC_MEAN,ARITHMETIC_MEAN,PRODUCT}
   [REF roman/align/roman_registration.py:11-14] */
#define ROMAN_FUSE_GEOMETRIC_MEAN  0
#define ROMAN_FUSE_ARITHMETIC_MEAN 1
#define ROMAN_FUSE_PRODUCT         2

/* The two formulas of the ROMAN invariant that live only in the absent clipperpy sources (SURVEY.md
   Appendix B7, DESIGN.md decisions H2 and H3) are SWITCHES, not hard-wired guesses: 0 is the pinned
   default, the other values are the alternative readings.  The reference never sets them
   ([REF roman/align/roman_registration.py:55-78] lists every attribute it assigns); they exist so that a
   build that can import the real clipperpy can find the reading that reproduces it
   (tests/test_real_clipperpy.py).  With d = displacement between the two objects of a map, h = its
   horizontal length, v = its signed vertical component, l = its full length, unc = gravity_unc_ang_rad:
     ROMAN_GRAV_COMBINED  ch=|h1-h2|, cv=max(0,|v1-v2|-sin(unc)*max(h1,h2)), c=sqrt(ch^2+cv^2); c<epsilon; exp(-c^2/2sigma^2)
     ROMAN_GRAV_SEPARATE  same ch, cv, c; each part gated on its own: ch<epsilon AND cv<epsilon
     ROMAN_GRAV_ZGATE     c=|l1-l2| as for EuclideanDistance, plus the hard gate |v1-v2| < epsilon+sin(unc)*max(l1,l2) */
#define ROMAN_GRAV_COMBINED 0
#define ROMAN_GRAV_SEPARATE 1
#define ROMAN_GRAV_ZGATE    2
/*   ROMAN_SINGLE_BOTH     M_pq = fuse(s_a(p,q), s_o(p), s_o(q)), M_pp = s_o(p)
     ROMAN_SINGLE_OFFDIAG  M_pq fused as above, M_pp = 1 (plain CLIPPER's implicit identity)
     ROMAN_SINGLE_DIAG     M_pq = s_a(p,q), M_pp = s_o(p)
   In these three readings an association whose single score is 0 is removed from the problem (its row and column
   of M and C are empty, its u stays 0).
     ROMAN_SINGLE_DIAG_KEEP  M_pq = s_a(p,q), M_pp = s_o(p) — and NOTHING is removed: an association with s_o = 0 keeps
                             its off-diagonal entries and only has a zero diagonal (SURVEY.md B7 read literally: "diagonal
                             M_pp = s_single(p)", the pair score alone off the diagonal).  Every input association is then
                             live (L = A): the large-live-set path (k_solve_wide) serves it. */
#define ROMAN_SINGLE_BOTH      0
#define ROMAN_SINGLE_OFFDIAG   1
#define ROMAN_SINGLE_DIAG      2
#define ROMAN_SINGLE_DIAG_KEEP 3

/*
 * Invariant + solver parameters.  Replaces clipperpy.invariants.ROMANParams /
 * EuclideanDistanceParams (attributes set at [REF roman/align/roman_registration.py:55-78],
 * [REF roman/align/dist_reg_with_pruning.py:49-52]) and clipperpy.Params (always
 * default-constructed: [REF roman/align/roman_registration.py:84]).
 * roman_params_default() fills the defaults documented in DESIGN.md §"Pinned decisions".
 */
typedef struct roman_params {
    /* invariant */
    int32_t invariant;            /* ROMAN_INV_*                                              */
    int32_t point_dim;            /* 2 or 3                                                   */
    int32_t ratio_feature_dim;    /* <= ROMAN_MAX_RATIO_FEATURES                              */
    int32_t cos_feature_dim;      /* descriptor length d (0 = no semantics)                   */
    int32_t fusion_method;        /* ROMAN_FUSE_*                                             */
    int32_t gravity_guided;       /* 0/1; requires point_dim==3                               */
    int32_t drift_aware;          /* must be 0 (the reference always passes False)            */
    int32_t rescale_u0;           /* clipperpy.Params.rescale_u0 (default 1)                  */
    double  sigma;
    double  epsilon;
    double  mindist;
    double  distance_weight;
    double  ratio_weight;
    double  cosine_weight;
    double  cosine_min;
    double  cosine_max;
    double  gravity_unc_ang_rad;
    double  ratio_epsilon[ROMAN_MAX_RATIO_FEATURES];
    /* solver: clipperpy.Params */
    double  tol_u;                /* 1e-8  */
    double  tol_F;                /* 1e-9  */
    double  beta;                 /* 0.25  */
    double  eps;                  /* 1e-9  */
    double  affinityeps;          /* 1e-4  */
    int32_t maxiniters;           /* 200   */
    int32_t maxoliters;           /* 1000  */
    int32_t maxlsiters;           /* 99    */
    /* decision switches (see ROMAN_GRAV_*, ROMAN_SINGLE_* above); 0 = the pinned default */
    int32_t gravity_mode;         /* ROMAN_GRAV_*   : reading of the gravity-guided pair score (H2)   */
    int32_t single_mode;          /* ROMAN_SINGLE_* : where single scores enter M (H3)                */
    int32_t reserved;             /* must be 0                                                        */
} roman_params_t;

/* per-problem statistics (the quantities SURVEY.md §8(d) builds the roofline from) */
typedef struct roman_stats {
    int32_t n_assoc_in;    /* A: associations scored                                          */
    int32_t n_live;        /* L: associations with a non-zero single score (== A for EUCLIDEAN) */
    int64_t nnz_upper;     /* stored non-zeros of the strict upper triangle of M              */
    int32_t n_pass;        /* sparse matrix-vector passes over M the solver performed         */
    int32_t outer_iters;   /* homotopy (d-update) iterations                                  */
    int32_t inner_iters;   /* accepted projected-gradient steps                               */
    int32_t ls_trials;     /* line-search trials (each is one pass)                           */
    double  score;         /* F = u'Mu at exit (clipper.get_solution().score)                 */
    double  d_final;       /* final homotopy penalty                                          */
} roman_stats_t;

typedef struct roman_ctx roman_ctx_t;

/* ------------------------------------------------------------------------------------------- */
/* library / context                                                                           */
/* ------------------------------------------------------------------------------------------- */

/* Fill *p with the reference defaults: ROMAN invariant, dim 3, sigma .4, epsilon .6,
   mindist .2 [REF roman/params/submap_align_params.py:66-74], weights 1
   [REF roman/align/roman_registration.py:64-66], clipperpy.Params() defaults. */
ROMAN_API int roman_params_default(roman_params_t* p);

/* Create a context bound to HIP device `device`.  `stream` is a hipStream_t passed as void*
   (NULL = the library creates and owns its own non-blocking stream).  The context owns all
   device workspace; it grows on demand and is reused across calls.  One context per
   (device, stream); calls on one context must not overlap.  Replaces the per-call
   `clipperpy.CLIPPER*(invariant, params)` construction of
   [REF roman/align/object_registration.py:25]. */
ROMAN_API int roman_ctx_create(roman_ctx_t** ctx, int device, void* stream);
ROMAN_API int roman_ctx_destroy(roman_ctx_t* ctx);

/* Batches in flight.  depth 1 (default): every call runs on the context's stream and its results are
   complete once that stream is synchronised.  depth 2 ... 6: consecutive roman_align_batch_dev calls rotate
   over that many internal workspaces, each on an internal stream that starts behind the work already queued
   on the context's stream, so the straggler tail of one batch's kernels overlaps the next batch's
   affinity build (the reference's loop [REF roman/align/submap_align.py:93-200] has no dependency
   between pairs).  With depth >= 2 the results of a batch call are complete after roman_ctx_sync() (or a
   device-wide synchronisation), NOT after synchronising the context's stream alone; the caller must give
   batches that may be in flight together distinct output buffers if it needs both results.  The
   host-pointer and stepwise entry points always drain the pipeline first and run synchronously.
   roman_align_batch_dev is a pure enqueue at every depth (it never waits for the device and reads nothing
   back), so one host thread keeps all the batches in flight fed; no library threads exist.  Device buffers
   must stay valid until roman_ctx_sync(). */
ROMAN_API int roman_ctx_set_pipeline(roman_ctx_t* ctx, int depth);
/* Enqueue on the context's stream a wait for the pipelined batches issued so far: all of them, or —
   skip_latest != 0 — all but the most recent one, so that work queued on the caller's stream afterwards
   (e.g. the all_gather of batch k-1's records) sees their results while batch k keeps running.  The host
   does not block. */
ROMAN_API int roman_ctx_join(roman_ctx_t* ctx, int skip_latest);
/* The same wait enqueued on ANOTHER stream of the caller's (a hipStream_t; NULL = the context's stream — NOT the legacy default
   stream, whose handle is also 0: work queued on the null stream cannot be ordered through this call; use an explicit stream): the collective that
   gathers batch k-1's records then runs on a side stream and never sits between two batch calls on the context's stream —
   every batch call starts behind what is queued THERE, so a wait for batch k-1 on it would hold back batch k+1 and cost a
   batch in flight.  The caller orders the reuse of an output buffer against its own side stream (an event). */
ROMAN_API int roman_ctx_join_on(roman_ctx_t* ctx, int skip_latest, void* stream);
/* Wait for every batch in flight on this context (all internal streams and the context's stream). */
ROMAN_API int roman_ctx_sync(roman_ctx_t* ctx);
/* How roman_align_batch (host pointers) issues a LARGE batch: more than `chunk` problems (default 2048) go to the device as
   calls of `chunk` problems with `depth` of them in flight (default 3; 1 = one call for the whole batch, as before round 5) —
   the pipelined loop a device-pointer caller would write around roman_align_batch_dev, done by the library for the caller of
   the reference's serial loop [REF roman/align/submap_align.py:93-200] who hands over every surviving pair at once.  Problems
   a call skipped for workspace are issued again (those only); without a sizing history for the parameter block the first call
   is waited for before the others are queued.  The depth set with roman_ctx_set_pipeline is restored on return. */
ROMAN_API int roman_ctx_set_host_batching(roman_ctx_t* ctx, int chunk, int depth);
/* Team mode of the whole-device solver for LARGE live sets (methods without a semantic gate,
   [REF roman/params/submap_align_params.py:98-116]: every association live): -1 (default) the library decides — several such
   problems in a batch are solved side by side, the workgroups of an XCD (or of half an XCD) on one problem each —, 0 never (the
   whole device on one problem at a time), 1 / 2 / 4 teams per XCD whenever the live sets fit.  Teams are formed on the device
   from the XCD every workgroup really runs on, their buffers are sized on the host from the XCD count the runtime reports: a
   team that cannot hold its problem leaves it ROMAN_ST_INTERNAL; roman_align_batch runs such problems again with teams off, a
   caller of roman_align_batch_dev does the same through this setter. */
ROMAN_API int roman_ctx_set_wide_teams(roman_ctx_t* ctx, int teams_per_xcd);

/* Human-readable text of the last error on this context (or of the last context-less error
   when ctx == NULL).  The pointer stays valid until the next call on the same context. */
ROMAN_API const char* roman_last_error(const roman_ctx_t* ctx);

/* ------------------------------------------------------------------------------------------- */
/* the hot path, batched: score -> solve -> select -> pose for B independent submap pairs      */
/* ------------------------------------------------------------------------------------------- */

/*
 * roman_align_batch_dev: bulk data device-resident.
 *
 * Replaces, for each of the B problems, the reference sequence
 *     clipper.score_pairwise_and_single_consistency(D1, D2, A)   [REF roman/align/roman_registration.py:95]
 *  or clipper.score_pairwise_consistency(D1, D2, A)              [REF roman/align/object_registration.py:47]
 *     clipper.solve()                                            [REF roman/align/object_registration.py:27]
 *     clipper.get_selected_associations()                        [REF roman/align/object_registration.py:28]
 *     ObjectRegistration.T_align(map1, map2, associations)       [REF roman/align/object_registration.py:88-129]
 * i.e. the body of the serial double loop at [REF roman/align/submap_align.py:93-200].
 *
 *   feats      DEVICE, float64: pool of object-major feature matrices (F doubles per object)
 *   off1/off2  HOST, int64[B]: index (in objects) of problem b's first map-1 / map-2 object
 *              in `feats` (several problems may share a submap: the all-pairs grid)
 *   n1/n2      HOST, int32[B]: objects in map 1 / map 2 of problem b
 *   F          features per object = point_dim + ratio_feature_dim + cos_feature_dim
 *   assoc      DEVICE, int32 (sum A_b, 2) or NULL (= all-to-all for every problem)
 *   assoc_off  HOST, int64[B+1] row offsets into assoc (ignored when assoc == NULL), non-decreasing from 0;
 *              a problem whose list is EMPTY is scored all-to-all, as clipperpy does with an empty A
 *              (the reference reaches that case when its prefilter prunes everything,
 *              [REF roman/align/dist_reg_with_pruning.py:94-96]).  Rows must satisfy 0 <= i < n1, 0 <= j < n2:
 *              the host-pointer entry below checks them, this one cannot (device memory) and trusts the caller
 *   u0         DEVICE, float64 initial vectors, concatenated per problem in association order,
 *              or NULL (= all ones; DESIGN.md decision H1)
 *   kmax       capacity (rows) of each problem's slot in assoc_out
 *   assoc_out  DEVICE, int32[B][kmax][2]: selected associations (map-1 index, map-2 index),
 *              in clipperpy order (descending u)
 *   n_assoc_out DEVICE, int32[B]
 *   T_out      DEVICE, float64[B][16]: pose map2->map1, row-major (dim+1)^2 in the leading
 *              entries; NaN-filled when status has ROMAN_ST_INSUFFICIENT/EMPTY_MAP
 *              (the sentinel of [REF roman/align/submap_align.py:179-184])
 *   status_out DEVICE, int32[B]
 *   stats_out  DEVICE, roman_stats_t[B] or NULL
 *
 * The small per-problem metadata arrays are host memory (the library stages them itself);
 * all bulk data stays in HBM.
 * A PURE ENQUEUE: never synchronises a stream, never reads anything back.  The sparse workspace is sized before
 * the live counts are known — from what earlier batches with the same parameter block needed (the totals of a
 * finished batch are picked up from pinned memory without waiting), or from first-call heuristics; the device
 * checks every capacity itself, and a problem that does not fit is SKIPPED: status ROMAN_ST_WORKSPACE, no
 * associations, NaN pose.  Run those problems again (by then the context knows their sizes).  Results are
 * complete once the stream is synchronised (depth 1) / after roman_ctx_sync() (depth >= 2).
 */
ROMAN_API int roman_align_batch_dev(roman_ctx_t* ctx, const roman_params_t* params, int32_t B,
                          const double* feats, const int64_t* off1, const int32_t* n1,
                          const int64_t* off2, const int32_t* n2, int32_t F,
                          const int32_t* assoc, const int64_t* assoc_off,
                          const double* u0,
                          int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out,
                          double* T_out, int32_t* status_out, roman_stats_t* stats_out);

/* Problems that roman_align_batch_dev calls on this context have reported with ROMAN_ST_WORKSPACE so far (a running
   total; only batches whose totals have arrived count: wait != 0 synchronises every stream of the context first, so
   that all issued batches count).  A caller that never looks at status_out can still tell that something was skipped:
       int64_t before, after;
       roman_ctx_skipped(ctx, 1, &before);
       roman_align_batch_dev(ctx, ...);                          // enqueue (any number of calls)
       roman_ctx_skipped(ctx, 1, &after);                        // waits for them
       if (after != before) { ... status_out[b] & ROMAN_ST_WORKSPACE marks the problems: issue THOSE again — same call
                                  with off1/n1/off2/n2/assoc_off restricted to them; the context has recorded their need,
                                  so the second attempt sizes its pools for them ... }
   roman_align_batch (host pointers) runs exactly this loop itself (at most 5 attempts, then ROMAN_E_NOMEM). */
ROMAN_API int roman_ctx_skipped(roman_ctx_t* ctx, int wait, int64_t* n_skipped);

/* Same contract with HOST pointers everywhere; `n_objects` = number of objects in `feats`.
   Copies in, runs roman_align_batch_dev, copies out, synchronises.  This is what a cgo/ctypes
   binding that holds NumPy arrays calls. */
ROMAN_API int roman_align_batch(roman_ctx_t* ctx, const roman_params_t* params, int32_t B,
                      const double* feats, int64_t n_objects,
                      const int64_t* off1, const int32_t* n1,
                      const int64_t* off2, const int32_t* n2, int32_t F,
                      const int32_t* assoc, const int64_t* assoc_off,
                      const double* u0,
                      int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out,
                      double* T_out, int32_t* status_out, roman_stats_t* stats_out);

/* Inputs in HBM as for roman_align_batch_dev (feats, assoc, u0: DEVICE pointers; the metadata arrays host memory), results on the
   HOST as for roman_align_batch: what a caller needs whose submaps' feature pool stays resident (the all-pairs grid: every submap
   is uploaded once, [REF roman/align/submap_align.py:93-94]) but who consumes associations and poses on the host
   ([REF roman/align/submap_align.py:155-166]).  Synchronous; chunks, pipelines and retries like roman_align_batch.  The outputs of all
   problems land in one device block and come back with ONE copy through a pinned landing block owned by the context (a single
   pair's whole result — associations, count, pose, status, statistics — is 1.8 KB).  Device buffers must not be in use by work
   queued on streams the context does not know (the call starts behind the context's stream). */
ROMAN_API int roman_align_batch_resident(roman_ctx_t* ctx, const roman_params_t* params, int32_t B,
                               const double* feats, const int64_t* off1, const int32_t* n1,
                               const int64_t* off2, const int32_t* n2, int32_t F,
                               const int32_t* assoc, const int64_t* assoc_off,
                               const double* u0,
                               int32_t kmax, int32_t* assoc_out, int32_t* n_assoc_out,
                               double* T_out, int32_t* status_out, roman_stats_t* stats_out);

/* Does the context hold a sizing history for this parameter block (params, F) — i.e. has a batch with it reported what its sparse
   pools needed?  *yes = 1 / 0.  A device-pointer caller that queues several calls at once asks this first: without a history the
   first call should be waited for (roman_ctx_sync), so that the calls behind it size their pools from what it needed instead of
   repeating its guess (roman_align_batch does the same internally).  The context keeps ONE history: that of the latest block. */
ROMAN_API int roman_ctx_has_history(roman_ctx_t* ctx, const roman_params_t* params, int32_t F, int32_t* yes);

/* The cosine stage of a batch ([REF roman/align/roman_registration.py:52-59]: the semantic similarity only counts above cosine_min) has two
   implementations: the dense f64 matrix-core product of all pairs, and a bf16 screen of all pairs followed by the exact f64 contraction of
   the pairs the screen cannot rule out (k_cos_sel: same bits wherever the gate lets a pair through).  The library picks per batch — large
   batches of maps of at most 256 objects take the screen unless the latest screened batch of the parameter block left more than a quarter of
   its problems to the dense kernel (descriptors that are all alike).  Diagnostics: the numbers of batches that took either since the context
   was created, and the share of the latest screened batch that fell back (any pointer may be NULL).  ROMAN_COS_SEL=0 / 1 forces either. */
ROMAN_API int roman_ctx_cosine_screen_stats(roman_ctx_t* ctx, int64_t* screened_batches, int64_t* dense_batches, double* latest_fallback_share);

/* The deal of a batch over `world` ranks (one process per GPU), for a C / C++ caller that shards with its own collective
   (roman_ros, [REF README.md:11]; the Python side is roman_amd.align.distributed.align_sharded).  The pairs of the serial loop
   [REF roman/align/submap_align.py:93-200] are independent: every rank aligns its share with roman_align_batch[_dev] and ONE
   all_gather of the outputs (n_assoc_out, assoc_out, T_out, status_out: fixed-size rows per problem) collects them — no other
   exchange.  The deal is a pure host function of the problem sizes, identical on every rank without communication: problems
   longest-first (work estimate = the SQUARE of the association count, n1*n2 for all-to-all: pair tests and matrix entries grow
   with it) to the rank with the least work so far, ties to the rank holding fewer problems, then to the lowest rank.
   assoc_off: NULL (all-to-all) or int64[B+1] (an empty list means all-to-all).  idx_out: int32[B] capacity; on return its first
   *n_out entries are the ascending problem indices of `rank`. */
ROMAN_API int roman_deal_problems(int32_t B, const int32_t* n1, const int32_t* n2, const int64_t* assoc_off,
                                  int32_t world, int32_t rank, int32_t* idx_out, int32_t* n_out);

/* ------------------------------------------------------------------------------------------- */
/* stepwise surface for the clipperpy-compatible shim (single problem, host pointers)          */
/* ------------------------------------------------------------------------------------------- */

/* clipperpy.utils.create_all_to_all(n1, n2) [REF roman/align/object_registration.py:41]:
   out is (n1*n2, 2) int32, row i*n2+j = (i, j).  Pure host helper. */
ROMAN_API int roman_create_all_to_all(int32_t n1, int32_t n2, int32_t* out);

/* score_pairwise_consistency / score_pairwise_and_single_consistency
   [REF roman/align/object_registration.py:47], [REF roman/align/roman_registration.py:95]:
   builds M (and C, which has M's pattern) for ONE problem on the device and keeps it in the
   context.  assoc may be NULL, or n_assoc 0 (all-to-all). */
ROMAN_API int roman_score(roman_ctx_t* ctx, const roman_params_t* params,
                const double* D1, int32_t n1, const double* D2, int32_t n2, int32_t F,
                const int32_t* assoc, int32_t n_assoc);

/* clipper.set_matrix_data(M=, C=) [REF roman/align/object_registration.py:64]: dense
   row-major (n,n) float64 M and C; like upstream only the strict upper triangles are used
   and the diagonal is the implicit identity.  M and C are uploaded and converted to the
   solver's layout on the device (host pointers in, two small read-backs); the matrices stay
   in the context. */
ROMAN_API int roman_set_matrix_data(roman_ctx_t* ctx, const roman_params_t* params,
                          const double* M, const double* C, int32_t n);

/* clipper.solve() [REF roman/align/object_registration.py:27,65] on the matrices held by
   the context.  u0 may be NULL (all ones). */
ROMAN_API int roman_solve(roman_ctx_t* ctx, const double* u0);

/* Size queries + getters for the last roman_score/roman_set_matrix_data/roman_solve. */
ROMAN_API int roman_num_associations(const roman_ctx_t* ctx, int32_t* n_assoc);       /* rows of A     */
ROMAN_API int roman_num_selected(const roman_ctx_t* ctx, int32_t* n_sel);             /* len(nodes)    */
/* clipper.get_selected_associations() [REF object_registration.py:28]: (n_sel,2) int32 */
ROMAN_API int roman_get_selected_associations(const roman_ctx_t* ctx, int32_t* out);
/* clipper.get_solution().nodes / .u / .score [REF object_registration.py:67-71] */
ROMAN_API int roman_get_solution(const roman_ctx_t* ctx, int32_t* nodes, double* u, double* score,
                       roman_stats_t* stats);
/* clipper.get_affinity_matrix() / get_constraint_matrix() [REF object_registration.py:53-54]:
   dense row-major (A,A) float64, symmetric, diagonal included (1 for EUCLIDEAN, the single
   score for ROMAN; C's diagonal is 1).  Caller provides A*A doubles each; either may be NULL. */
ROMAN_API int roman_get_dense_matrices(const roman_ctx_t* ctx, double* M, double* C);
/* Sparse export of the same (strict upper triangle, CSR over association indices, ascending
   columns) for tests: call with NULL arrays to get nnz first. */
ROMAN_API int roman_get_upper_csr(const roman_ctx_t* ctx, int64_t* nnz, int64_t* rowptr /*A+1*/,
                        int32_t* cols, double* vals, double* diag /*A*/);

/* ------------------------------------------------------------------------------------------- */
/* pose from given correspondences                                                             */
/* ------------------------------------------------------------------------------------------- */

/* ObjectRegistration.T_align(map1, map2, correspondences) [REF object_registration.py:88-129]
   for B independent correspondence sets, host pointers.  pts1/pts2: (sum k_b, dim) float64
   row-major already gathered point pairs ([REF :110-111]); corr_off: int64[B+1].
   T_out: B x 16 doubles; status_out: ROMAN_ST_INSUFFICIENT when k_b < dim ([REF :107-108]). */
ROMAN_API int roman_pose_batch(roman_ctx_t* ctx, int32_t dim, int32_t B,
                     const double* pts1, const double* pts2, const int64_t* corr_off,
                     double* T_out, int32_t* status_out);

/* ------------------------------------------------------------------------------------------- */
/* instrumentation                                                                             */
/* ------------------------------------------------------------------------------------------- */

/* When enabled, the batch call brackets each of its kernels with hipEvents on the context's
   stream; roman_profile_get returns the accumulated per-stage milliseconds and launch counts
   since the last roman_profile_reset.  Stage ids: */
#define ROMAN_STAGE_SINGLE   0   /* norms, cos-sim MFMA GEMM, distance tables, single scores + live compaction */
#define ROMAN_STAGE_COUNT_PASS 1 /* affinity pair tests, mirror, degree sort, candidate lists, problem scans   */
#define ROMAN_STAGE_FILL     2   /* affinity fill: candidate lists -> matrix values                            */
#define ROMAN_STAGE_SOLVE    3   /* persistent projected-gradient solvers (+ select + pose)                    */
#define ROMAN_STAGE_COUNT    4
ROMAN_API int roman_profile_enable(roman_ctx_t* ctx, int on);
ROMAN_API int roman_profile_reset(roman_ctx_t* ctx);
ROMAN_API int roman_profile_get(roman_ctx_t* ctx, double ms[ROMAN_STAGE_COUNT],
                      int64_t launches[ROMAN_STAGE_COUNT]);

/* Diagnostics for tests: evaluate a device math primitive elementwise (host pointers).
   kind 0: sqrt(x)  1: exp(x)  2: cbrt(x)  3: x/y (y = in2)  4: pow(x,y)  5: the library's fixed-sequence
   exp  6: its fixed-sequence cbrt  7: fma(x,y,x).  Used to check that the device's +,-,*,/,sqrt,fma and the
   two fixed-sequence functions are bit-identical to the host's (pattern parity) and to
   measure the ulp distance of the device's libm. */
ROMAN_API int roman_debug_math(roman_ctx_t* ctx, int kind, const double* in1, const double* in2,
                     int64_t n, double* out);
/* Diagnostics for tests: normalised cosine matrix (n1 x n2, row-major) of the cosine-feature
   blocks of two object-major feature matrices, computed by the f64 MFMA kernel of stage SINGLE. */
ROMAN_API int roman_debug_cosine(roman_ctx_t* ctx, const roman_params_t* params,
                       const double* D1, int32_t n1, const double* D2, int32_t n2, int32_t F,
                       double* out);
/* Diagnostics for tests: live association list of the last roman_score (original association
   index and single score of every live association, ascending). n_live via roman_get_solution
   stats or by calling with NULL arrays. */
ROMAN_API int roman_debug_live(const roman_ctx_t* ctx, int32_t* n_live, int32_t* idx, double* score);

/* Library build info: "roman_hip <version> gfx950 ..." */
ROMAN_API const char* roman_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ROMAN_HIP_H */
