// TT input x TT DRMs: the whole streaming sketch (both chains, Omega, Psi) as one C call.
// See include/ttsk.h (ttsk_tt_sketch) for the contract and the reference lines it replaces.
#include <cstdlib>
#include <vector>
#include "common.h"
#include "skinny.h"
#include "chain_fused.h"
#include "chain_wide.h"
#include "chain_sum.h"
#include "tt_chain.h"
#include "stream_small.h"

namespace ttsk {

constexpr int SMAX_ROWS = 112;   // rows a fused step addresses in an X slab (tt_step.hip SMAX)
constexpr int NCLS = 12;      // 0-5 TT pipeline classes, 6 samplers, 7 sparse Psi / Omega, 8 small factorisations, 9-10 free, 11 everything else
struct ProfRec { hipEvent_t a, b; int cls; double flops; };
static char g_kname[NCLS][96];
static double g_kname_flops[NCLS];   // the name kept per class is that of its largest launch
static bool g_prof = false;
static int g_cls = NCLS - 1;
static std::vector<ProfRec> g_recs;
static int64_t g_launches[NCLS];
static double g_ms[NCLS], g_flops[NCLS];

// called by ttsk_gemm around its main kernel launch (not the split-K reduce / zero fill)
bool prof_on() { return g_prof; }
void prof_open(hipStream_t st, double flops, int family, int tiles, bool ak, bool bk)
{
    // name of the contraction-kernel instantiation as rocprofv3 prints it
    const int wm = family == 0 ? 2 : (family == 1 ? 1 : 4), wn = family == 0 ? 2 : (family == 1 ? 4 : 1);
    const int tm = family == 0 ? 2 : (family == 1 ? tiles : 1), tn = family == 0 ? 2 : (family == 1 ? 1 : tiles);
    if (flops < g_kname_flops[g_cls]) family = -1;
    else g_kname_flops[g_cls] = flops;
    if (family < 0) {
    } else if (family == 3)        // streamed x small: tiles = 10 * column tiles of the small operand + mode, ak = strips > 0
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "skinny_s_kernel<%d, %d, %d, %d>",
                 tiles / 100 % 10, tiles % 10, bk ? 4 : 5, tiles / 1000);
    else if (family == 5)
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "small_gemm_kernel");
    else if (family == 7)   // streamed x small, contiguous form: tiles = 10 * full tiles + strips
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "stream_small_kernel<%d, %d, 5, %d>", tiles / 10, tiles % 10, ak ? 25 : 5);
    else if (family == 6)   // fused chain step: tiles = 100 * full tiles + 10 * strips + (T written)
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "chain_step_kernel<%d, %d, %d, %d, 5, %s, 1, %d, %d>", tiles / 100, tiles / 10 % 10,
                 tiles / 100, tiles / 10 % 10, tiles % 10 ? "true" : "false", tiles / 100 <= 4 ? 2 : 1, ak ? 25 : 5);
    else if (family == 4)   // long-K: tiles = 10 * row tiles + column tiles
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "skinny_r_kernel<%d, %d, 4>", tiles / 10, tiles % 10);
    else
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "gemm_f64_kernel<%d, %d, %d, %d, %s, %s>", wm, wn, tm, tn,
                 ak ? "true" : "false", bk ? "true" : "false");
    ProfRec r{};
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, st);
    r.cls = g_cls;
    r.flops = flops;
    g_recs.push_back(r);
}
void prof_close(hipStream_t st) { (void)hipEventRecord(g_recs.back().b, st); }

// brackets for kernels outside ttsk_gemm (samplers, sparse segmented sums, factorisations): class and name given
// by the caller, `work` in the class's own unit (samples, bytes, flops)
void prof_open_named(hipStream_t st, int cls, double work, const char *name)
{
    if (cls == -2) cls = g_cls;                   // the class the TT driver has set for this product
    if (cls < 0 || cls >= NCLS) cls = NCLS - 1;
    if (work >= g_kname_flops[cls]) {
        g_kname_flops[cls] = work;
        snprintf(g_kname[cls], sizeof(g_kname[0]), "%s", name);
    }
    ProfRec r{};
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, st);
    r.cls = cls;
    r.flops = work;
    g_recs.push_back(r);
}

static void prof_flush()
{
    for (auto &r : g_recs) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_launches[r.cls]++;
            g_ms[r.cls] += ms;
            g_flops[r.cls] += r.flops;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
}

// C[M,N] (+)= A * B with generic strides; tags the launch with its profiling class.
static int gemm(int cls, int64_t M, int64_t N, int64_t Ko, int64_t Ki, const double *A, int64_t a_m,
                int64_t a_ko, int64_t a_ki, const double *B, int64_t b_ko, int64_t b_ki, int64_t b_n,
                double *C, int64_t c_m, int64_t c_n, int accumulate, int stream)
{
    ttsk_gemm_desc d{};
    d.batch = 1; d.M = M; d.N = N; d.Ko = Ko; d.Ki = Ki;
    d.a_m = a_m; d.a_ko = a_ko; d.a_ki = a_ki;
    d.b_ko = b_ko; d.b_ki = b_ki; d.b_n = b_n;
    d.c_m = c_m; d.c_n = c_n;
    d.alpha = 1.0; d.accumulate = accumulate; d.split_k = 0;
    g_cls = cls;
    int rc = ttsk_gemm(&d, A, B, C, nullptr, stream);
    g_cls = NCLS - 1;
    return rc;
}

// One product for every tensor of a batch: the chain kernels take all nb problems in one launch
// (skinny_try_batch); shapes they do not cover fall back to one ttsk_gemm per tensor.
struct BatchPtrs {
    const double *A[SK_MAXB], *B[SK_MAXB];
    double *C[SK_MAXB];
};

static int gemm_batch(int cls, int nb, ttsk_gemm_desc d, const BatchPtrs &p, int stream, hipStream_t st,
                      bool chain_only = false)
{
    g_cls = cls;
    ttsk_gemm_desc n = d;
    if (n.Ki == 1) { n.Ki = n.Ko; n.Ko = 1; n.a_ki = n.a_ko; n.b_ki = n.b_ko; }
    else if (n.Ko > 1 && n.a_ko == n.Ki * n.a_ki && n.b_ko == n.Ki * n.b_ki) { n.Ki *= n.Ko; n.Ko = 1; }
    int rc = skinny_try_batch(n, nb, p.A, p.B, p.C, stream, st);
    if (rc == 0 && !chain_only) rc = small_try_batch(n, nb, p.A, p.B, p.C, stream, st);
    if (rc == 0 && !chain_only) {
        rc = 1;
        for (int b = 0; b < nb && rc == 1; ++b) {
            const int e = ttsk_gemm(&d, p.A[b], p.B[b], p.C[b], nullptr, stream);
            if (e != TTSK_OK) rc = e;
        }
    }
    g_cls = NCLS - 1;
    return rc;   // 1 = done, 0 = not covered (chain_only), < 0 = error
}

static ttsk_gemm_desc desc2(int64_t M, int64_t N, int64_t Ko, int64_t Ki, int64_t a_m, int64_t a_ko, int64_t a_ki,
                            int64_t b_ko, int64_t b_ki, int64_t b_n, int64_t c_m, int64_t c_n, int accumulate)
{
    ttsk_gemm_desc d{};
    d.batch = 1; d.M = M; d.N = N; d.Ko = Ko; d.Ki = Ki;
    d.a_m = a_m; d.a_ko = a_ko; d.a_ki = a_ki;
    d.b_ko = b_ko; d.b_ki = b_ki; d.b_n = b_n;
    d.c_m = c_m; d.c_n = c_n;
    d.alpha = 1.0; d.accumulate = accumulate; d.split_k = 0;
    return d;
}

}  // namespace ttsk

using namespace ttsk;

extern "C" {

int ttsk_prof_enable(int on)
{
    if (ensure_init() != TTSK_OK) return TTSK_ERR_HIP;
    if (!on) prof_flush();
    else {
        prof_flush();
        for (int i = 0; i < NCLS; ++i) { g_launches[i] = 0; g_ms[i] = 0; g_flops[i] = 0; g_kname_flops[i] = 0; }
    }
    g_prof = on != 0;
    return TTSK_OK;
}

int ttsk_prof_kernel_name(int cls, char *buf, size_t len)
{
    TTSK_ARG(cls >= 0 && cls < NCLS && buf && len > 0, "ttsk_prof_kernel_name: bad argument");
    snprintf(buf, len, "%s", g_kname[cls]);
    return TTSK_OK;
}

int ttsk_prof_read(int cls, int64_t *launches, double *total_ms, double *flops)
{
    TTSK_ARG(cls >= 0 && cls < NCLS, "ttsk_prof_read: class %d", cls);
    prof_flush();
    if (launches) *launches = g_launches[cls];
    if (total_ms) *total_ms = g_ms[cls];
    if (flops) *flops = g_flops[cls];
    return TTSK_OK;
}

int64_t ttsk_tt_sketch_size(int d, const int64_t *n, const int64_t *l_lo, const int64_t *l_hi,
                            const int64_t *r_lo, const int64_t *r_hi)
{
    int64_t tot = 0;
    for (int mu = 0; mu < d; ++mu) {
        int64_t l = mu == 0 ? 1 : l_hi[mu - 1] - l_lo[mu - 1];
        int64_t r = mu == d - 1 ? 1 : r_hi[d - 2 - mu] - r_lo[d - 2 - mu];
        tot += l * n[mu] * r;
    }
    for (int mu = 0; mu < d - 1; ++mu)
        tot += (l_hi[mu] - l_lo[mu]) * (r_hi[d - 2 - mu] - r_lo[d - 2 - mu]);
    return tot;
}

int ttsk_tt_sketch(int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                   const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                   const double *const *X, const double *const *DL, const double *const *DR, double *out,
                   int accumulate, int stream)
{
    return ttsk_tt_sketch_batch(1, d, n, s, lt, l_lo, l_hi, rt, r_lo, r_hi, X, DL, DR, out, 0, accumulate, stream);
}

static int tt_sketch_core(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                          const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                          const double *const *X, const double *const *DL, const double *const *DR, double *out,
                          int64_t out_stride, int accumulate, int stream, bool sum, ttsk::TTChains *co = nullptr);

int ttsk_tt_sketch_batch(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                         const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                         const double *const *X, const double *const *DL, const double *const *DR, double *out,
                         int64_t out_stride, int accumulate, int stream)
{
    return tt_sketch_core(nb, d, n, s, lt, l_lo, l_hi, rt, r_lo, r_hi, X, DL, DR, out, out_stride, accumulate, stream, false);
}

int ttsk_tt_sketch_sum(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                       const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                       const double *const *X, const double *const *DL, const double *const *DR, double *out,
                       int accumulate, int stream)
{
    return tt_sketch_core(nb, d, n, s, lt, l_lo, l_hi, rt, r_lo, r_hi, X, DL, DR, out, 0, accumulate, stream, true);
}

// sum = false: sketch b at out + b * out_stride.  sum = true: ONE sketch, of the sum of the nb tensors -- the chains
// run per tensor as before (a sum of TTs is a TT with block-diagonal cores), Psi and Omega contract over
// (tensor, rank) at once: the per-tensor workspaces are equally spaced, so that pair is a two-level contracted
// index of one product, and no per-tensor sketch is ever written or summed.
static int tt_sketch_core(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                          const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                          const double *const *X, const double *const *DL, const double *const *DR, double *out,
                          int64_t out_stride, int accumulate, int stream, bool sum, ttsk::TTChains *co)
{
    // co (tt_chain.h): the chains and Omega only -- no Psi, T not kept; the chain matrices stay in the workspace and
    // their addresses are handed back (the orthogonalising sketches of tt_orth.hip go on from there)
    TTSK_STREAM(st, stream);
    TTSK_ARG(nb >= 1, "ttsk_tt_sketch_batch: need nb >= 1, got %d", nb);
    TTSK_ARG(d >= 2, "ttsk_tt_sketch: need d >= 2, got %d", d);
    TTSK_ARG(n && s && lt && l_lo && l_hi && rt && r_lo && r_hi && X && DR && (co ? (DL || !co->want_left) : (DL && out)),
             "ttsk_tt_sketch: NULL argument");
    TTSK_ARG(!co || (nb <= SK_MAXB && !sum && !accumulate), "ttsk_tt_sketch: chains-only mode takes one slice of tensors");
    const bool no_left = co && !co->want_left;
    TTSK_ARG(s[0] == 1 && s[d] == 1 && lt[0] == 1 && rt[0] == 1, "ttsk_tt_sketch: boundary ranks must be 1");
    for (int mu = 0; mu < d - 1; ++mu) {
        TTSK_ARG(0 <= l_lo[mu] && l_lo[mu] <= l_hi[mu] && l_hi[mu] <= lt[mu + 1],
                 "ttsk_tt_sketch: left rank slice %d out of range", mu);
        TTSK_ARG(0 <= r_lo[mu] && r_lo[mu] <= r_hi[mu] && r_hi[mu] <= rt[mu + 1],
                 "ttsk_tt_sketch: right rank slice %d out of range", mu);
    }
    const int64_t one = ttsk_tt_sketch_size(d, n, l_lo, l_hi, r_lo, r_hi);
    TTSK_ARG(co || sum || nb == 1 || out_stride >= one, "ttsk_tt_sketch_batch: out_stride %lld < sketch size %lld",
             (long long)out_stride, (long long)one);
    if (nb > SK_MAXB) {   // larger batches in slices of SK_MAXB tensors
        for (int b0 = 0; b0 < nb; b0 += SK_MAXB) {
            const int cnt = nb - b0 < SK_MAXB ? nb - b0 : SK_MAXB;
            int rc = tt_sketch_core(cnt, d, n, s, lt, l_lo, l_hi, rt, r_lo, r_hi, X + (size_t)b0 * d, DL, DR,
                                    sum ? out : out + (size_t)b0 * out_stride, out_stride, (sum && b0) ? 1 : accumulate,
                                    stream, sum);
            if (rc) return rc;
        }
        return TTSK_OK;
    }
    if (nb == 1) sum = false;
    // The right chain runs on the caller's stream, the left chain on a helper stream (the two are
    // independent until Psi / Omega need both); the Psi products are then dealt over both.  The
    // helper is forked from / joined into `stream`, so callers (and hipGraph capture) see one stream.
    // TTSK_SINGLE_STREAM=1 keeps everything on `stream`: per-kernel event times are then free of
    // cross-stream sharing and match rocprofv3's kernel durations (bench.py roofline leg).
    const char *single = getenv("TTSK_SINGLE_STREAM");
    const int aux = (single && single[0] == '1') ? stream : (stream + 1) % TTSK_NUM_STREAMS;
    TTSK_STREAM(st_aux, aux);
    // Workspace (slot DRIVER of `stream`), one block per QUANTITY with the nb tensors behind one another:
    // Lc[mu] (s[mu+1] x lt[mu+1]), Rc[j] (s[d-1-j] x rt[j+1]), T[mu] per left mode (kept for the Psi phase), one
    // T buffer for the right chain.  Tensor b of a quantity of size sz sits at block + b * stride(sz), stride =
    // sz rounded up to even (16-byte operand loads).  Where sz is even the nb chain matrices of a mode are ONE
    // (nb s) x rank matrix: the two-launch chain step and the Psi / Omega of a sum then run as one product over
    // all tensors ("merged" below) and read the DRM core once instead of once per tensor.
    auto even = [](size_t v) { return v + (v & 1); };
    auto blk = [](size_t v) { return (v + 31) & ~(size_t)31; };
    size_t tot = 0;
    std::vector<size_t> offL(d - 1), offR(d - 1), offT(d), szL(d - 1), szR(d - 1), szT(d);
    for (int mu = 0; mu < d - 1; ++mu) { szL[mu] = even((size_t)s[mu + 1] * lt[mu + 1]); offL[mu] = tot; tot += blk(nb * szL[mu]); }
    for (int j = 0; j < d - 1; ++j) { szR[j] = even((size_t)s[d - 1 - j] * rt[j + 1]); offR[j] = tot; tot += blk(nb * szR[j]); }
    for (int mu = 1; mu < d; ++mu) { szT[mu] = even((size_t)lt[mu] * n[mu] * s[mu + 1]); offT[mu] = tot; tot += blk(nb * szT[mu]); }
    size_t tr_max = 0;
    for (int mu = 1; mu < d - 1; ++mu) {
        size_t tr = (size_t)rt[d - 1 - mu] * n[mu] * s[mu];
        tr_max = tr > tr_max ? tr : tr_max;
    }
    const size_t szTR = even(tr_max), offTR = tot;
    tot += blk(nb * szTR);
    const size_t szP0 = even((size_t)n[0] * (r_hi[d - 2] - r_lo[d - 2])), offP0 = tot;   // sum mode: Psi_0 per tensor
    if (sum) tot += blk(nb * szP0);
    // sum mode, larger TT ranks: Psi_mu per tensor (the streamed kernel), then one sum -- faster than the generic
    // tiles on a contracted index of nb * s (measured: s = 60, 100); one block per stream of the Psi phase
    static const int sum_psi_split = [] { const char *e = getenv("TTSK_SUM_PSI_SPLIT"); return e ? atoi(e) : 48; }();
    // (small TT ranks: the same blocks take the partial Psi of the K chunks of the one product over (tensor, rank), below)
    static const int sum_psi_chunks = [] { const char *e = getenv("TTSK_SUM_PSI_CHUNKS"); return e ? atoi(e) : 1; }();
    size_t szPs = 0;
    if (sum)
        for (int mu = 1; mu < d - 1; ++mu)
            if (s[mu + 1] > sum_psi_split || sum_psi_chunks) {
                const size_t v = even((size_t)(l_hi[mu - 1] - l_lo[mu - 1]) * n[mu] * (r_hi[d - 2 - mu] - r_lo[d - 2 - mu]));
                szPs = v > szPs ? v : szPs;
            }
    const size_t offPs = tot;
    tot += 2 * blk(nb * szPs);
    double *ws0 = (double *)scratch(stream, SCRATCH_DRIVER, tot * 8);
    if (!ws0) return TTSK_ERR_HIP;
    auto Lp = [&](int b, int mu) { return ws0 + offL[mu] + (size_t)b * szL[mu]; };
    auto Rp = [&](int b, int j) { return ws0 + offR[j] + (size_t)b * szR[j]; };
    auto Tp0 = [&](int b, int mu) { return ws0 + offT[mu] + (size_t)b * szT[mu]; };     // per-tensor T[q][k][p']
    auto TRp = [&](int b) { return ws0 + offTR + (size_t)b * szTR; };
    // "merged" needs the tensors of a chain matrix exactly behind one another
    auto packedL = [&](int mu) { return szL[mu] == (size_t)s[mu + 1] * lt[mu + 1]; };
    auto packedR = [&](int j) { return szR[j] == (size_t)s[d - 1 - j] * rt[j + 1]; };
    static const int merge_on = [] { const char *e = getenv("TTSK_TT_MERGE"); return e ? atoi(e) : 1; }();
    std::vector<int> t_inter(d, 0);     // T[mu] stored interleaved: T[(q,k)][(b,p')], row length nb * s[mu+1]
    auto Xc = [&](int b, int mu) { return X[(size_t)b * d + mu]; };
    auto outb = [&](int b) { return out + (size_t)b * out_stride; };
    int rc;
#define CK(x) do { rc = (x); if (rc < 0) return rc; } while (0)
    // ---- the pieces: one step of either chain, and the Psi / Omega of one mode on a given stream
    // right chain: walks modes d-1, ..., 1 on the transposed tensor (views only).  Xt_j[p,k,p''] = X_mu[p'',k,p], mu = d-1-j.
    auto right_step = [&](int j) -> int {
        const int mu = d - 1 - j;
        const int64_t sp = s[mu + 1], sn = s[mu], nn = n[mu], rho = rt[j], rhop = rt[j + 1];
        BatchPtrs p{};
        if (j == 0) {
            // Rc_0[p'',q'] = sum_k X[p'',k,0] E[0,k,q']
            for (int b = 0; b < nb; ++b) { p.A[b] = Xc(b, mu); p.B[b] = DR[j]; p.C[b] = Rp(b, j); }
            CK(gemm_batch(5, nb, desc2(sn, rhop, 1, nn, nn * sp, 0, sp, 0, rhop, 1, rhop, 1, 0), p, stream, st));
            return TTSK_OK;
        }
        // first choice: both products in one launch, T never written (chain_fused.h)
        {
            const double *Wp[SK_MAXB], *Xp[SK_MAXB];
            double *Op[SK_MAXB];
            for (int b = 0; b < nb; ++b) { Wp[b] = Rp(b, j - 1); Xp[b] = Xc(b, mu); Op[b] = Rp(b, j); }
            ChainStepArgs cs{nb, (int)nn, (int)sp, (int)rho, (int)rhop, (int)sn, Wp, rho, Xp, nn * sp, sp, 1, sn * nn * sp,
                             DR[j], nullptr, Op};
            g_cls = 1;
            int fz = (sp <= 128 && sn <= 128 && rho <= 128 && rhop <= 128) ? chain_fused_try(cs, stream, st) : 0;
            // many low-rank tensors (the terms of a sum): rows of several terms stacked into full tiles (chain_sum.h)
            if (fz == 0) fz = chain_sum_try(ChainSumArgs{cs, nullptr, 0, 0, 0}, stream, st);
            if (fz == 0) fz = chain_wide_try(cs, stream, st);
            g_cls = NCLS - 1;
            if (fz < 0) return fz;
            if (fz == 1) return TTSK_OK;
        }
        // otherwise two launches.  T[q, k, b, p''] = sum_p Rc_b[p,q] X_b[p'',k,p]: a product batched over k whose
        // batch index joins the streamed index, written interleaved over the tensors (row (q,k), columns (b,p'')) ...
        const bool merged = merge_on && nb > 1 && packedR(j) && (int64_t)nb * sn * rho * nn <= (int64_t)nb * szTR;
        const int64_t ldt = merged ? (int64_t)nb * sn : sn;
        for (int b = 0; b < nb; ++b) { p.A[b] = Rp(b, j - 1); p.B[b] = Xc(b, mu); p.C[b] = merged ? TRp(0) + (size_t)b * sn : TRp(b); }
        ttsk_gemm_desc g1{};
        g1.batch = nn; g1.M = rho; g1.N = sn; g1.Ko = 1; g1.Ki = sp;
        g1.a_b = 0; g1.a_m = 1; g1.a_ki = rho;
        g1.b_b = sp; g1.b_ki = 1; g1.b_n = nn * sp;
        g1.c_b = ldt; g1.c_m = nn * ldt; g1.c_n = 1;
        g1.alpha = 1.0;
        const int fast = gemm_batch(0, nb, g1, p, stream, st, !merged);
        if (fast < 0) return fast;
        BatchPtrs q{};
        if (merged) {
            // ... so that Rn_all[(b,p''), q'] = sum_{q,k} T[(q,k), (b,p'')] E[q,k,q'] is ONE long product: E is read
            // once, not once per tensor
            q.A[0] = TRp(0); q.B[0] = DR[j]; q.C[0] = Rp(0, j);
            CK(gemm_batch(1, 1, desc2((int64_t)nb * sn, rhop, rho, nn, 1, nn * ldt, ldt, nn * rhop, rhop, 1, rhop, 1, 0), q,
                          stream, st));
            return TTSK_OK;
        }
        for (int b = 0; b < nb; ++b) { q.A[b] = TRp(b); q.B[b] = DR[j]; q.C[b] = Rp(b, j); }
        if (fast == 1) {
            // Rn[p'', q'] = sum_{q,k} T[q,k,p''] E[q,k,q']
            CK(gemm_batch(1, nb, desc2(sn, rhop, rho, nn, 1, nn * sn, sn, nn * rhop, rhop, 1, rhop, 1, 0), q, stream, st));
        } else {
            // T[q, p'', k] = sum_p Rc[p,q] X[p'',k,p]    (M=q, N=(p'',k), K=p)
            CK(gemm_batch(0, nb, desc2(rho, sn * nn, 1, sp, 1, 0, rho, 0, 1, sp, sn * nn, 1, 0), p, stream, st));
            // Rn[p'', q'] = sum_{q,k} T[q,p'',k] E[q,k,q']
            CK(gemm_batch(1, nb, desc2(sn, rhop, rho, nn, nn, sn * nn, 1, nn * rhop, rhop, 1, rhop, 1, 0), q, stream, st));
        }
        return TTSK_OK;
    };
    // left chain: L_mu and the shared products T_mu = L_{mu-1}^T X_mu
    auto left_step = [&](int mu) -> int {
        const int64_t sn = s[mu], sp = s[mu + 1], nn = n[mu];
        if (no_left || (co && mu == d - 1)) return TTSK_OK;         // (the last step only makes Psi_{d-1})
        BatchPtrs p{};
        if (mu == 0) {
            // L_0[p',q'] = sum_k X_0[0,k,p'] D_0[0,k,q']
            for (int b = 0; b < nb; ++b) { p.A[b] = Xc(b, 0); p.B[b] = DL[0]; p.C[b] = Lp(b, 0); }
            CK(gemm_batch(5, nb, desc2(sp, lt[1], 1, nn, 1, 0, sp, 0, lt[1], 1, lt[1], 1, 0), p, aux, st_aux));
            return TTSK_OK;
        }
        const int64_t lfull = lt[mu];
        if (mu < d - 1) {
            // first choice: T and L_mu from one launch (chain_fused.h); T is still stored, Psi_mu needs it
            const double *Wp[SK_MAXB], *Xp[SK_MAXB];
            double *Op[SK_MAXB], *Tp[SK_MAXB];
            for (int b = 0; b < nb; ++b) { Wp[b] = Lp(b, mu - 1); Xp[b] = Xc(b, mu); Op[b] = Lp(b, mu); Tp[b] = Tp0(b, mu); }
            ChainStepArgs cs{nb, (int)nn, (int)sn, (int)lfull, (int)lt[mu + 1], (int)sp, Wp, lfull, Xp, 1, sp, nn * sp,
                             sn * nn * sp, DL[mu], co ? nullptr : Tp, Op};
            g_cls = 3;
            int fz = (sp <= 128 && sn <= 128 && lfull <= 128 && lt[mu + 1] <= 128) ? chain_fused_try(cs, aux, st_aux) : 0;
            if (fz == 0) {
                // stacked-terms kernel: T goes out interleaved over the terms for the Psi of a sum (one product over (term,
                // rank)), per term otherwise
                const bool inter = sum && merge_on && nb > 1 && packedL(mu);
                ChainSumArgs ca{cs, nullptr, 0, 0, 0};
                if (!co) {
                    ca.Tint = inter ? ws0 + offT[mu] : Tp0(0, mu);
                    ca.t_b = inter ? sp : (int64_t)szT[mu];
                    ca.t_ld = inter ? (int64_t)nb * sp : sp;
                    ca.t_extent = (int64_t)nb * (int64_t)szT[mu];
                }
                fz = chain_sum_try(ca, aux, st_aux);
                if (fz == 1 && inter) t_inter[mu] = 1;
            }
            if (fz == 0) fz = chain_wide_try(cs, aux, st_aux);
            g_cls = NCLS - 1;
            if (fz < 0) return fz;
            if (fz == 1) return TTSK_OK;
        }
        const bool merged = merge_on && nb > 1 && mu < d - 1 && packedL(mu);
        if (merged) {
            // T[q, k, b, p'] = sum_p Lc_b[p,q] X_b[p,k,p'] interleaved over the tensors (batched over k) ...
            t_inter[mu] = 1;
            const int64_t ldt = (int64_t)nb * sp;
            double *T0 = ws0 + offT[mu];
            for (int b = 0; b < nb; ++b) { p.A[b] = Lp(b, mu - 1); p.B[b] = Xc(b, mu); p.C[b] = T0 + (size_t)b * sp; }
            ttsk_gemm_desc g1{};
            g1.batch = nn; g1.M = lfull; g1.N = sp; g1.Ko = 1; g1.Ki = sn;
            g1.a_b = 0; g1.a_m = 1; g1.a_ki = lfull;
            g1.b_b = sp; g1.b_ki = nn * sp; g1.b_n = 1;
            g1.c_b = ldt; g1.c_m = nn * ldt; g1.c_n = 1;
            g1.alpha = 1.0;
            CK(gemm_batch(2, nb, g1, p, aux, st_aux));
            // ... and L_all[(b,p'), q'] = sum_{q,k} T[(q,k), (b,p')] D[q,k,q'] as one long product
            BatchPtrs q{};
            q.A[0] = T0; q.B[0] = DL[mu]; q.C[0] = Lp(0, mu);
            CK(gemm_batch(3, 1, desc2((int64_t)nb * sp, lt[mu + 1], lfull, nn, 1, nn * ldt, ldt, nn * lt[mu + 1], lt[mu + 1], 1,
                                      lt[mu + 1], 1, 0), q, aux, st_aux));
            return TTSK_OK;
        }
        // T[q,k,p'] = sum_p Lc[p,q] X[p,k,p']      (M=q (all lfull columns), N=(k,p'), K=p)
        for (int b = 0; b < nb; ++b) { p.A[b] = Lp(b, mu - 1); p.B[b] = Xc(b, mu); p.C[b] = Tp0(b, mu); }
        CK(gemm_batch(mu == d - 1 ? 5 : 2, nb, desc2(lfull, nn * sp, 1, sn, 1, 0, lfull, 0, nn * sp, 1, nn * sp, 1, 0), p,
                      aux, st_aux));
        if (mu < d - 1) {
            // L_mu[p',q'] = sum_{q,k} T[q,k,p'] D[q,k,q']
            BatchPtrs q{};
            for (int b = 0; b < nb; ++b) { q.A[b] = Tp0(b, mu); q.B[b] = DL[mu]; q.C[b] = Lp(b, mu); }
            CK(gemm_batch(3, nb, desc2(sp, lt[mu + 1], 1, lfull * nn, 1, 0, sp, 0, lt[mu + 1], 1, lt[mu + 1], 1, 0), q,
                          aux, st_aux));
        }
        return TTSK_OK;
    };
    std::vector<size_t> psi_at(d), om_at(d - 1);
    {
        size_t p = 0;
        for (int mu = 0; mu < d; ++mu) {
            int64_t l = mu == 0 ? 1 : l_hi[mu - 1] - l_lo[mu - 1];
            int64_t r = mu == d - 1 ? 1 : r_hi[d - 2 - mu] - r_lo[d - 2 - mu];
            psi_at[mu] = p;
            p += l * n[mu] * r;
        }
        for (int mu = 0; mu < d - 1; ++mu) {
            om_at[mu] = p;
            p += (l_hi[mu] - l_lo[mu]) * (r_hi[d - 2 - mu] - r_lo[d - 2 - mu]);
        }
    }
    auto psi_omega = [&](int mu, int q, bool do_psi = true, bool do_omega = true) -> int {
        const int64_t sp = s[mu + 1], nn = n[mu];
        hipStream_t stq = stream_of(q);
        // right contraction of modes mu+1.. : Rc[j] with j = d-2-mu, columns [r_lo, r_hi)
        const int jr = d - 2 - mu;
        const int64_t ldr = mu < d - 1 ? rt[jr + 1] : 0, r = mu < d - 1 ? r_hi[jr] - r_lo[jr] : 1;
        auto Rm = [&](int b) { return Rp(b, jr) + r_lo[jr]; };
        BatchPtrs p{};
        if (!do_psi) {
        } else if (mu == 0 && sum) {
            // the cores X_b,0 are anywhere in memory: per-tensor products into the workspace, then one sum
            for (int b = 0; b < nb; ++b) { p.A[b] = Xc(b, 0); p.B[b] = Rm(b); p.C[b] = ws0 + offP0 + (size_t)b * szP0; }
            CK(gemm_batch(5, nb, desc2(nn, r, 1, sp, sp, 0, 1, 0, ldr, 1, r, 1, 0), p, q, stq));
            CK(ttsk_sum_slices(out + psi_at[0], ws0 + offP0, nb, szP0, (size_t)(nn * r), accumulate, q));
        } else if (mu == 0) {
            // Psi_0[0,k,c] = sum_{p'} X_0[0,k,p'] R_0[p',c]
            for (int b = 0; b < nb; ++b) { p.A[b] = Xc(b, 0); p.B[b] = Rm(b); p.C[b] = outb(b) + psi_at[0]; }
            CK(gemm_batch(5, nb, desc2(nn, r, 1, sp, sp, 0, 1, 0, ldr, 1, r, 1, accumulate), p, q, stq));
        } else {
            const int64_t l = l_hi[mu - 1] - l_lo[mu - 1];
            // T_b[(q,k), p'] for the rows of the rank slice: tensor b, row stride
            const int64_t ldt = t_inter[mu] ? (int64_t)nb * sp : sp;
            auto Tm = [&](int b) {
                return t_inter[mu] ? ws0 + offT[mu] + (size_t)(l_lo[mu - 1] * nn) * ldt + (size_t)b * sp
                                   : Tp0(b, mu) + (size_t)(l_lo[mu - 1] * nn) * sp;
            };
            bool psi_summed = false;
            if (mu < d - 1 && sum && sp > sum_psi_split && !t_inter[mu]) {
                // Psi_mu of the sum in ONE launch: every wave keeps its tile of the output over all terms (stream_small.h)
                StreamSmallSumArgs sa{nb, (int)(l * nn), (int)sp, (int)r, Tm(0), ldt, (int64_t)szT[mu], Rm(0), ldr, (int64_t)szR[jr],
                                      out + psi_at[mu], r, accumulate};
                g_cls = 4;
                const int fz = (l * nn < (1ll << 30)) ? stream_small_sum_try(sa, q, stq) : 0;
                g_cls = NCLS - 1;
                if (fz < 0) return fz;
                psi_summed = fz == 1;
            }
            if (psi_summed) {
            } else if (mu < d - 1 && sum && sp > sum_psi_split) {
                double *blk0 = ws0 + offPs + (size_t)(mu & 1) * blk(nb * szPs);
                for (int b = 0; b < nb; ++b) { p.A[b] = Tm(b); p.B[b] = Rm(b); p.C[b] = blk0 + (size_t)b * szPs; }
                StreamSmallArgs ss{nb, (int)(l * nn), (int)sp, (int)r, p.A, ldt, p.B, ldr, p.C, r, 0};
                g_cls = 4;
                const int fz = (l * nn < (1ll << 30)) ? stream_small_try(ss, q, stq) : 0;
                g_cls = NCLS - 1;
                if (fz < 0) return fz;
                if (fz == 0) CK(gemm_batch(4, nb, desc2(l * nn, r, 1, sp, ldt, 0, 1, 0, ldr, 1, r, 1, 0), p, q, stq));
                CK(ttsk_sum_slices(out + psi_at[mu], blk0, nb, szPs, (size_t)(l * nn * r), accumulate, q));
            } else if (mu < d - 1 && sum) {
                // Psi[(q,k), c] = sum_{b, p'} T_b[(q,k), p'] R_b[p', c]: (b, p') is one contracted index when both
                // operands hold the tensors behind one another, a two-level one otherwise
                bool chunked = false;
                if (t_inter[mu] && packedR(jr) && sum_psi_chunks && l * nn >= 1024 && l * nn < (1ll << 30)) {
                    // K = nb * sp (640 at C5) in chunks whose R image fits the LDS of the streamed kernel (stream_small.h:
                    // fragments of T straight from memory, R in LDS, no barrier after staging): the chunks are the
                    // "problems" of one launch, their partial Psi meet in one sum.  (One product on the generic tiles: 52 us
                    // per mode at C5.)
                    const int64_t K = (int64_t)nb * sp;
                    int nch = 0;
                    for (int t = 1; t <= nb && t <= SK_MAXB && !nch; ++t)
                        if (K % t == 0 && (K / t) % 4 == 0 && 4 * (K / t / 4 + 6) * ((r + 3) / 4 * 4) * 8 <= 150 * 1024) nch = t;
                    if (nch >= 1) {
                        const int64_t Kc = K / nch;
                        double *blk0 = ws0 + offPs + (size_t)(mu & 1) * blk(nb * szPs);
                        for (int cidx = 0; cidx < nch; ++cidx) {
                            p.A[cidx] = Tm(0) + (size_t)cidx * Kc; p.B[cidx] = Rm(0) + (size_t)cidx * Kc * ldr;
                            p.C[cidx] = nch == 1 ? out + psi_at[mu] : blk0 + (size_t)cidx * szPs;
                        }
                        StreamSmallArgs ss{nch, (int)(l * nn), (int)Kc, (int)r, p.A, ldt, p.B, ldr, p.C, r, nch == 1 ? accumulate : 0};
                        g_cls = 4;
                        const int fz = stream_small_try(ss, q, stq);
                        g_cls = NCLS - 1;
                        if (fz < 0) return fz;
                        if (fz == 1) {
                            chunked = true;
                            if (nch > 1) CK(ttsk_sum_slices(out + psi_at[mu], blk0, nch, szPs, (size_t)(l * nn * r), accumulate, q));
                        }
                    }
                }
                if (chunked) rc = 0;
                else if (t_inter[mu] && packedR(jr))
                    rc = gemm(4, l * nn, r, 1, (int64_t)nb * sp, Tm(0), ldt, 0, 1, Rm(0), 0, ldr, 1, out + psi_at[mu], r, 1,
                              accumulate, q);
                else
                    rc = gemm(4, l * nn, r, nb, sp, Tm(0), ldt, t_inter[mu] ? sp : (int64_t)szT[mu], 1, Rm(0), (int64_t)szR[jr],
                              ldr, 1, out + psi_at[mu], r, 1, accumulate, q);
                if (rc) return rc;
            } else if (mu < d - 1) {
                // Psi[q,k,c] = sum_{p'} T[q,k,p'] R[p',c]   (M=(q,k), N=c, K=p')
                for (int b = 0; b < nb; ++b) { p.A[b] = Tm(b); p.B[b] = Rm(b); p.C[b] = outb(b) + psi_at[mu]; }
                {
                    StreamSmallArgs ss{nb, (int)(l * nn), (int)sp, (int)r, p.A, ldt, p.B, ldr, p.C, r, accumulate};
                    g_cls = 4;
                    const int fz = (l * nn < (1ll << 30)) ? stream_small_try(ss, q, stq) : 0;
                    g_cls = NCLS - 1;
                    if (fz < 0) return fz;
                    if (fz == 0) CK(gemm_batch(4, nb, desc2(l * nn, r, 1, sp, ldt, 0, 1, 0, ldr, 1, r, 1, accumulate), p, q, stq));
                }
            } else if (sum) {
                // last mode: Psi_{d-1}[q,k,0] = sum_b T_b[q,k,0]
                CK(ttsk_sum_slices(out + psi_at[mu], Tm(0), nb, szT[mu], (size_t)(l * nn), accumulate, q));
            } else {
                // last mode: Psi_{d-1}[q,k,0] = T[q,k,0]
                for (int b = 0; b < nb; ++b) {
                    if (accumulate) CK(ttsk_axpby(outb(b) + psi_at[mu], Tm(b), 1.0, 1.0, (size_t)(l * nn), q));
                    else TTSK_HIP(hipMemcpyAsync(outb(b) + psi_at[mu], Tm(b), (size_t)(l * nn) * 8, hipMemcpyDeviceToDevice, stq));
                }
            }
        }
        if (mu < d - 1 && do_omega) {
            const int64_t l = l_hi[mu] - l_lo[mu];
            if (sum) {
                // Omega[q, c] = sum_{b, p} L_b[p, q] R_b[p, c]
                if (packedL(mu) && packedR(jr))
                    rc = gemm(5, l, r, 1, (int64_t)nb * sp, Lp(0, mu) + l_lo[mu], 1, 0, lt[mu + 1], Rm(0), 0, ldr, 1,
                              out + om_at[mu], r, 1, accumulate, q);
                else
                    rc = gemm(5, l, r, nb, sp, Lp(0, mu) + l_lo[mu], 1, (int64_t)szL[mu], lt[mu + 1], Rm(0), (int64_t)szR[jr], ldr,
                              1, out + om_at[mu], r, 1, accumulate, q);
                if (rc) return rc;
                return TTSK_OK;
            }
            // Omega_mu = L_mu[:, lo:hi]^T R_mu[:, lo:hi]; the workspace blocks and the outputs of a
            // batch are equally spaced, so the nb products are one batched launch
            ttsk_gemm_desc od = desc2(l, r, 1, sp, 1, 0, lt[mu + 1], 0, ldr, 1, r, 1, accumulate);
            od.batch = nb; od.a_b = (int64_t)szL[mu]; od.b_b = (int64_t)szR[jr]; od.c_b = out_stride;
            g_cls = 5;
            rc = ttsk_gemm(&od, Lp(0, mu) + l_lo[mu], Rm(0), outb(0) + om_at[mu], nullptr, q);
            g_cls = NCLS - 1;
            if (rc) return rc;
        }
        return TTSK_OK;
    };

    // ---- the schedule.  Right chain on the caller's stream, left chain on the helper `aux`, steps enqueued
    // alternately (neither chain waits for the host to have queued the other: 335 -> 319 us for one C3 tensor); a
    // join; then Psi / Omega dealt over the two streams.  (Starting Psi_mu / Omega_mu on a third stream as soon as
    // left step mu and right step d-2-mu are done was measured too: 325 us eager, 377 us replayed from a hipGraph,
    // slower for 6-8 tensors -- the early products compete with the chain steps for the CUs.  Not kept.)
    CK(ttsk_stream_wait(aux, stream));   // fork
    for (int t = 0; t < d; ++t) {
        if (t < d - 1) CK(right_step(t));
        CK(left_step(t));
    }
    // both chains are needed from here on, on both streams
    if (co) {
        for (int j = 0; j < d - 1; ++j) { co->Rc[j] = Rp(0, j); co->r_stride[j] = (int64_t)szR[j]; }
        for (int mu = 0; mu < d - 1; ++mu) { co->Lc[mu] = no_left ? nullptr : Lp(0, mu); co->l_stride[mu] = (int64_t)szL[mu]; }
        if (no_left) return TTSK_OK;
        CK(ttsk_stream_wait(stream, aux));   // join
        // Omega_mu = L_mu^T R_mu: batched launches over (tensor, mode) when the modes share a shape
        bool same = true;
        for (int mu = 1; mu < d - 1; ++mu)
            same = same && s[mu + 1] == s[1] && lt[mu + 1] == lt[1] && rt[d - 1 - mu] == rt[d - 1];
        if (same) {
            BatchPtrs o{};
            int cnt = 0;
            for (int b = 0; b < nb; ++b)
                for (int mu = 0; mu < d - 1; ++mu) {
                    o.A[cnt] = Lp(b, mu); o.B[cnt] = Rp(b, d - 2 - mu); o.C[cnt] = co->omega[(size_t)b * (d - 1) + mu];
                    if (++cnt == SK_MAXB || (b == nb - 1 && mu == d - 2)) {
                        CK(gemm_batch(5, cnt, desc2(lt[1], rt[d - 1], 1, s[1], 1, 0, lt[1], 0, rt[d - 1], 1, rt[d - 1], 1, 0), o, stream, st));
                        cnt = 0;
                    }
                }
        } else {
            for (int b = 0; b < nb; ++b)
                for (int mu = 0; mu < d - 1; ++mu) {
                    BatchPtrs o1{};
                    o1.A[0] = Lp(b, mu); o1.B[0] = Rp(b, d - 2 - mu); o1.C[0] = co->omega[(size_t)b * (d - 1) + mu];
                    const int64_t l = lt[mu + 1], r = rt[d - 1 - mu];
                    CK(gemm_batch(5, 1, desc2(l, r, 1, s[mu + 1], 1, 0, l, 0, r, 1, r, 1, 0), o1, stream, st));
                }
        }
        return TTSK_OK;
    }
    // Psi_0 = X_0 R_0 needs the right chain only (this stream): queued BEFORE the join, it runs while this stream would wait for the
    // left chain (the join costs ~16 us of cross-queue latency) instead of standing behind the Omega launch at the very end
    // (one C3 tensor: 0.254 -> 0.239 ms).  (Psi_{d-1}, a copy of the left chain's last T, stays behind the join: on the helper
    // stream in front of it, it would lengthen the chain the join waits for.)
    const bool ends_early = !sum && d >= 3;
    if (ends_early) CK(psi_omega(0, stream, true, false));
    CK(ttsk_stream_wait(stream, aux));
    CK(ttsk_stream_wait(aux, stream));
    // Few tensors of one shape in every mode (one C3 tensor: 4 Psi, 5 Omega of equal shapes): the interior Psi
    // products as ONE launch of the streamed kernel, all Omega as one batched launch -- the tail after the chains is
    // then two launches deep on either stream instead of five.
    bool grouped = !sum && d >= 4 && nb * (d - 1) <= SK_MAXB;
    for (int mu = 1; mu < d - 1 && grouped; ++mu)
        grouped = n[mu] == n[1] && s[mu] == s[1] && s[mu + 1] == s[1] && t_inter[mu] == t_inter[1] &&
                  l_hi[mu - 1] - l_lo[mu - 1] == l_hi[0] - l_lo[0] && lt[mu] == lt[1] && lt[mu + 1] == lt[1] &&
                  r_hi[d - 2 - mu] - r_lo[d - 2 - mu] == r_hi[0] - r_lo[0] && rt[d - 1 - mu] == rt[1];
    grouped = grouped && s[d - 1] == s[1] && l_hi[d - 2] - l_lo[d - 2] == l_hi[0] - l_lo[0] && lt[d - 1] == lt[1] &&
              r_hi[d - 2] - r_lo[d - 2] == r_hi[0] - r_lo[0] && rt[d - 1] == rt[1];
    if (grouped) {
        const int64_t sp = s[1], nn = n[1], l = l_hi[0] - l_lo[0], r = r_hi[0] - r_lo[0], ldr = rt[1];
        const int64_t ldt = t_inter[1] ? (int64_t)nb * sp : sp;
        BatchPtrs p{};
        int cnt = 0;
        for (int mu = 1; mu < d - 1; ++mu)
            for (int b = 0; b < nb; ++b, ++cnt) {
                const int jr = d - 2 - mu;
                p.A[cnt] = t_inter[mu] ? ws0 + offT[mu] + (size_t)(l_lo[mu - 1] * nn) * ldt + (size_t)b * sp
                                       : Tp0(b, mu) + (size_t)(l_lo[mu - 1] * nn) * sp;
                p.B[cnt] = Rp(b, jr) + r_lo[jr];
                p.C[cnt] = outb(b) + psi_at[mu];
            }
        StreamSmallArgs ss{cnt, (int)(l * nn), (int)sp, (int)r, p.A, ldt, p.B, ldr, p.C, r, accumulate};
        g_cls = 4;
        const int fz = (l * nn < (1ll << 30)) ? stream_small_try(ss, stream, st) : 0;
        g_cls = NCLS - 1;
        if (fz < 0) return fz;
        if (fz == 0) grouped = false;                 // shape outside the streamed kernel's cover: mode by mode
        else {
            BatchPtrs o{};
            cnt = 0;
            for (int mu = 0; mu < d - 1; ++mu)
                for (int b = 0; b < nb; ++b, ++cnt) {
                    const int jr = d - 2 - mu;
                    o.A[cnt] = Lp(b, mu) + l_lo[mu]; o.B[cnt] = Rp(b, jr) + r_lo[jr]; o.C[cnt] = outb(b) + om_at[mu];
                }
            CK(gemm_batch(5, cnt, desc2(l, r, 1, sp, 1, 0, lt[1], 0, ldr, 1, r, 1, accumulate), o, aux, st_aux));
            if (!ends_early) CK(psi_omega(0, aux, true, false));
            CK(psi_omega(d - 1, stream, true, false));
        }
    }
    if (!grouped) {
        // sum of tensors, all modes of one shape: the d - 1 Omega_mu = sum_{(b,p)} L_all[(b,p), q] R_all[(b,p), c] as ONE
        // batched small launch (they were 5 launches of 15 us at C5, K = 640 each)
        bool om_batched = false;
        if (sum && d - 1 <= SK_MAXB && d >= 3) {
            bool same = true;
            for (int mu = 0; mu < d - 1 && same; ++mu)
                same = s[mu + 1] == s[1] && lt[mu + 1] == lt[1] && rt[d - 1 - mu] == rt[d - 1] && packedL(mu) && packedR(d - 2 - mu) &&
                       l_hi[mu] - l_lo[mu] == l_hi[0] - l_lo[0] && r_hi[d - 2 - mu] - r_lo[d - 2 - mu] == r_hi[d - 2] - r_lo[d - 2];
            if (same) {
                BatchPtrs o{};
                for (int mu = 0; mu < d - 1; ++mu) {
                    const int jr = d - 2 - mu;
                    o.A[mu] = Lp(0, mu) + l_lo[mu]; o.B[mu] = Rp(0, jr) + r_lo[jr]; o.C[mu] = out + om_at[mu];
                }
                const int64_t l = l_hi[0] - l_lo[0], r = r_hi[d - 2] - r_lo[d - 2];
                CK(gemm_batch(5, d - 1, desc2(l, r, 1, (int64_t)nb * s[1], 1, 0, lt[1], 0, rt[d - 1], 1, r, 1, accumulate), o, aux, st_aux));
                om_batched = true;
            }
        }
        // ... and the interior Psi of the sum, all modes of one shape: the K chunks of EVERY mode as the problems of ONE launch of
        // the streamed kernel (4 modes x 4 chunks at C5: 800 workgroups instead of four launches of 200 on two streams)
        bool psi_batched = false;
        if (sum && sum_psi_chunks && d >= 4) {
            const int64_t sp = s[2], nn = n[1], l = l_hi[0] - l_lo[0], r = r_hi[d - 3] - r_lo[d - 3], ldr = rt[d - 2];
            bool same = l * nn >= 1024 && l * nn < (1ll << 30);
            for (int mu = 1; mu < d - 1 && same; ++mu) {
                const int jr = d - 2 - mu;
                same = t_inter[mu] && packedR(jr) && n[mu] == nn && s[mu + 1] == sp && l_hi[mu - 1] - l_lo[mu - 1] == l &&
                       r_hi[jr] - r_lo[jr] == r && rt[jr + 1] == ldr;
            }
            const int64_t K = (int64_t)nb * sp;
            int nch = 0;
            for (int t = 1; same && t <= nb && !nch; ++t)
                if (K % t == 0 && (K / t) % 4 == 0 && 4 * (K / t / 4 + 6) * ((r + 3) / 4 * 4) * 8 <= 150 * 1024) nch = t;
            if (same && nch > 1 && (d - 2) * nch <= SK_MAXB && (d - 2) * nch <= 2 * nb && szPs >= (size_t)(l * nn * r)) {
                const int64_t Kc = K / nch, ldt = (int64_t)nb * sp;
                double *blk0 = ws0 + offPs;
                BatchPtrs p{};
                int cnt = 0;
                for (int mu = 1; mu < d - 1; ++mu) {
                    const int jr = d - 2 - mu;
                    const double *T0 = ws0 + offT[mu] + (size_t)(l_lo[mu - 1] * nn) * ldt, *R0 = Rp(0, jr) + r_lo[jr];
                    for (int cidx = 0; cidx < nch; ++cidx, ++cnt) {
                        p.A[cnt] = T0 + (size_t)cidx * Kc; p.B[cnt] = R0 + (size_t)cidx * Kc * ldr; p.C[cnt] = blk0 + (size_t)cnt * szPs;
                    }
                }
                StreamSmallArgs ss{cnt, (int)(l * nn), (int)Kc, (int)r, p.A, ldt, p.B, ldr, p.C, r, 0};
                g_cls = 4;
                const int fz = stream_small_try(ss, stream, st);
                g_cls = NCLS - 1;
                if (fz < 0) return fz;
                if (fz == 1) {
                    psi_batched = true;
                    CK(ttsk_stream_wait(aux, stream));
                    for (int mu = 1; mu < d - 1; ++mu)
                        CK(ttsk_sum_slices(out + psi_at[mu], blk0 + (size_t)(mu - 1) * nch * szPs, nch, szPs, (size_t)(l * nn * r), accumulate,
                                           (mu & 1) ? aux : stream));
                }
            }
        }
        for (int mu = 0; mu < d; ++mu) {
            const bool psi_here = !(psi_batched && mu >= 1 && mu < d - 1) && !(ends_early && mu == 0);
            if (!psi_here && (om_batched || mu == d - 1)) continue;
            CK(psi_omega(mu, (mu & 1) ? aux : stream, psi_here, !om_batched));
        }
    }
    CK(ttsk_stream_wait(stream, aux));   // join
    return TTSK_OK;
#undef CK
}

}  // extern "C"

namespace ttsk {

int tt_chains(int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt, const double *const *X,
              const double *const *DL, const double *const *DR, TTChains *out, int stream)
{
    return tt_chains_batch(1, d, n, s, lt, rt, X, DL, DR, out, stream);
}

int tt_chains_batch(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt, const double *const *X,
                    const double *const *DL, const double *const *DR, TTChains *out, int stream)
{
    if (d < 2 || d > 64 || !out) { set_error("tt_chains: bad argument"); return TTSK_ERR_ARG; }
    std::vector<int64_t> zero(d, 0), lhi(d, 1), rhi(d, 1), ones(d + 1, 1);
    for (int mu = 0; mu < d - 1; ++mu) {
        if (out->want_left) lhi[mu] = lt[mu + 1];
        rhi[mu] = rt[mu + 1];
    }
    if (nb < 1 || nb > SK_MAXB) { set_error("tt_chains: %d tensors (1 .. %d)", nb, SK_MAXB); return TTSK_ERR_ARG; }
    return tt_sketch_core(nb, d, n, s, out->want_left ? lt : ones.data(), zero.data(), lhi.data(), rt, zero.data(), rhi.data(), X,
                          DL, DR, nullptr, 0, 0, stream, false, out);
}

}  // namespace ttsk

