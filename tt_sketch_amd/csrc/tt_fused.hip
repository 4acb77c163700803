// TT input x TT DRMs: the whole streaming sketch (both chains, Omega, Psi) as one C call.
// See include/ttsk.h (ttsk_tt_sketch) for the contract and the reference lines it replaces.
#include <cstdlib>
#include <vector>
#include "common.h"
#include "skinny.h"
#include "tt_step.h"

namespace ttsk {

constexpr int SMAX_ROWS = 112;   // rows a fused step addresses in an X slab (tt_step.hip SMAX)
constexpr int NCLS = 8;
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
    } else if (family == 3)        // streamed x small: tiles = column tiles of the small operand, ak = shared fifth block
        snprintf(g_kname[g_cls], sizeof(g_kname[0]), "skinny_s_kernel<%d, %s, 5>", tiles, ak ? "true" : "false");
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

// Right-chain GEMM1 in the layout the long-K kernel wants, T[q][k][p''] (p'' contiguous):
// a product batched over k whose batch index joins the streamed index.  1 = launched by the
// streamed-x-small kernel, 0 = shape not covered (caller uses the generic layout), < 0 = error.
static int right_gemm1_batched(int cls, int64_t rho, int64_t sn, int64_t nn, int64_t sp, const double *Rc,
                               const double *X, double *T, int stream, hipStream_t st)
{
    ttsk_gemm_desc d{};
    d.batch = nn; d.M = rho; d.N = sn; d.Ko = 1; d.Ki = sp;
    d.a_b = 0; d.a_m = 1; d.a_ko = 0; d.a_ki = rho;
    d.b_b = sp; d.b_ko = 0; d.b_ki = 1; d.b_n = nn * sp;
    d.c_b = sn; d.c_m = nn * sn; d.c_n = 1;
    d.alpha = 1.0; d.accumulate = 0; d.split_k = 0;
    g_cls = cls;
    const int rc = skinny_try(d, Rc, X, T, nullptr, stream, st);
    g_cls = NCLS - 1;
    return rc;
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
    TTSK_STREAM(st, stream);
    TTSK_ARG(d >= 2, "ttsk_tt_sketch: need d >= 2, got %d", d);
    TTSK_ARG(n && s && lt && l_lo && l_hi && rt && r_lo && r_hi && X && DL && DR && out,
             "ttsk_tt_sketch: NULL argument");
    TTSK_ARG(s[0] == 1 && s[d] == 1 && lt[0] == 1 && rt[0] == 1, "ttsk_tt_sketch: boundary ranks must be 1");
    for (int mu = 0; mu < d - 1; ++mu) {
        TTSK_ARG(0 <= l_lo[mu] && l_lo[mu] <= l_hi[mu] && l_hi[mu] <= lt[mu + 1],
                 "ttsk_tt_sketch: left rank slice %d out of range", mu);
        TTSK_ARG(0 <= r_lo[mu] && r_lo[mu] <= r_hi[mu] && r_hi[mu] <= rt[mu + 1],
                 "ttsk_tt_sketch: right rank slice %d out of range", mu);
    }
    // The right chain runs on the caller's stream, the left chain on a helper stream (the two are
    // independent until Psi / Omega need both); the Psi products are then dealt over both.  The
    // helper is forked from / joined into `stream`, so callers (and hipGraph capture) see one stream.
    // TTSK_SINGLE_STREAM=1 (or an active profiling pass) keeps everything on `stream`: per-kernel event
    // times are then free of cross-stream sharing and match rocprofv3's kernel durations.
    const char *single = getenv("TTSK_SINGLE_STREAM");
    const int aux = (single && single[0] == '1') ? stream : (stream + 1) % TTSK_NUM_STREAMS;
    TTSK_STREAM(st_aux, aux);
    (void)st_aux;
    // workspace (per stream slot DRIVER of `stream`): Lc[mu] (s[mu+1] x lt[mu+1]), Rc[j] (s[d-1-j] x rt[j+1]),
    // one T buffer per left mode (kept for the Psi phase) and one T buffer for the right chain
    size_t tot = 0;
    std::vector<size_t> offL(d - 1), offR(d - 1), offT(d);
    for (int mu = 0; mu < d - 1; ++mu) { offL[mu] = tot; tot += (size_t)s[mu + 1] * lt[mu + 1]; }
    for (int j = 0; j < d - 1; ++j) { offR[j] = tot; tot += (size_t)s[d - 1 - j] * rt[j + 1]; }
    for (int mu = 1; mu < d; ++mu) { offT[mu] = tot; tot += (size_t)lt[mu] * n[mu] * s[mu + 1]; }
    size_t tr_max = 0;
    for (int mu = 1; mu < d - 1; ++mu) {
        size_t tr = (size_t)rt[d - 1 - mu] * n[mu] * s[mu];
        tr_max = tr > tr_max ? tr : tr_max;
    }
    const size_t offTR = tot;
    tot += tr_max;
    double *ws = (double *)scratch(stream, SCRATCH_DRIVER, tot * 8);
    if (!ws) return TTSK_ERR_HIP;
    int rc;
#define CK(x) do { rc = (x); if (rc) return rc; } while (0)
    // ---- fused schedule: every interior chain step is ONE kernel (tt_step.hip) + the slab sum.
    // The left steps carry Psi and therefore need R_mu, which the right chain produces last-to-first:
    // the right chain runs to completion first, then the left chain; the small products (first /
    // last mode, Omega) go to the helper stream.
    // Measured at the bench workload: 42 us per fused step; the sequential right-then-left schedule
    // it needs totals 0.51 ms per sketch against 0.41 ms for the two-stream GEMM schedule below, so
    // the fused step is opt-in (TTSK_FUSED_STEP=1) until its MFMA duty improves.
    const char *fenv = getenv("TTSK_FUSED_STEP");
    bool fuse = fenv != nullptr && fenv[0] == '1';
    for (int mu = 1; mu < d - 1 && fuse; ++mu) {
        const int j = d - 1 - mu;
        fuse = tt_step_fits(s[mu + 1], s[mu], rt[j], rt[j + 1], 0, 0, (int64_t)SMAX_ROWS * n[mu] * s[mu + 1]) &&
               tt_step_fits(s[mu], s[mu + 1], lt[mu], lt[mu + 1], r_hi[d - 2 - mu] - r_lo[d - 2 - mu],
                            l_hi[mu - 1] - l_lo[mu - 1], (int64_t)(s[mu] + 32) * n[mu] * s[mu + 1]);
    }
    if (fuse) {
        std::vector<double *> psi_at(d), om_at(d - 1);
        {
            double *p = out;
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
        CK(ttsk_stream_wait(aux, stream));   // fork
        // L_0[p',q'] = sum_k X_0[0,k,p'] D_0[0,k,q'] on the helper stream (independent of the right chain)
        CK(gemm(5, s[1], lt[1], 1, n[0], X[0], 1, 0, s[1], DL[0], 0, lt[1], 1, ws + offL[0], lt[1], 1, 0, aux));
        // right chain
        for (int j = 0; j < d - 1; ++j) {
            const int mu = d - 1 - j;
            const int64_t sp = s[mu + 1], sn = s[mu], nn = n[mu], rho = rt[j], rhop = rt[j + 1];
            double *Rn = ws + offR[j];
            if (j == 0) {
                CK(gemm(5, sn, rhop, 1, nn, X[mu], nn * sp, 0, sp, DR[j], 0, rhop, 1, Rn, rhop, 1, 0, stream));
                continue;
            }
            StepArgs g{};
            g.s_in = (int)sp; g.s_out = (int)sn; g.rho = (int)rho; g.rhop = (int)rhop; g.r = 0;
            g.x_k = sp; g.x_a = nn * sp; g.x_b = 1; g.x_extent = sn * nn * sp;
            g.X = X[mu]; g.Cin = ws + offR[j - 1]; g.ldc = rho; g.c_extent = sp * rho;
            g.D = DR[j]; g.d_q = nn * rhop; g.d_k = rhop; g.d_extent = rho * nn * rhop;
            CK(tt_step_launch(true, nn, g, Rn, rhop, stream));
        }
        CK(ttsk_stream_wait(stream, aux));   // L_0 ready
        // Psi_0[0,k,c] = sum_{p'} X_0[0,k,p'] R_0[p',c] and Omega_0 on the helper stream
        CK(ttsk_stream_wait(aux, stream));   // right chain complete
        {
            const int jr = d - 2;
            const double *Rm = ws + offR[jr] + r_lo[jr];
            const int64_t ldr = rt[jr + 1], r = r_hi[jr] - r_lo[jr];
            CK(gemm(5, n[0], r, 1, s[1], X[0], s[1], 0, 1, Rm, 0, ldr, 1, psi_at[0], r, 1, accumulate, aux));
            CK(gemm(5, l_hi[0] - l_lo[0], r, 1, s[1], ws + offL[0] + l_lo[0], 1, 0, lt[1], Rm, 0, ldr, 1, om_at[0],
                    r, 1, accumulate, aux));
        }
        // left chain with Psi
        for (int mu = 1; mu < d; ++mu) {
            const int64_t sn = s[mu], sp = s[mu + 1], nn = n[mu], lfull = lt[mu];
            const int64_t l = l_hi[mu - 1] - l_lo[mu - 1];
            const double *Lc = ws + offL[mu - 1];
            if (mu == d - 1) {
                // last mode: Psi_{d-1}[q,k,0] = sum_p L[p, lo+q] X[p,k,0]
                CK(gemm(5, l, nn, 1, sn, Lc + l_lo[mu - 1], 1, 0, lfull, X[mu], 0, nn * sp, 1, psi_at[mu], nn, 1,
                        accumulate, stream));
                break;
            }
            const int jr = d - 2 - mu;
            const double *Rm = ws + offR[jr] + r_lo[jr];
            const int64_t ldr = rt[jr + 1], r = r_hi[jr] - r_lo[jr];
            StepArgs g{};
            g.s_in = (int)sn; g.s_out = (int)sp; g.rho = (int)lfull; g.rhop = (int)lt[mu + 1]; g.r = (int)r;
            g.q_lo = (int)l_lo[mu - 1]; g.q_cnt = (int)l; g.accumulate_psi = accumulate;
            g.x_k = sp; g.x_a = 1; g.x_b = nn * sp; g.x_extent = sn * nn * sp;
            g.X = X[mu]; g.Cin = Lc; g.ldc = lfull; g.c_extent = sn * lfull;
            g.D = DL[mu]; g.d_q = nn * lt[mu + 1]; g.d_k = lt[mu + 1]; g.d_extent = lfull * nn * lt[mu + 1];
            g.R = Rm; g.ldr = ldr; g.r_extent = (sp - 1) * ldr + r;
            g.Psi = psi_at[mu]; g.psi_q = nn * r; g.psi_k = r; g.psi_c = 1;
            CK(tt_step_launch(false, nn, g, ws + offL[mu], lt[mu + 1], stream));
            // Omega_mu = L_mu[:, lo:hi]^T R_mu[:, lo:hi] on the helper stream
            CK(ttsk_stream_wait(aux, stream));
            CK(gemm(5, l_hi[mu] - l_lo[mu], r, 1, sp, ws + offL[mu] + l_lo[mu], 1, 0, lt[mu + 1], Rm, 0, ldr, 1,
                    om_at[mu], r, 1, accumulate, aux));
        }
        CK(ttsk_stream_wait(stream, aux));   // join
        return TTSK_OK;
    }
    CK(ttsk_stream_wait(aux, stream));   // fork

    // ---- right chain (stream): walks modes d-1, ..., 1 on the transposed tensor (views only).
    // Xt_j[p,k,p''] = X_mu[p'',k,p], mu = d-1-j.
    {
        double *T = ws + offTR;
        for (int j = 0; j < d - 1; ++j) {
            const int mu = d - 1 - j;
            const int64_t sp = s[mu + 1], sn = s[mu], nn = n[mu], rho = rt[j], rhop = rt[j + 1];
            double *Rn = ws + offR[j];
            if (j == 0) {
                // Rc_0[p'',q'] = sum_k X[p'',k,0] E[0,k,q']
                CK(gemm(5, sn, rhop, 1, nn, X[mu], nn * sp, 0, sp, DR[j], 0, rhop, 1, Rn, rhop, 1, 0, stream));
            } else {
                const double *Rc = ws + offR[j - 1];                   // (sp x rho)
                // preferred: T[q, k, p''] = sum_p Rc[p,q] X[p'',k,p] (batched over k), so that both
                // operands of GEMM2 are contiguous along their output index
                const int fast = right_gemm1_batched(0, rho, sn, nn, sp, Rc, X[mu], T, stream, st);
                if (fast < 0) return fast;
                if (fast == 1) {
                    // Rn[p'', q'] = sum_{q,k} T[q,k,p''] E[q,k,q']
                    CK(gemm(1, sn, rhop, rho, nn, T, 1, nn * sn, sn, DR[j], nn * rhop, rhop, 1, Rn, rhop, 1, 0, stream));
                } else {
                    // T[q, p'', k] = sum_p Rc[p,q] X[p'',k,p]    (M=q, N=(p'',k), K=p)
                    CK(gemm(0, rho, sn * nn, 1, sp, Rc, 1, 0, rho, X[mu], 0, 1, sp, T, sn * nn, 1, 0, stream));
                    // Rn[p'', q'] = sum_{q,k} T[q,p'',k] E[q,k,q']
                    CK(gemm(1, sn, rhop, rho, nn, T, nn, sn * nn, 1, DR[j], nn * rhop, rhop, 1, Rn, rhop, 1, 0, stream));
                }
            }
        }
    }
    // ---- left chain (aux): L_mu and the shared products T_mu = L_{mu-1}^T X_mu
    for (int mu = 0; mu < d; ++mu) {
        const int64_t sn = s[mu], sp = s[mu + 1], nn = n[mu];
        if (mu == 0) {
            // L_0[p',q'] = sum_k X_0[0,k,p'] D_0[0,k,q']
            CK(gemm(5, sp, lt[1], 1, nn, X[0], 1, 0, sp, DL[0], 0, lt[1], 1, ws + offL[0], lt[1], 1, 0, aux));
            continue;
        }
        const int64_t lfull = lt[mu];
        const double *Lc = ws + offL[mu - 1];                   // (sn x lfull)
        double *T = ws + offT[mu];
        // T[q,k,p'] = sum_p Lc[p,q] X[p,k,p']      (M=q (all lfull columns), N=(k,p'), K=p)
        CK(gemm(mu == d - 1 ? 5 : 2, lfull, nn * sp, 1, sn, Lc, 1, 0, lfull, X[mu], 0, nn * sp, 1, T, nn * sp, 1, 0, aux));
        if (mu < d - 1)
            // L_mu[p',q'] = sum_{q,k} T[q,k,p'] D[q,k,q']
            CK(gemm(3, sp, lt[mu + 1], 1, lfull * nn, T, 1, 0, sp, DL[mu], 0, lt[mu + 1], 1, ws + offL[mu],
                    lt[mu + 1], 1, 0, aux));
    }
    // both chains are needed from here on, on both streams
    CK(ttsk_stream_wait(stream, aux));
    CK(ttsk_stream_wait(aux, stream));

    // ---- Psi and Omega, dealt over the two streams
    std::vector<double *> psi_at(d), om_at(d - 1);
    {
        double *p = out;
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
    for (int mu = 0; mu < d; ++mu) {
        const int64_t sp = s[mu + 1], nn = n[mu];
        const int q = (mu & 1) ? aux : stream;
        hipStream_t stq = stream_of(q);
        // right contraction of modes mu+1.. : Rc[j] with j = d-2-mu, columns [r_lo, r_hi)
        const int jr = d - 2 - mu;
        const double *Rm = mu < d - 1 ? ws + offR[jr] + r_lo[jr] : nullptr;
        const int64_t ldr = mu < d - 1 ? rt[jr + 1] : 0, r = mu < d - 1 ? r_hi[jr] - r_lo[jr] : 1;
        if (mu == 0) {
            // Psi_0[0,k,c] = sum_{p'} X_0[0,k,p'] R_0[p',c]
            CK(gemm(5, nn, r, 1, sp, X[0], sp, 0, 1, Rm, 0, ldr, 1, psi_at[0], r, 1, accumulate, q));
        } else {
            const int64_t l = l_hi[mu - 1] - l_lo[mu - 1];
            const double *Ts = ws + offT[mu] + l_lo[mu - 1] * nn * sp;   // rows of the rank slice
            if (mu < d - 1) {
                // Psi[q,k,c] = sum_{p'} T[q,k,p'] R[p',c]   (M=(q,k), N=c, K=p')
                CK(gemm(4, l * nn, r, 1, sp, Ts, sp, 0, 1, Rm, 0, ldr, 1, psi_at[mu], r, 1, accumulate, q));
            } else {
                // last mode: Psi_{d-1}[q,k,0] = T[q,k,0]
                if (accumulate) CK(ttsk_axpby(psi_at[mu], Ts, 1.0, 1.0, (size_t)(l * nn), q));
                else TTSK_HIP(hipMemcpyAsync(psi_at[mu], Ts, (size_t)(l * nn) * 8, hipMemcpyDeviceToDevice, stq));
            }
        }
        if (mu < d - 1) {
            // Omega_mu = L_mu[:, lo:hi]^T R_mu[:, lo:hi]
            const int64_t l = l_hi[mu] - l_lo[mu];
            CK(gemm(5, l, r, 1, sp, ws + offL[mu] + l_lo[mu], 1, 0, lt[mu + 1], Rm, 0, ldr, 1, om_at[mu], r, 1,
                    accumulate, q));
        }
    }
    CK(ttsk_stream_wait(stream, aux));   // join
#undef CK
    return TTSK_OK;
}

}  // extern "C"
