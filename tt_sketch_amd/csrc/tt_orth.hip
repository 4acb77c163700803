// orthogonal_sketch / hmt_sketch of a tensor train with tensor-train DRMs as ONE call
// (reference sketch_dispatch.py:160-193, 202-275 with method = orthogonal / hmt; sketch.py:44-151).
//
//   1. the right DRM chain, and for `orthogonal` the left chain and Omega_mu = L_mu^T R_mu: the kernels of the
//      streaming sketch (tt_chains, tt_fused.hip), two streams;
//   2. orthogonal: the d - 1 pseudo-inverses as batched launches (ttsk_pinv_batch_deferred) and W_mu = R_mu Omega_mu^+
//      (hmt: W_mu = R_mu);
//   3. mode by mode: T = Q-chain (x) X_mu,  M = T W_mu  (= Psi_mu Omega_mu^+ without ever forming Psi_mu),
//      Q~_mu = qr(M) by CholeskyQR2 (R's diagonal positive),  next Q-chain = Q~_mu^T T  (T is formed once and used twice);
//   4. beside that chain, on stream + 1: LAPACK's Householder column signs of every mode (qr_signs) and, as soon as the
//      chain has read a core for the last time, its row and column signs applied (apply_signs).
// Also here: ttsk_tt_assemble (assemble_sketched_tt as one call).
//
// Nothing is read back: the verdicts of the fast factorisations accumulate in the stream's deferred flag
// (ttsk_deferred_status); the caller repeats a rejected sketch on the robust path (ttsk_pinv / ttsk_qr_thin).
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "skinny.h"
#include "tt_chain.h"
#include "linalg_int.h"

using namespace ttsk;

namespace {

int gemm2(int64_t M, int64_t N, int64_t K, const double *A, int64_t a_m, int64_t a_k, const double *B, int64_t b_k, int64_t b_n,
          double *C, int stream, double alpha = 1.0, int accumulate = 0)
{
    ttsk_gemm_desc d{};
    d.batch = 1; d.M = M; d.N = N; d.Ko = 1; d.Ki = K;
    d.a_m = a_m; d.a_ki = a_k; d.b_ki = b_k; d.b_n = b_n; d.c_m = N; d.c_n = 1;
    d.alpha = alpha; d.accumulate = accumulate;
    return ttsk_gemm(&d, A, B, C, nullptr, stream);
}

}  // namespace

extern "C" {

int ttsk_pinv_batch_deferred(int count, const double *const *dev_omegas, int64_t l, int64_t r, double *const *dev_pinvs, int stream);
int ttsk_pinv_batch(int count, const double *const *dev_omegas, int64_t l, int64_t r, double *const *dev_pinvs, int stream);

int ttsk_tt_orth_sketch(int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt,
                        const double *const *X, const double *const *DL, const double *const *DR,
                        double *const *cores_out, double *const *omega_out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(d >= 2 && d <= 64 && n && s && rt && X && DR && cores_out, "ttsk_tt_orth_sketch: bad argument");
    const bool orth = DL != nullptr;                       // orthogonal: left DRM and Omega; hmt: right DRM only
    TTSK_ARG(!orth || (lt && omega_out), "ttsk_tt_orth_sketch: the orthogonal method needs lt and omega_out");
    TTSK_ARG(s[0] == 1 && s[d] == 1 && rt[0] == 1 && (!orth || lt[0] == 1), "ttsk_tt_orth_sketch: boundary ranks must be 1");
    int *sticky = deferred_flag(stream);
    auto rr = [&](int mu) { return rt[d - 1 - mu]; };                          // right sketch rank of mode mu < d - 1
    auto kk = [&](int mu) { return mu < 0 ? (int64_t)1 : (orth ? lt[mu + 1] : rr(mu)); };   // rank of the output cores
    bool ok = fast_solves() && sticky != nullptr;
    int64_t mmax = 0, kmax = 0, smax = 0, tmax = 0;
    for (int mu = 0; mu < d; ++mu) {
        TTSK_ARG(n[mu] >= 1 && s[mu + 1] >= 1 && X[mu] && cores_out[mu], "ttsk_tt_orth_sketch: bad mode %d", mu);
        const int64_t m = kk(mu - 1) * n[mu];
        tmax = std::max(tmax, m * s[mu + 1]);
        smax = std::max(smax, s[mu + 1]);
        if (mu == d - 1) break;
        TTSK_ARG(DR[mu] && rt[mu + 1] >= 1 && (!orth || (DL[mu] && lt[mu + 1] >= 1 && omega_out[mu])), "ttsk_tt_orth_sketch: bad DRM core %d", mu);
        ok = ok && kk(mu) <= QR_CHOL_MAX_N && m >= kk(mu);
        if (orth) ok = ok && std::min(lt[mu + 1], rr(mu)) <= QR_CHOL_MAX_N;
        mmax = std::max(mmax, m);
        kmax = std::max(kmax, kk(mu));
    }
    if (!ok || d - 1 > SK_MAXB) {
        set_error("ttsk_tt_orth_sketch: outside the one-call path (ranks, shapes or TTSK_FAST_SOLVES=0)");
        return TTSK_ERR_UNSUPPORTED;
    }
    int rc;
#define CK(x) do { rc = (x); if (rc < 0) return rc; } while (0)
    // ---- 1. chains (and Omega)
    TTChains ch{};
    ch.want_left = orth ? 1 : 0;
    ch.omega = omega_out;
    CK(tt_chains(d, n, s, lt, rt, X, DL, DR, &ch, stream));
    // ---- workspace of this driver
    auto blk = [](size_t v) { return (v + 31) & ~(size_t)31; };
    // (the d - 1 pseudo-inverses as batched launches when the Omega share one shape with min(l, r) <= 128)
    bool one_shape = orth;
    size_t pmax = 0, pws = 0;
    int64_t lmax = 1;
    for (int mu = 0; orth && mu < d - 1; ++mu) {
        one_shape = one_shape && lt[mu + 1] == lt[1] && rr(mu) == rr(0) && s[mu + 1] == s[1];
        pmax = std::max(pmax, (size_t)rr(mu) * lt[mu + 1]);
        pws = std::max(pws, pinv_deferred_ws_elems(lt[mu + 1], rr(mu)));
        lmax = std::max(lmax, lt[mu + 1]);
    }
    one_shape = one_shape && std::min(lt[1], rr(0)) <= 128;
    const size_t szP = orth ? blk(pmax) : 0, szW = orth ? blk((size_t)smax * lmax) : 0;
    const size_t szL = blk((size_t)smax * kmax), szT = blk((size_t)tmax);
    const size_t szQ = blk(qr_ws_elems(mmax, (int)kmax));
    const size_t szS = blk((size_t)kmax), szSW = kmax > 128 ? blk((size_t)kmax * kmax) : 0, szPW = one_shape ? 0 : blk(pws);
    double *ws = (double *)scratch(stream, SCRATCH_ORTH, ((size_t)(d - 1) * (szP + szW + szS) + 2 * szL + szT + szQ + szSW + szPW) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *P0 = ws, *W0 = P0 + (size_t)(d - 1) * szP, *Lb = W0 + (size_t)(d - 1) * szW, *T = Lb + 2 * szL, *qws = T + szT, *Sb = qws + szQ;
    double *signs_work = szSW ? Sb + (size_t)(d - 1) * szS : nullptr, *pinv_work = Sb + (size_t)(d - 1) * szS + szSW;
    // ---- 2. W_mu = R_mu Omega_mu^+  (s[mu+1] x l)
    std::vector<const double *> W(d - 1);
    if (orth) {
        const double *om[SK_MAXB];
        double *pv[SK_MAXB];
        for (int mu = 0; mu < d - 1; ++mu) { om[mu] = omega_out[mu]; pv[mu] = P0 + (size_t)mu * szP; }
        if (one_shape) {
            CK(ttsk_pinv_batch_deferred(d - 1, om, lt[1], rr(0), pv, stream));
        } else {
            for (int mu = 0; mu < d - 1; ++mu) {
                rc = pinv_deferred(om[mu], lt[mu + 1], rr(mu), pv[mu], stream, st, pinv_work, sticky);
                if (rc < 0) return rc;
                if (rc == 0) { set_error("ttsk_tt_orth_sketch: pseudo-inverse outside the fast path"); return TTSK_ERR_UNSUPPORTED; }
            }
        }
        const double *A[SK_MAXB], *B[SK_MAXB];
        double *C[SK_MAXB];
        for (int mu = 0; mu < d - 1; ++mu) { A[mu] = ch.Rc[d - 2 - mu]; B[mu] = pv[mu]; C[mu] = W0 + (size_t)mu * szW; W[mu] = C[mu]; }
        int done = 0;
        if (one_shape) {
            ttsk_gemm_desc g{};
            g.batch = 1; g.M = s[1]; g.N = lt[1]; g.Ko = 1; g.Ki = rr(0);
            g.a_m = rr(0); g.a_ki = 1; g.b_ki = lt[1]; g.b_n = 1; g.c_m = lt[1]; g.c_n = 1; g.alpha = 1.0;
            CK(done = small_try_batch(g, d - 1, A, B, C, stream, st));
        }
        if (!done)
            for (int mu = 0; mu < d - 1; ++mu)
                CK(gemm2(s[mu + 1], lt[mu + 1], rr(mu), A[mu], rr(mu), 1, B[mu], lt[mu + 1], 1, C[mu], stream));
    } else {
        for (int mu = 0; mu < d - 1; ++mu) W[mu] = ch.Rc[d - 2 - mu];
    }
    // ---- 3. the modes.  The Householder column signs are NOT on this chain: the factors come out with CholeskyQR's
    // signs (Q~), the chain goes on with Q~ -- a sign flip of column b of Q_mu flips row b of the next chain matrix and
    // with it the rows (b, i) of the next unfolding, which changes neither that unfolding's R nor Q~ beyond the same row
    // flips -- and the sign reconstruction of every mode (an n-step elimination in one workgroup) runs on the helper
    // stream beside the next mode's products.  One pass over the cores at the end applies S_{mu-1} (rows) and S_mu (columns).
    static const int signs_beside = [] { const char *e = getenv("TTSK_ORTH_SIGNS_BESIDE"); return e ? atoi(e) : 1; }();
    bool beside = signs_beside && kmax <= QR_CHOL_MAX_N;
    for (int mu = 1; mu < d - 1; ++mu) beside = beside && kk(mu - 1) * n[mu] >= 2 * kk(mu);   // (the one-workgroup Householder QR of a nearly square unfolding signs its Q itself: only mode 0 may take it)
    const int aux = (stream + 1) % TTSK_NUM_STREAMS;
    TTSK_STREAM(st_aux, aux);
    double *Lc = Lb, *Ln = Lb + szL;                      // Q-chain (s[mu] x k_{mu-1}) of this mode / of the next one
    std::vector<const double *> Sg(d, nullptr);           // signs of mode mu (nullptr: none / all ones)
    // core[a, i, b] *= S_{mu-1}[a] S_mu[b] on the helper stream, behind the sign kernels and behind the main stream's work so far
    static const int fix_beside = [] { const char *e = getenv("TTSK_ORTH_FIX_BESIDE"); return e ? atoi(e) : 1; }();
    auto fix_core = [&](int mu) -> int {
        const double *spv = mu > 0 ? Sg[mu - 1] : nullptr, *snv = mu < d - 1 ? Sg[mu] : nullptr;
        if (!spv && !snv) return TTSK_OK;
        double *c = cores_out[mu];
        const int k0 = (int)kk(mu - 1), nv = (int)n[mu], k1 = mu < d - 1 ? (int)kk(mu) : 1;
        int r = ttsk_stream_wait(aux, stream);
        if (r < 0) return r;
        return apply_signs(1, &c, &spv, &snv, &k0, &nv, &k1, st_aux);
    };
    for (int mu = 0; mu < d; ++mu) {
        const int64_t kp = kk(mu - 1), nn = n[mu], sn = s[mu], sp = s[mu + 1], m = kp * nn;
        const double *Tm;
        if (mu == 0) {
            Tm = X[0];                                    // (n_0 x s_1)
        } else {
            // T[q, i, p'] = sum_p Lc[p, q] X[p, i, p']
            double *dst = mu == d - 1 ? cores_out[mu] : T;
            CK(gemm2(kp, nn * sp, sn, Lc, 1, kp, X[mu], nn * sp, 1, dst, stream));
            Tm = dst;
        }
        if (mu == d - 1) {
            if (mu == 0) TTSK_HIP(hipMemcpyAsync(cores_out[0], X[0], (size_t)nn * 8, hipMemcpyDeviceToDevice, st));
            break;
        }
        const int64_t k = kk(mu);
        double *Q = cores_out[mu];
        CK(gemm2(m, k, sp, Tm, sp, 1, W[mu], k, 1, Q, stream));                     // M = T W
        rc = qr_cholesky(Q, m, k, stream, st, qws, sticky, beside);
        if (rc < 0) return rc;
        if (rc == 0) { set_error("ttsk_tt_orth_sketch: QR outside the fast path"); return TTSK_ERR_UNSUPPORTED; }
        if (beside && rc == 2 && mu > 0) { set_error("ttsk_tt_orth_sketch: unexpected small unfolding"); return TTSK_ERR_UNSUPPORTED; }
        if (beside && rc != 2) {
            double *Sm = Sb + (size_t)mu * szS;
            CK(ttsk_stream_wait(aux, stream));                                     // Q~ is there
            CK(qr_signs(Q, (int)k, m == k ? 1 : 0, mu > 0 ? Sg[mu - 1] : nullptr, (int)nn, Sm, st_aux, signs_work));
            Sg[mu] = Sm;
        }
        // next chain matrix Ln[p', q'] = sum_{(q,i)} T[(q,i), p'] Q[(q,i), q']
        CK(gemm2(sp, k, m, Tm, 1, sp, Q, k, 1, Ln, stream));
        std::swap(Lc, Ln);
        if (beside && fix_beside) CK(fix_core(mu));       // the chain has read Q~ for the last time
    }
    if (beside) {
        if (!fix_beside)
            for (int mu = 0; mu < d - 1; ++mu) CK(fix_core(mu));
        CK(fix_core(d - 1));
        CK(ttsk_stream_wait(stream, aux));
    }
#undef CK
    return TTSK_OK;
}

// verdict of one tensor of a batch: move the stream's deferred flag to the tensor's slot and clear it
__global__ void orth_take_flag_kernel(int *sticky, int *dst)
{
    if (threadIdx.x == 0) { *dst = *sticky; *sticky = 0; }
}

// verdict of a fused batch: one flag for all its tensors
__global__ void orth_spread_flag_kernel(int *sticky, int *dst, int count)
{
    if (threadIdx.x == 0) {
        const int v = *sticky;
        *sticky = 0;
        for (int b = 0; b < count; ++b) dst[b] = v;
    }
}

// The batch as ONE chain of launches: every step of ttsk_tt_orth_sketch over all `count` tensors at once -- the DRM chains and Omega
// (tt_chains_batch), the pseudo-inverses and W = R Omega^+ (batched over (tensor, mode)), then per mode T, M = T W, CholeskyQR2
// (qr_cholesky_batch), the sign reconstruction (qr_signs_batch, helper stream) and the next chain matrix, each one launch with
// `count` problems.  ~90 launches per batch instead of per tensor, on operands count times larger.  The fast factorisations'
// verdict is one flag for the whole batch (a rejection repeats every tensor of it on its own).
// 1 = done, 0 = shapes outside this path (nothing written that the caller may not overwrite), < 0 = error.
static int orth_batch_fused(int count, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt,
                            const double *const *X, const double *const *DL, const double *const *DR, double *const *cores_out,
                            double *const *omega_out, int *dev_status, int stream)
{
    TTSK_STREAM(st, stream);
    const bool orth = DL != nullptr;
    static const int on = [] { const char *e = getenv("TTSK_ORTH_BATCH_FUSED"); return e ? atoi(e) : 1; }();
    if (!on || count < 2 || count > 16 || d < 2 || d - 1 > SK_MAXB || !fast_solves()) return 0;
    int *sticky = deferred_flag(stream);
    if (!sticky) return 0;
    auto rr = [&](int mu) { return rt[d - 1 - mu]; };
    auto kk = [&](int mu) { return mu < 0 ? (int64_t)1 : (orth ? lt[mu + 1] : rr(mu)); };
    int64_t mmax = 0, kmax = 0, smax = 0, tmax = 0, lmax = 1;
    size_t pmax = 0;
    for (int mu = 0; mu < d; ++mu) {
        const int64_t m = kk(mu - 1) * n[mu];
        tmax = std::max(tmax, m * s[mu + 1]);
        smax = std::max(smax, s[mu + 1]);
        if (mu == d - 1) break;
        // every unfolding at least twice as tall as wide (CholeskyQR2 + signs beside the chain), factors within one workgroup's LDS
        if (kk(mu) > 128 || m < 2 * kk(mu)) return 0;
        if (orth && (std::min(lt[mu + 1], rr(mu)) > 128 || lt[mu + 1] != lt[1] || rr(mu) != rr(0) || s[mu + 1] != s[1])) return 0;
        mmax = std::max(mmax, m);
        kmax = std::max(kmax, kk(mu));
        if (orth) { pmax = std::max(pmax, (size_t)rr(mu) * lt[mu + 1]); lmax = std::max(lmax, lt[mu + 1]); }
    }
    int rc;
#define CK(x) do { rc = (x); if (rc < 0) return rc; } while (0)
    // ---- 1. chains (and Omega) of all tensors
    TTChains ch{};
    ch.want_left = orth ? 1 : 0;
    ch.omega = omega_out;
    CK(tt_chains_batch(count, d, n, s, lt, rt, X, DL, DR, &ch, stream));
    auto Rc = [&](int b, int j) { return ch.Rc[j] + (size_t)b * ch.r_stride[j]; };
    // ---- workspace
    auto blk = [](size_t v) { return (v + 31) & ~(size_t)31; };
    const size_t szP = orth ? blk(pmax) : 0, szW = orth ? blk((size_t)smax * lmax) : 0;
    const size_t szL = blk((size_t)smax * kmax), szT = blk((size_t)tmax), szS = blk((size_t)kmax);
    const size_t szQ = blk(qr_batch_ws_elems(count, mmax, (int)kmax));
    const size_t per_t = (size_t)(d - 1) * (szP + szW + szS) + 2 * szL + szT;
    double *ws = (double *)scratch(stream, SCRATCH_ORTH, ((size_t)count * per_t + szQ) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *P0 = ws, *W0 = P0 + (size_t)count * (d - 1) * szP, *Sb = W0 + (size_t)count * (d - 1) * szW;
    double *Lb = Sb + (size_t)count * (d - 1) * szS, *Tb = Lb + (size_t)count * 2 * szL, *qws = Tb + (size_t)count * szT;
    auto Pv = [&](int b, int mu) { return P0 + ((size_t)b * (d - 1) + mu) * szP; };
    auto Wv = [&](int b, int mu) { return W0 + ((size_t)b * (d - 1) + mu) * szW; };
    auto Sv = [&](int b, int mu) { return Sb + ((size_t)b * (d - 1) + mu) * szS; };
    auto desc = [](int64_t M, int64_t N, int64_t K, int64_t a_m, int64_t a_k, int64_t b_k, int64_t b_n) {
        ttsk_gemm_desc g{};
        g.batch = 1; g.M = M; g.N = N; g.Ko = 1; g.Ki = K;
        g.a_m = a_m; g.a_ki = a_k; g.b_ki = b_k; g.b_n = b_n; g.c_m = N; g.c_n = 1; g.alpha = 1.0;
        return g;
    };
    // one product for every tensor: a batched launch where the shape has one, tensor by tensor otherwise
    auto prod = [&](const ttsk_gemm_desc &g, const double *const *A, const double *const *B, double *const *C) -> int {
        int r = skinny_try_batch(g, count, A, B, C, stream, st);
        if (r == 0) r = small_try_batch(g, count, A, B, C, stream, st);
        if (r != 0) return r < 0 ? r : TTSK_OK;
        for (int b = 0; b < count; ++b) {
            r = gemm2(g.M, g.N, g.Ki, A[b], g.a_m, g.a_ki, B[b], g.b_ki, g.b_n, C[b], stream);
            if (r < 0) return r;
        }
        return TTSK_OK;
    };
    // ---- 2. W_b,mu = R_b,mu Omega_b,mu^+
    std::vector<const double *> W((size_t)count * (d - 1));
    if (orth) {
        const double *om[SK_MAXB], *A[SK_MAXB], *B[SK_MAXB];
        double *pv[SK_MAXB], *C[SK_MAXB];
        int cnt = 0;
        for (int b = 0; b < count; ++b)
            for (int mu = 0; mu < d - 1; ++mu) {
                om[cnt] = omega_out[(size_t)b * (d - 1) + mu]; pv[cnt] = Pv(b, mu);
                A[cnt] = Rc(b, d - 2 - mu); B[cnt] = pv[cnt]; C[cnt] = Wv(b, mu);
                W[(size_t)b * (d - 1) + mu] = C[cnt];
                if (++cnt == SK_MAXB || (b == count - 1 && mu == d - 2)) {
                    CK(ttsk_pinv_batch_deferred(cnt, om, lt[1], rr(0), pv, stream));
                    ttsk_gemm_desc g = desc(s[1], lt[1], rr(0), rr(0), 1, lt[1], 1);
                    int done = small_try_batch(g, cnt, A, B, C, stream, st);
                    if (done < 0) return done;
                    if (!done)
                        for (int q = 0; q < cnt; ++q) CK(gemm2(s[1], lt[1], rr(0), A[q], rr(0), 1, B[q], lt[1], 1, C[q], stream));
                    cnt = 0;
                }
            }
    } else {
        for (int b = 0; b < count; ++b)
            for (int mu = 0; mu < d - 1; ++mu) W[(size_t)b * (d - 1) + mu] = Rc(b, d - 2 - mu);
    }
    // ---- 3. the modes, every step over all tensors (signs beside the chain as in the single call)
    const int aux = (stream + 1) % TTSK_NUM_STREAMS;
    TTSK_STREAM(st_aux, aux);
    std::vector<const double *> Sg((size_t)count * d, nullptr);
    auto fix_cores = [&](int mu) -> int {
        double *c[16];
        const double *spv[16], *snv[16];
        int k0[16], nv[16], k1[16];
        bool any = false;
        for (int b = 0; b < count; ++b) {
            spv[b] = mu > 0 ? Sg[(size_t)b * d + mu - 1] : nullptr;
            snv[b] = mu < d - 1 ? Sg[(size_t)b * d + mu] : nullptr;
            any = any || spv[b] || snv[b];
            c[b] = cores_out[(size_t)b * d + mu];
            k0[b] = (int)kk(mu - 1); nv[b] = (int)n[mu]; k1[b] = mu < d - 1 ? (int)kk(mu) : 1;
        }
        if (!any) return TTSK_OK;
        int r = ttsk_stream_wait(aux, stream);
        if (r < 0) return r;
        return apply_signs(count, c, spv, snv, k0, nv, k1, st_aux);
    };
    const double *Tm[16], *Lc[16], *Xm[16], *Wm[16];
    double *Ln[16], *Q[16], *Td[16];
    int flip = 0;
    for (int mu = 0; mu < d; ++mu) {
        const int64_t kp = kk(mu - 1), nn = n[mu], sn = s[mu], sp = s[mu + 1], m = kp * nn;
        for (int b = 0; b < count; ++b) {
            Xm[b] = X[(size_t)b * d + mu];
            Lc[b] = Lb + ((size_t)b * 2 + flip) * szL;
            Ln[b] = Lb + ((size_t)b * 2 + (flip ^ 1)) * szL;
        }
        if (mu == 0) {
            for (int b = 0; b < count; ++b) Tm[b] = Xm[b];
        } else {
            for (int b = 0; b < count; ++b) { Td[b] = mu == d - 1 ? cores_out[(size_t)b * d + mu] : Tb + (size_t)b * szT; Tm[b] = Td[b]; }
            CK(prod(desc(kp, nn * sp, sn, 1, kp, nn * sp, 1), Lc, Xm, Td));                 // T[q, i, p'] = sum_p Lc[p, q] X[p, i, p']
        }
        if (mu == d - 1) {
            if (mu == 0)
                for (int b = 0; b < count; ++b)
                    TTSK_HIP(hipMemcpyAsync(cores_out[(size_t)b * d], Xm[b], (size_t)nn * 8, hipMemcpyDeviceToDevice, st));
            break;
        }
        const int64_t k = kk(mu);
        for (int b = 0; b < count; ++b) { Q[b] = cores_out[(size_t)b * d + mu]; Wm[b] = W[(size_t)b * (d - 1) + mu]; }
        CK(prod(desc(m, k, sp, sp, 1, k, 1), Tm, Wm, Q));                                     // M = T W
        rc = qr_cholesky_batch(count, Q, m, (int)k, stream, st, qws, sticky);
        if (rc < 0) return rc;
        if (rc == 0) { set_error("ttsk_tt_orth_sketch_batch: batched QR outside its cover at mode %d", mu); return TTSK_ERR_UNSUPPORTED; }
        {
            const double *cq[16], *sprev[16];
            double *sout[16];
            for (int b = 0; b < count; ++b) {
                cq[b] = Q[b]; sprev[b] = mu > 0 ? Sg[(size_t)b * d + mu - 1] : nullptr; sout[b] = Sv(b, mu);
                Sg[(size_t)b * d + mu] = sout[b];
            }
            CK(ttsk_stream_wait(aux, stream));                                               // the unsigned factors are there
            rc = qr_signs_batch(count, cq, (int)k, m == k ? 1 : 0, mu > 0 ? sprev : nullptr, (int)nn, sout, st_aux);
            if (rc < 0) return rc;
            if (rc == 0) { set_error("ttsk_tt_orth_sketch_batch: batched sign reconstruction outside its cover"); return TTSK_ERR_UNSUPPORTED; }
        }
        {
            const double *cq[16];
            for (int b = 0; b < count; ++b) cq[b] = Q[b];
            CK(prod(desc(sp, k, m, 1, sp, k, 1), Tm, cq, Ln));                                // next chain matrix Ln = T^T Q
        }
        flip ^= 1;
        CK(fix_cores(mu));                                                                     // the chain has read Q for the last time
    }
    CK(fix_cores(d - 1));
    CK(ttsk_stream_wait(stream, aux));
    hipLaunchKernelGGL(orth_spread_flag_kernel, dim3(1), dim3(64), 0, st, sticky, dev_status, count);
    TTSK_LAUNCH_CHECK();
#undef CK
    return 1;
}

// `count` tensor trains of ONE signature against ONE pair of DRMs (sketch.py:292-301: further tensors are sketched with the
// DRMs of the first): the sketches are independent chains of ~90 short, dependent launches each -- latency, not throughput
// (0.07 of the matrix peak per call at C3) -- so tensor b runs on the stream pair (2 b mod 8, + 1) beside three others; the
// library's streams are forked from `stream` and joined back into it.  dev_status[b] = 1 if a fast factorisation of tensor
// b was rejected (the caller repeats that tensor on the robust path); the deferred flags of the streams used end up clear.
int ttsk_tt_orth_sketch_batch(int count, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt,
                              const double *const *X, const double *const *DL, const double *const *DR,
                              double *const *cores_out, double *const *omega_out, int *dev_status, int stream)
{
    TTSK_STREAM(st, stream);
    (void)st;
    TTSK_ARG(count >= 1 && X && cores_out && dev_status, "ttsk_tt_orth_sketch_batch: bad argument");
    TTSK_ARG(!DL || omega_out, "ttsk_tt_orth_sketch_batch: the orthogonal method needs omega_out");
    int rc = orth_batch_fused(count, d, n, s, lt, rt, X, DL, DR, cores_out, omega_out, dev_status, stream);
    if (rc == 1) return TTSK_OK;
    if (rc < 0 && rc != TTSK_ERR_UNSUPPORTED) return rc;
    // (outside the fused batch's cover: the tensors as concurrent chains, below)
    bool used[TTSK_NUM_STREAMS] = {};
    const int lanes = TTSK_NUM_STREAMS / 2;
    auto pair_of = [&](int b) { return (stream + 2 * (b % lanes)) % TTSK_NUM_STREAMS; };
    auto join_all = [&]() {
        int worst = TTSK_OK;
        for (int k = 0; k < TTSK_NUM_STREAMS; ++k)
            if (used[k] && k != stream) { const int r = ttsk_stream_wait(stream, k); if (r < 0) worst = r; }
        return worst;
    };
    // the pairs' streams: behind what `stream` has queued so far (the DRM cores may still be in the making).  A flag left on
    // a pair's stream by an earlier, unrelated call is not this batch's: cleared.
    for (int b = 0; b < count && b < lanes; ++b) {
        const int q = pair_of(b), qa = (q + 1) % TTSK_NUM_STREAMS;
        if (q != stream) { if ((rc = ttsk_stream_wait(q, stream)) < 0) return rc; }
        if (qa != stream) { if ((rc = ttsk_stream_wait(qa, stream)) < 0) return rc; }
        used[q] = used[qa] = true;
        int *sticky = deferred_flag(q);
        if (!sticky) return TTSK_ERR_HIP;
        if (q != stream) hipLaunchKernelGGL(orth_take_flag_kernel, dim3(1), dim3(64), 0, stream_of(q), sticky, dev_status + b);
    }
    auto one = [&](int b) -> int {
        const int q = pair_of(b);
        int r = ttsk_tt_orth_sketch(d, n, s, lt, rt, X + (size_t)b * d, DL, DR, cores_out + (size_t)b * d,
                                    omega_out ? omega_out + (size_t)b * (d - 1) : nullptr, q);
        if (r < 0) return r;
        hipLaunchKernelGGL(orth_take_flag_kernel, dim3(1), dim3(64), 0, stream_of(q), deferred_flag(q), dev_status + b);
        return hipGetLastError() == hipSuccess ? TTSK_OK : TTSK_ERR_HIP;
    };
    // (Queuing the tensors of each stream pair from a thread of their own was measured: 0.65 against 0.67 ms per tensor at C3,
    // batch 8 -- the host's 0.4 ms per sketch is not what binds, the device-side sum of ~90 small kernels per sketch is; not kept.)
    for (int b = 0; b < count; ++b)
        if ((rc = one(b)) < 0) { (void)join_all(); return rc; }
    return join_all();
}

// assemble_sketched_tt (sketch.py:400-443) as ONE call: C_mu = Psi_mu pinv(Omega_mu) ("right", direction = 0) or
// pinv(Omega_{mu-1}) Psi_mu ("left", direction = 1).  The d - 1 (pseudo-inverse, product) pairs are independent: pair k
// runs on library stream k mod nstreams -- fast attempt, the Jacobi kernel queued behind it with the attempt's verdict
// as its predicate (ttsk_pinv_begin / ttsk_pinv_end without a rank read-back), then the product -- and every stream is
// joined into `stream` before the call returns.  No host synchronisation.
//   psi[mu]     (l_{mu-1}, n[mu], r_mu) contiguous, l_{-1} = r_{d-1} = 1;   omega[mu] (l_mu, r_mu), mu < d - 1
//   cores_out   right: (l_{mu-1}, n, l_mu), last = psi[d-1];  left: (r_{mu-1}, n, r_mu), first = psi[0]
//   work[k]     (r_k, l_k) doubles each: the pseudo-inverses (caller's, so that they outlive the call)
int ttsk_tt_assemble(int d, const int64_t *n, const int64_t *lr, const int64_t *rr, const double *const *psi,
                     const double *const *omega, double *const *cores_out, double *const *work, int direction, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(d >= 2 && d <= 64 && n && lr && rr && psi && omega && cores_out && work && (direction == 0 || direction == 1),
             "ttsk_tt_assemble: bad argument");
    int rc;
#define CK(x) do { rc = (x); if (rc < 0) return rc; } while (0)
    // Omega of one shape: the pseudo-inverses as batched launches on `stream`, then only the products are dealt out
    bool batched = d - 1 >= 2 && d - 1 <= SK_MAXB;
    for (int k = 0; k < d - 1; ++k) {
        TTSK_ARG(omega[k] && work[k] && lr[k] >= 1 && rr[k] >= 1, "ttsk_tt_assemble: bad Omega %d", k);
        batched = batched && lr[k] == lr[0] && rr[k] == rr[0];
    }
    static const int batch_on = [] { const char *e = getenv("TTSK_ASSEMBLE_BATCH"); return e ? atoi(e) : 1; }();
    static const int refine = [] { const char *e = getenv("TTSK_ASSEMBLE_REFINE"); return e ? atoi(e) : 1; }();
    if (batched && batch_on) {
        rc = ttsk_pinv_batch(d - 1, omega, lr[0], rr[0], work, stream);
        if (rc == TTSK_ERR_UNSUPPORTED) batched = false;
        else if (rc < 0) return rc;
    } else {
        batched = false;
    }
    // "right" direction, pseudo-inverses batched, interior modes of one size: their products (and the refinement's) as
    // batched launches on `stream` as well -- 3 launches + the copies instead of 3 per pair
    // A helper stream is forked the first time a pair actually lands on it and exactly the forked ones are joined at the end.
    // The end pairs (never grouped) get theirs here, behind the pseudo-inverses and IN FRONT of the grouped launches: they run
    // beside those instead of behind them (one C3 to_tt: ~30 us).
    bool forked[TTSK_NUM_STREAMS] = {};
    auto qof = [&](int k) { return (stream + 1 + k) % TTSK_NUM_STREAMS; };
    if (batched && direction == 0 && d - 1 >= 3)
        for (int k : {0, d - 2}) {
            const int q = qof(k);
            if (q != stream && !forked[q]) { CK(ttsk_stream_wait(q, stream)); forked[q] = true; }
        }
    std::vector<char> grouped(d - 1, 0);
    if (batched && direction == 0 && d - 1 >= 3) {
        std::vector<int> G;
        for (int k = 1; k < d - 1; ++k)
            if (n[k] == n[1] && psi[k] && cores_out[k]) G.push_back(k);
        const int nb = (int)G.size();
        const int64_t m = lr[0] * n[1], l = lr[0], r = rr[0];
        if (nb >= 2 && nb <= SK_MAXB) {
            double *Rall = refine ? (double *)scratch(stream, SCRATCH_ORTH, (size_t)nb * m * r * 8) : nullptr;
            if (refine && !Rall) return TTSK_ERR_HIP;
            const double *A[SK_MAXB], *B[SK_MAXB], *Om[SK_MAXB], *Rc[SK_MAXB];
            double *C[SK_MAXB], *R[SK_MAXB];
            for (int b = 0; b < nb; ++b) {
                A[b] = psi[G[b]]; B[b] = work[G[b]]; C[b] = cores_out[G[b]]; Om[b] = omega[G[b]];
                R[b] = Rall ? Rall + (size_t)b * m * r : nullptr; Rc[b] = R[b];
            }
            auto desc = [](int64_t M, int64_t N, int64_t K, double alpha, int acc) {
                ttsk_gemm_desc g{};
                g.batch = 1; g.M = M; g.N = N; g.Ko = 1; g.Ki = K;
                g.a_m = K; g.a_ki = 1; g.b_ki = N; g.b_n = 1; g.c_m = N; g.c_n = 1; g.alpha = alpha; g.accumulate = acc;
                return g;
            };
            rc = skinny_try_batch(desc(m, l, r, 1.0, 0), nb, A, B, C, stream, st);                       // C = Psi P
            if (rc < 0) return rc;
            if (rc == 1) {
                for (int b = 0; b < nb; ++b) grouped[G[b]] = 1;
                if (refine) {
                    // R <- Psi: the interior cores of a packed sketch lie behind one another: one copy
                    bool adjacent = true;
                    for (int b = 1; b < nb; ++b) adjacent = adjacent && A[b] == A[b - 1] + (size_t)m * r;
                    if (adjacent) {
                        TTSK_HIP(hipMemcpyAsync(R[0], A[0], (size_t)nb * m * r * 8, hipMemcpyDeviceToDevice, st));
                    } else {
                        for (int b = 0; b < nb; ++b)
                            TTSK_HIP(hipMemcpyAsync(R[b], A[b], (size_t)m * r * 8, hipMemcpyDeviceToDevice, st));
                    }
                    const double *Cc[SK_MAXB];
                    for (int b = 0; b < nb; ++b) Cc[b] = C[b];
                    rc = skinny_try_batch(desc(m, r, l, -1.0, 1), nb, Cc, Om, R, stream, st);           // R = Psi - C Omega
                    if (rc < 0) return rc;
                    if (rc == 0)
                        for (int b = 0; b < nb; ++b) CK(gemm2(m, r, l, C[b], l, 1, Om[b], r, 1, R[b], stream, -1.0, 1));
                    rc = skinny_try_batch(desc(m, l, r, 1.0, 1), nb, Rc, B, C, stream, st);              // C += R P
                    if (rc < 0) return rc;
                    if (rc == 0)
                        for (int b = 0; b < nb; ++b) CK(gemm2(m, l, r, R[b], r, 1, B[b], l, 1, C[b], stream, 1.0, 1));
                }
            }
        }
    }
    // (grouped pairs never touch a helper stream, so "k below the stream count" is not the moment to fork: with d - 1 >= 10 an
    // ungrouped pair could meet a stream whose earlier pair had been grouped and that was therefore never forked)
    for (int k = 0; k < d - 1; ++k) {
        if (grouped[k]) continue;
        const int q = qof(k);
        if (q != stream && !forked[q]) { CK(ttsk_stream_wait(q, stream)); forked[q] = true; }     // fork
        if (!batched) {
            CK(ttsk_pinv_begin(omega[k], lr[k], rr[k], -1.0, work[k], q));
            CK(ttsk_pinv_end(omega[k], lr[k], rr[k], -1.0, work[k], nullptr, q));
        }
        // One step of iterative refinement behind every product: C <- C + (Psi - C Omega) P.  What an explicitly formed
        // pseudo-inverse costs in accuracy is not its own error (a Newton-Schulz step on P changed nothing) but the product
        // Psi P: P's large entries (1 / sigma_min) meet the components of Psi that should cancel them, and the rounding of
        // that cancellation does not stay inside the singular directions.  The sketches of TT-GMRES iterates have singular
        // values down to 1e-7: the assembled tensor was off by 3e-9 (median of 300 solves, 2 % of them beyond 1e-7) where
        // scipy's lstsq -- the reference, utils.py:98-109 -- reproduces itself to 6e-14.  The residual is small and is
        // computed in full precision, so the same P corrects it: 5e-14 (median), 7e-12 (max of 200).  Two more products and
        // a copy per pair, on the pair's own stream.
        if (direction == 0) {
            // C_k[(a, i), b] = sum_c Psi_k[(a, i), c] P_k[c, b]
            const int64_t m = (k ? lr[k - 1] : 1) * n[k];
            TTSK_ARG(psi[k] && cores_out[k], "ttsk_tt_assemble: NULL core %d", k);
            CK(gemm2(m, lr[k], rr[k], psi[k], rr[k], 1, work[k], lr[k], 1, cores_out[k], q));
            if (refine) {
                double *R = (double *)scratch(q, SCRATCH_DRIVER, (size_t)m * rr[k] * 8);
                if (!R) return TTSK_ERR_HIP;
                TTSK_HIP(hipMemcpyAsync(R, psi[k], (size_t)m * rr[k] * 8, hipMemcpyDeviceToDevice, stream_of(q)));
                CK(gemm2(m, rr[k], lr[k], cores_out[k], lr[k], 1, omega[k], rr[k], 1, R, q, -1.0, 1));       // R = Psi - C Omega
                CK(gemm2(m, lr[k], rr[k], R, rr[k], 1, work[k], lr[k], 1, cores_out[k], q, 1.0, 1));         // C += R P
            }
        } else {
            // C_{k+1}[c, (i, b)] = sum_a P_k[c, a] Psi_{k+1}[a, (i, b)]
            const int64_t cols = n[k + 1] * (k + 1 < d - 1 ? rr[k + 1] : 1);
            TTSK_ARG(psi[k + 1] && cores_out[k + 1], "ttsk_tt_assemble: NULL core %d", k + 1);
            CK(gemm2(rr[k], cols, lr[k], work[k], lr[k], 1, psi[k + 1], cols, 1, cores_out[k + 1], q));
            if (refine) {
                double *R = (double *)scratch(q, SCRATCH_DRIVER, (size_t)lr[k] * cols * 8);
                if (!R) return TTSK_ERR_HIP;
                TTSK_HIP(hipMemcpyAsync(R, psi[k + 1], (size_t)lr[k] * cols * 8, hipMemcpyDeviceToDevice, stream_of(q)));
                CK(gemm2(lr[k], cols, rr[k], omega[k], rr[k], 1, cores_out[k + 1], cols, 1, R, q, -1.0, 1)); // R = Psi - Omega C
                CK(gemm2(rr[k], cols, lr[k], work[k], lr[k], 1, R, cols, 1, cores_out[k + 1], q, 1.0, 1));   // C += P R
            }
        }
    }
    const int e = direction == 0 ? d - 1 : 0;                                   // the core that is copied
    TTSK_ARG(psi[e] && cores_out[e], "ttsk_tt_assemble: NULL core %d", e);
    const int64_t sz = direction == 0 ? lr[d - 2] * n[d - 1] : n[0] * rr[0];
    if (cores_out[e] != psi[e]) TTSK_HIP(hipMemcpyAsync(cores_out[e], psi[e], (size_t)sz * 8, hipMemcpyDeviceToDevice, st));
    for (int q = 0; q < TTSK_NUM_STREAMS; ++q)
        if (forked[q]) CK(ttsk_stream_wait(stream, q));                           // join
#undef CK
    return TTSK_OK;
}

}  // extern "C"
