// Host side of the chunked fused chain step (chain_wide.h): chunk plan, LDS plan, wave table, launch geometry.
#include <cstdlib>
#include <cstring>
#include "chain_wide.h"

namespace ttsk {

#define TTSK_CW_DECL(T) \
    int launch_chain_wide_##T(const ChainWide &a, int nn, int sn, bool wt, int unr, size_t lds, int grid, hipStream_t st);
TTSK_CW_DECL(0a) TTSK_CW_DECL(0b) TTSK_CW_DECL(1a) TTSK_CW_DECL(1b) TTSK_CW_DECL(2a) TTSK_CW_DECL(2b) TTSK_CW_DECL(3a)
TTSK_CW_DECL(3b) TTSK_CW_DECL(4a) TTSK_CW_DECL(4b) TTSK_CW_DECL(5a) TTSK_CW_DECL(5b) TTSK_CW_DECL(6a) TTSK_CW_DECL(6b)
#undef TTSK_CW_DECL

// chunk structures that are instantiated: columns = 16 tiles + 4 strips
static const int CW_NQ[7] = {1, 1, 2, 2, 3, 3, 4}, CW_SQ[7] = {0, 2, 0, 2, 0, 2, 0};

static int launch_chain_wide(int ci, const ChainWide &a, int nn, int sn, bool wt, int unr, size_t lds, int grid, hipStream_t st)
{
    const bool lo = nn <= 4;
    switch (ci) {
    case 0: return lo ? launch_chain_wide_0a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_0b(a, nn, sn, wt, unr, lds, grid, st);
    case 1: return lo ? launch_chain_wide_1a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_1b(a, nn, sn, wt, unr, lds, grid, st);
    case 2: return lo ? launch_chain_wide_2a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_2b(a, nn, sn, wt, unr, lds, grid, st);
    case 3: return lo ? launch_chain_wide_3a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_3b(a, nn, sn, wt, unr, lds, grid, st);
    case 4: return lo ? launch_chain_wide_4a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_4b(a, nn, sn, wt, unr, lds, grid, st);
    case 5: return lo ? launch_chain_wide_5a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_5b(a, nn, sn, wt, unr, lds, grid, st);
    default: return lo ? launch_chain_wide_6a(a, nn, sn, wt, unr, lds, grid, st) : launch_chain_wide_6b(a, nn, sn, wt, unr, lds, grid, st);
    }
}

static int cw_num_cu()
{
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

static void cw_tile_split(int r, int &nf, int &str)
{
    const int rem = r % 16;
    nf = r / 16;
    if (rem == 0) str = 0;
    else if (rem <= 4) str = 1;
    else if (rem <= 8) str = 2;
    else { nf += 1; str = 0; }
}

// TTSK_CHAIN_WIDE: 0 = never, 1 = when it pays (default), 2 = whenever the shape is covered
static int cw_mode()
{
    static int m = [] { const char *e = getenv("TTSK_CHAIN_WIDE"); return e ? atoi(e) : 1; }();
    return m;
}

// Row tiles -> waves.  Wave w sits on SIMD w & 3 (two waves per SIMD); the tiles are dealt so that the SIMDs
// carry equal shares, SIMD 3 the lightest one: its second wave is the loader.  Returns false if the rows do not fit.
// tpw > 1: `tpw` tensors per workgroup, each with ceil(NT / 2) (or NT) waves of its own, in order.
static bool cw_wave_table(int NT, bool mt2_ok, int tpw, ChainWide &a, bool &uses_mt2)
{
    memset(a.tile0, -1, sizeof(a.tile0));
    memset(a.tile1, -1, sizeof(a.tile1));
    memset(a.slot, 0, sizeof(a.slot));
    a.loader = 7;
    uses_mt2 = false;
    if (tpw > 1) {
        const int wpt = mt2_ok ? (NT + 1) / 2 : NT;
        if (tpw * wpt > 7) return false;
        int w = 0;
        for (int sl = 0; sl < tpw; ++sl)
            for (int t = 0; t < NT; ++w) {
                a.slot[w] = (signed char)sl;
                a.tile0[w] = (signed char)t++;
                if (mt2_ok && t < NT) { a.tile1[w] = (signed char)t++; uses_mt2 = true; }
            }
        return true;
    }
    if (NT <= 7) {
        for (int w = 0; w < NT; ++w) a.tile0[w] = (signed char)w;
        return true;
    }
    if (!mt2_ok || NT > 11) return false;          // SIMD 3 has one compute wave: at most two tiles there
    int next = 0;
    for (int s = 0; s < 4; ++s) {
        const int t = NT / 4 + (s < NT % 4 ? 1 : 0);
        const int first = s == 3 ? t : (t + 1) / 2;  // tiles of wave s; the rest go to wave s + 4
        if (first > 2 || t - first > 2) return false;
        a.tile0[s] = (signed char)next++;
        if (first == 2) { a.tile1[s] = (signed char)next++; uses_mt2 = true; }
        if (t - first >= 1) a.tile0[s + 4] = (signed char)next++;
        if (t - first == 2) { a.tile1[s + 4] = (signed char)next++; uses_mt2 = true; }
    }
    return next == NT;
}

int chain_wide_try(const ChainStepArgs &c, int stream, hipStream_t st, bool force)
{
    const int mode = force ? 2 : cw_mode();
    if (!mode || c.nb < 1 || c.nb > SK_MAXB) return 0;
    if (c.J < 1 || c.K1 < 1 || c.A < 1 || c.A2 < 1 || c.n < 1) return 0;
    if (((uintptr_t)c.E & 7) || c.x_j < 0 || c.x_k < 0 || c.x_c < 0 || c.w_c < c.A) return 0;
    int nn, sn;
    cw_tile_split(c.A2, nn, sn);
    if (nn + (sn ? 1 : 0) > 10 || (nn == 10 && sn)) return 0;
    const int NT = (c.J + 15) / 16;
    if (NT > 11 || (NT > 7 && nn + (sn ? 1 : 0) > 7)) return 0;     // rows beyond 7 tiles need waves with two tiles
    ChainWide a{};
    a.nb = c.nb; a.n = c.n; a.K1 = c.K1; a.A = c.A; a.A2 = c.A2; a.J = c.J;
    a.w_c = c.w_c; a.x_j = c.x_j; a.x_k = c.x_k; a.x_c = c.x_c; a.x_extent = c.x_extent;
    a.E = c.E;
    a.A2P = c.A2 + (c.A2 & 1);
    // k-blocks of phase A in straight-line runs of 25 or 5, padded to whole runs (chain_fused.hip)
    const int kb = (c.K1 + 3) / 4;
    const int pad25 = (kb + 24) / 25 * 25, pad5 = (kb + 4) / 5 * 5;
    const int unr = pad25 <= pad5 + 1 ? 25 : 5;
    const int KB1 = unr == 25 ? pad25 : pad5;
    // ---- chunk plan: the fewest chunks of A whose images fit the LDS; two row tiles per wave only with <= 3 chunk tiles.
    // Few rows per tensor (NT <= 3) and a batch: several tensors per workgroup, as many as have waves and LDS.
    bool uses_mt2 = false;
    int ci = -1, nac = 0, tpw = 1;
    size_t lds = 0;
    const int cus = cw_num_cu();
    const bool out_mt2 = nn + (sn ? 1 : 0) <= 7;
    for (int tryn = 1; tryn <= c.A && tryn <= 64 && ci < 0; ++tryn) {
        const int need = (int)((cdiv(c.A, tryn) + 3) / 4 * 4);
        for (int i = 0; i < 7 && ci < 0; ++i) {
            const int ap = 16 * CW_NQ[i] + 4 * CW_SQ[i];
            if (ap < need) continue;
            const bool mt2_ok = CW_NQ[i] <= 3 && out_mt2;
            int want = 1;
            if (c.nb >= 2 && NT <= 3) {
                want = 7 / (mt2_ok ? (NT + 1) / 2 : NT);
                if (want > c.nb) want = c.nb;
                if (want < 1) want = 1;
            }
            const int64_t wimg = ((int64_t)4 * KB1 * ap + 1) & ~(int64_t)1;
            const int64_t units = (int64_t)2 * (ap / 4) * a.A2P;
            const int eunits = (int)cdiv(units, 64) * 64;
            if (eunits / 64 > CF_MAX_DMA) break;
            // tensors per workgroup: as many as fit beside ONE E image (two images if they fit as well)
            while (want > 1 && ((size_t)want * wimg + (size_t)eunits * 2) * 8 > 160 * 1024) --want;
            if (!cw_wave_table(NT, mt2_ok, want, a, uses_mt2)) { if (ap >= 64) break; continue; }
            const int ebase = (int)((int64_t)want * wimg);
            const size_t one = ((size_t)ebase + (size_t)eunits * 2) * 8, two = ((size_t)ebase + (size_t)eunits * 4) * 8;
            if (one > 160 * 1024) break;               // larger structures only need more: more chunks
            ci = i; nac = tryn; tpw = want;
            a.ac = need; a.ebase = ebase; a.eunits = eunits; a.wimg = (int)wimg;
            a.ebuf2 = two <= 160 * 1024 ? 1 : 0;
            lds = a.ebuf2 ? two : one;
        }
    }
    if (ci < 0) return 0;
    // TT rank much smaller than the DRM rank (C5: 20 against 50 / 100): the two-launch form merges the rows of ALL tensors
    // of the batch into one long-K product (no 20 -> 32 row padding) and wins -- measured per right step: 81 us against 101 us
    // with seven tensors per workgroup here (128 us with one); TTSK_CHAIN_WIDE=2 takes this kernel anyway
    if (mode == 1 && 2 * c.K1 < c.A) return 0;
    if (!cw_wave_table(NT, CW_NQ[ci] <= 3 && out_mt2, tpw, a, uses_mt2)) return 0;
    a.tpw = tpw;
    a.nac = nac;
    // 32-bit byte offsets: the X walk (incl. the prefetch one slice past the end) and T
    if ((c.x_extent + c.x_k + ((int64_t)KB1 * 4 + 32) * c.x_c) * 8 >= (1ll << 32) - 64) return 0;
    const bool wt = c.T != nullptr;
    a.t_extent = (int64_t)c.A * c.n * c.J;
    if (wt && (a.t_extent + (int64_t)80 * c.n * c.J) * 8 >= (1ll << 32) - 64) return 0;
    // geometry: one workgroup per CU, each a contiguous range of slices of one chunk
    const int ng = (c.nb + tpw - 1) / tpw;            // workgroup groups of tensors
    int wpp = cus / (ng * nac) > 0 ? cus / (ng * nac) : 1;
    if (wpp > c.n) wpp = c.n;
    a.wpp = wpp;
    const int units = wpp * nac;
    a.xcd_map = (units % 8 == 0) ? 1 : 0;
    for (int b = 0; b < c.nb; ++b) {
        if ((uintptr_t)c.X[b] & 7) return 0;
        a.W[b] = c.W[b];
        a.X[b] = c.X[b];
        a.T[b] = wt ? c.T[b] : nullptr;
    }
    const int64_t nslab = (int64_t)c.nb * units;
    a.slab = (double *)scratch(stream, SCRATCH_GEMM, (size_t)nslab * c.J * c.A2 * 8 + 64);
    if (!a.slab) return TTSK_ERR_HIP;
    const bool prof = prof_on();
    if (prof) {
        char name[96];
        snprintf(name, sizeof(name), "chain_wide_kernel<%d, %d, %d, %d, %s, %d, %s>", CW_NQ[ci], CW_SQ[ci], nn, sn, wt ? "true" : "false", unr,
                 (CW_NQ[ci] <= 3 && nn + (sn ? 1 : 0) <= 7) ? "true" : "false");
        prof_open_named(st, -2, 2.0 * c.nb * (double)c.n * c.J * ((double)c.K1 * c.A + (double)c.A * c.A2), name);
    }
    int rc = launch_chain_wide(ci, a, nn, sn, wt, unr, lds, ng * units, st);
    if (rc == TTSK_OK) {
        ReduceOut ro{};
        for (int b = 0; b < c.nb; ++b) ro.C[b] = c.Out[b];
        rc = launch_r_reduce(st, a.slab, units, c.J, c.A2, 1, (int64_t)c.J, ro, c.nb, (int64_t)c.A2, (int64_t)1, 1.0, 0);
    }
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : (rc == 1 ? 0 : rc);
}

}  // namespace ttsk

using namespace ttsk;

extern "C" int ttsk_chain_step_wide(int nb, int n, int K1, int A, int A2, int J, const double *const *W, int64_t w_c,
                                    const double *const *X, int64_t x_j, int64_t x_k, int64_t x_c, int64_t x_extent,
                                    const double *E, double *const *T, double *const *Out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(W && X && E && Out && nb >= 1, "ttsk_chain_step_wide: NULL argument");
    ChainStepArgs c{nb, n, K1, A, A2, J, W, w_c, X, x_j, x_k, x_c, x_extent, E, T, Out};
    const int rc = chain_wide_try(c, stream, st, true);
    if (rc == 0) {
        set_error("ttsk_chain_step_wide: shape (n=%d K1=%d A=%d A2=%d J=%d nb=%d) is not covered by the chunked fused kernel", n, K1,
                  A, A2, J, nb);
        return TTSK_ERR_UNSUPPORTED;
    }
    return rc < 0 ? rc : TTSK_OK;
}
