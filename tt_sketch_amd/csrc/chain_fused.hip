// Host side of the fused chain step (chain_fused.h): shape tests, LDS plan, launch geometry.
#include <cstdlib>
#include "chain_fused.h"

namespace ttsk {

int launch_chain_step_a(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st);
int launch_chain_step_b(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st);
int launch_chain_step_c(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st);
int launch_chain_step_d(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st);
int launch_chain_step_e(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st);

static int launch_chain_step(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st)
{
    if (nf <= 2) return launch_chain_step_a(a, nf, str, wt, ebuf, unr, lds, grid, st);
    if (nf <= 4) return launch_chain_step_b(a, nf, str, wt, ebuf, unr, lds, grid, st);
    if (nf == 5) return launch_chain_step_c(a, nf, str, wt, ebuf, unr, lds, grid, st);
    if (nf == 6) return launch_chain_step_d(a, nf, str, wt, ebuf, unr, lds, grid, st);
    return launch_chain_step_e(a, nf, str, wt, ebuf, unr, lds, grid, st);
}

static int cf_num_cu()
{
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

// rank -> full 16-wide tiles + 4-wide strips (a remainder of 9..15 is a zero-padded full tile)
static void tile_split(int r, int &nf, int &str)
{
    const int rem = r % 16;
    nf = r / 16;
    if (rem == 0) str = 0;
    else if (rem <= 4) str = 1;
    else if (rem <= 8) str = 2;
    else { nf += 1; str = 0; }
}

// TTSK_CHAIN_FUSED: 0 = never, 1 = when it pays (default), 2 = whenever the shape is covered (tests)
static int cf_mode()
{
    static int m = [] { const char *e = getenv("TTSK_CHAIN_FUSED"); return e ? atoi(e) : 1; }();
    return m;
}

int chain_fused_try(const ChainStepArgs &c, int stream, hipStream_t st, bool force)
{
    const int mode = force ? 2 : cf_mode();
    if (!mode || c.nb < 1 || c.nb > SK_MAXB) return 0;
    if (c.J < 1 || c.J > 112 || c.K1 < 1 || c.K1 > 128 || c.A < 4 || c.A2 < 4 || c.n < 1) return 0;
    if ((c.A2 & 1) || ((uintptr_t)c.E & 15)) return 0;                 // 16-byte units of E rows
    int nq, sq, nn, sn;
    tile_split(c.A, nq, sq);
    tile_split(c.A2, nn, sn);
    if (nq != nn || sq != sn || nq + (sq ? 1 : 0) > 7 || nq < 1) return 0;   // one instantiation per (tiles, strips)
    if (c.x_j < 0 || c.x_k < 0 || c.x_c < 0 || c.w_c < c.A) return 0;
    ChainStep a{};
    a.nb = c.nb; a.n = c.n; a.K1 = c.K1; a.A = c.A; a.A2 = c.A2; a.J = c.J;
    a.w_c = c.w_c; a.x_j = c.x_j; a.x_k = c.x_k; a.x_c = c.x_c; a.x_extent = c.x_extent;
    a.E = c.E;
    // k-blocks of phase A are issued in straight-line runs of 25 or of 5 and padded to whole runs (the padded ones
    // meet zero rows of the W image): 25 when that pads at most one k-block more than 5 does
    const int kb = (c.K1 + 3) / 4;
    const int pad25 = (kb + 24) / 25 * 25, pad5 = (kb + 4) / 5 * 5;
    const int unr = pad25 <= pad5 + 1 ? 25 : 5;
    const int KB1 = unr == 25 ? pad25 : pad5, KB2 = nq * 4 + sq;       // k-blocks of the two phases
    // the loader brings A x A2 doubles per slice while phase A runs K1 deep: a short phase A cannot hide it, and
    // there is little T to keep on chip anyway (the two-launch form is then the faster one: measured on C5)
    if (mode == 1 && 2 * c.K1 < c.A) return 0;
    a.AP = 16 * nq + 4 * sq;
    a.A2P = c.A2;
    if (a.AP < 4 * KB2) return 0;
    const int64_t wl = (int64_t)4 * KB1 * a.AP;                          // doubles
    a.ebase = (int)((wl + 1) & ~(int64_t)1);
    const int64_t units = (int64_t)2 * KB2 * a.A2P;                      // 16-byte units of the E image
    a.eunits = (int)cdiv(units, 64) * 64;
    if (a.eunits / 64 > CF_MAX_DMA) return 0;
    // two E images for the small structures (the load of E_{k+1} then has a whole slice to land: their
    // phases are too short to hide it), one for the large ones (no LDS room; their phases are long)
    const int ebuf = nq <= 4 ? 2 : 1;
    const size_t lds = ((size_t)a.ebase + (size_t)a.eunits * 2 * ebuf) * 8;
    if (lds > 160 * 1024) return 0;
#ifdef TTSK_LAB
    { static int dg = [] { const char *e = getenv("TTSK_CF_DIAG"); return e ? atoi(e) : 0; }(); a.diag = dg; }
#endif
    // 32-bit byte offsets: the X walk (incl. the masked prefetch one slice past the end) and T
    if ((c.x_extent + c.x_k + 132 * c.x_c) * 8 >= (1ll << 32) - 64) return 0;
    if ((int64_t)c.A * c.n * c.A2 * 8 >= (1ll << 32) - 64) return 0;
    const bool wt = c.T != nullptr;
    a.t_extent = (int64_t)c.A * c.n * c.J;
    if (wt && (a.t_extent + (int64_t)16 * c.n * c.J) * 8 >= (1ll << 32) - 64) return 0;
    // geometry: one workgroup per CU (the LDS images fill it), each a contiguous range of slices
    const int cus = cf_num_cu();
    int wpp = cus / c.nb > 0 ? cus / c.nb : 1;
    if (wpp > c.n) wpp = c.n;
    // (one slice per workgroup -- a single tensor -- still beats the two-launch form: 313 vs 355 us per C3 sketch)
    a.wpp = wpp;
    a.xcd_map = (wpp % 8 == 0 && wpp >= 8) ? 1 : 0;
    for (int b = 0; b < c.nb; ++b) {
        if ((uintptr_t)c.X[b] & 7) return 0;
        a.W[b] = c.W[b];
        a.X[b] = c.X[b];
        a.T[b] = wt ? c.T[b] : nullptr;
    }
    const int64_t nslab = (int64_t)c.nb * wpp;
    a.slab = (double *)scratch(stream, SCRATCH_GEMM, (size_t)nslab * c.J * c.A2 * 8 + 64);
    if (!a.slab) return TTSK_ERR_HIP;
    const bool prof = prof_on();
    // flops of BOTH products of the step (the pair this kernel replaces), reduce launch inside the bracket
    if (prof) prof_open(st, 2.0 * c.nb * (double)c.n * c.J * ((double)c.K1 * c.A + (double)c.A * c.A2), 6,
                        nq * 100 + sq * 10 + (wt ? 1 : 0), unr == 25, false);
#ifdef TTSK_LAB
    static int stamps_on = [] { const char *e = getenv("TTSK_CF_STAMPS"); return e ? atoi(e) : 0; }();
#else
    constexpr int stamps_on = 0;
#endif
    long long *stamps_dev = nullptr;
    if (stamps_on) {
        if (hipMalloc(&stamps_dev, 8 * 8 * 8 * 8) != hipSuccess) return TTSK_ERR_HIP;
        (void)hipMemset(stamps_dev, 0, 8 * 8 * 8 * 8);
        a.stamps = stamps_dev;
    }
    int rc = launch_chain_step(a, nq, sq, wt, ebuf, unr, lds, (int)nslab, st);
    if (stamps_on) {
        long long h[8 * 8 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, stamps_dev, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(stamps_dev);
        long long t0 = 0;
        for (int i = 0; i < 8 * 8 * 8; ++i) if (h[i] && (!t0 || h[i] < t0)) t0 = h[i];
        fprintf(stderr, "[cf stamps] workgroup 0: cycles since its first stamp; per slice, wave: start | endA | afterB1 | endB | afterB2\n");
        for (int sl = 0; sl < 4; ++sl)
            for (int w = 0; w < 8; ++w) {
                const long long *r = h + (sl * 8 + w) * 8;
                if (!r[1] && !r[4]) continue;
                fprintf(stderr, "  slice %d wave %d: %8lld %8lld %8lld %8lld %8lld\n", sl, w, r[0] ? r[0] - t0 : -1, r[1] - t0, r[2] - t0,
                        r[3] - t0, r[4] - t0);
            }
    }
    if (rc == TTSK_OK) {
        ReduceOut ro{};
        for (int b = 0; b < c.nb; ++b) ro.C[b] = c.Out[b];
        rc = launch_r_reduce(st, a.slab, wpp, c.J, c.A2, 1, (int64_t)c.J, ro, c.nb, (int64_t)c.A2, (int64_t)1, 1.0, 0);
    }
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : (rc == 1 ? 0 : rc);
}

}  // namespace ttsk

using namespace ttsk;

extern "C" int ttsk_chain_step(int nb, int n, int K1, int A, int A2, int J, const double *const *W, int64_t w_c,
                               const double *const *X, int64_t x_j, int64_t x_k, int64_t x_c, int64_t x_extent,
                               const double *E, double *const *T, double *const *Out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(W && X && E && Out && nb >= 1, "ttsk_chain_step: NULL argument");
    ChainStepArgs c{nb, n, K1, A, A2, J, W, w_c, X, x_j, x_k, x_c, x_extent, E, T, Out};
    const int rc = chain_fused_try(c, stream, st, true);
    if (rc == 0) {
        set_error("ttsk_chain_step: shape (n=%d K1=%d A=%d A2=%d J=%d nb=%d) is not covered by the fused kernel", n, K1, A,
                  A2, J, nb);
        return TTSK_ERR_UNSUPPORTED;
    }
    return rc < 0 ? rc : TTSK_OK;
}
