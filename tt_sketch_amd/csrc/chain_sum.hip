// Host side of the stacked-terms chain step (chain_sum.h): shape tests, LDS plan, the table that deals the work of
// the second product over the eight waves, launch geometry, slab reduce.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include "chain_sum.h"

namespace ttsk {

static int cs_num_cu()
{
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

// TTSK_CHAIN_SUM: 0 = never, 1 = default
static int cs_mode()
{
    static int m = [] { const char *e = getenv("TTSK_CHAIN_SUM"); return e ? atoi(e) : 1; }();
    return m;
}

// ---- the wave table of phase B -----------------------------------------------------------------------------------
// The (NRT x NNF) full tiles are cut into one or two row bands, each band into rectangles of the bodies the kernel
// instantiates; the strip column (NS 4-wide strips, all row tiles) is a piece of its own or rides on a (1, 1) / (1, 3)
// rectangle.  The pieces (at most 8) are dealt to the wave slots so that the SIMDs -- waves s and s + 4 share SIMD s --
// carry level loads; cost of a piece = its 16x16x4 instructions per k-block (a 4-wide strip tile = 1/4).
struct Piece { int rt0, ct0, rt, ct, srt0, sr; double cost; };

static bool body_ok(int rt, int ct)
{
    static const int ok[][2] = {{1, 1}, {1, 2}, {1, 3}, {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {4, 1}, {5, 1}};
    for (auto &b : ok)
        if (b[0] == rt && b[1] == ct) return true;
    return false;
}

static void bands(int left, int h, std::vector<int> &cur, std::vector<std::vector<int>> &out)
{
    if (left == 0) { out.push_back(cur); return; }
    for (int wdt = 1; wdt <= 3 && wdt <= left; ++wdt)
        if (body_ok(h, wdt)) { cur.push_back(wdt); bands(left - wdt, h, cur, out); cur.pop_back(); }
}

// best assignment of the pieces to 4 SIMDs x 2 slots: returns the largest SIMD load, slot[i] = wave of piece i
static double deal(const std::vector<Piece> &pc, std::vector<int> &slot)
{
    const int np = (int)pc.size();
    std::vector<int> order(np);
    for (int i = 0; i < np; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return pc[x].cost > pc[y].cost; });
    double best = 1e30;
    std::vector<int> cur(np), bst(np);
    double load[4] = {0, 0, 0, 0};
    int cnt[4] = {0, 0, 0, 0};
    // depth-first over SIMD choices, largest pieces first
    struct Rec {
        static void go(int i, const std::vector<Piece> &pc, const std::vector<int> &order, double *load, int *cnt, std::vector<int> &cur,
                       std::vector<int> &bst, double &best)
        {
            const int np = (int)pc.size();
            double mx = 0;
            for (int s = 0; s < 4; ++s) mx = load[s] > mx ? load[s] : mx;
            if (mx >= best) return;
            if (i == np) { best = mx; bst = cur; return; }
            const int p = order[i];
            bool tried_empty = false;
            for (int s = 0; s < 4; ++s) {
                if (cnt[s] >= 2) continue;
                if (cnt[s] == 0) { if (tried_empty) continue; tried_empty = true; }
                cur[p] = s + 4 * cnt[s];
                load[s] += pc[p].cost; cnt[s]++;
                go(i + 1, pc, order, load, cnt, cur, bst, best);
                load[s] -= pc[p].cost; cnt[s]--;
            }
        }
    };
    Rec::go(0, pc, order, load, cnt, cur, bst, best);
    slot = bst;
    return best;
}

static bool wave_table_search(int NRT, int NNF, int NS, ChainSumRole *role);

// (the search costs ~0.1 ms of host time: once per structure)
static bool wave_table(int NRT, int NNF, int NS, ChainSumRole *role)
{
    struct Entry { bool ok; ChainSumRole role[8]; };
    static std::mutex mu;
    static std::map<int, Entry> cache;
    std::lock_guard<std::mutex> lk(mu);
    const int key = (NRT * 64 + NNF) * 8 + NS;
    auto it = cache.find(key);
    if (it == cache.end()) {
        Entry e{};
        e.ok = wave_table_search(NRT, NNF, NS, e.role);
        it = cache.emplace(key, e).first;
    }
    for (int wv = 0; wv < 8; ++wv) role[wv] = it->second.role[wv];
    return it->second.ok;
}

static bool wave_table_search(int NRT, int NNF, int NS, ChainSumRole *role)
{
    if (NRT < 1 || NRT > 10 || (NS && NRT > CS_SRMAX)) return false;
    double best = 1e30;
    int best_reads = 1 << 30;
    std::vector<Piece> best_pc;
    std::vector<int> best_slot;
    for (int h1 = (NRT + 1) / 2; h1 <= NRT && h1 <= 5; ++h1) {
        const int h2 = NRT - h1;
        if (h2 > 5) continue;
        std::vector<std::vector<int>> b1, b2;
        std::vector<int> cur;
        if (NNF) bands(NNF, h1, cur, b1); else b1.push_back({});
        if (h2 && NNF) bands(NNF, h2, cur, b2); else b2.push_back({});
        for (auto &x1 : b1)
            for (auto &x2 : b2) {
                std::vector<Piece> pc;
                int c0 = 0;
                for (int wdt : x1) { pc.push_back({0, c0, h1, wdt, 0, 0, (double)h1 * wdt}); c0 += wdt; }
                c0 = 0;
                for (int wdt : x2) { pc.push_back({h1, c0, h2, wdt, 0, 0, (double)h2 * wdt}); c0 += wdt; }
                // the strip column (all row tiles): a piece of its own
                {
                    std::vector<Piece> q = pc;
                    if (NS) q.push_back({0, 0, 0, 0, 0, NRT, 0.25 * NRT * NS});
                    if (q.empty() || q.size() > 8) continue;
                    std::vector<int> slot;
                    const double mx = deal(q, slot);
                    int reads = 0;
                    for (auto &p : q) reads += p.rt + p.ct + (p.sr ? p.sr + NS : 0);
                    if (mx < best - 1e-9 || (mx < best + 1e-9 && reads < best_reads)) {
                        best = mx; best_reads = reads; best_pc = q; best_slot = slot;
                    }
                }
            }
    }
    if (best_pc.empty()) return false;
    for (int wv = 0; wv < 8; ++wv) { role[wv].body = 0; role[wv].rt0 = role[wv].ct0 = role[wv].pad0 = role[wv].pad1 = 0; }
    for (int i = 0; i < (int)best_pc.size(); ++i) {
        const Piece &p = best_pc[i];
        ChainSumRole &r = role[best_slot[i]];
        r.body = (unsigned char)(p.sr ? 128 + 16 * p.sr + NS : 16 * p.rt + p.ct);
        r.rt0 = (unsigned char)p.rt0; r.ct0 = (unsigned char)p.ct0;
    }
    return true;
}

int chain_sum_try(const ChainSumArgs &cc, int stream, hipStream_t st, bool force)
{
    const ChainStepArgs &c = cc.s;
    if ((!cs_mode() && !force) || c.nb < 1 || c.nb > SK_MAXB) return 0;
    constexpr int JS = 5, KB1 = 5;                      // the instantiated structure: J, K1 <= 20
    if (c.J < 1 || c.J > 4 * JS || c.K1 < 1 || c.K1 > 4 * KB1 || c.A < 4 || c.A > 128 || c.A2 < 4 || c.A2 > 128 || c.n < 1) return 0;
    if ((c.A2 & 1) || ((uintptr_t)c.E & 15)) return 0;                 // 16-byte units of E rows
    if (c.x_j < 0 || c.x_k < 0 || c.x_c < 0 || c.w_c < c.A) return 0;
    if (!force && c.nb < 4) return 0;                   // few terms: the rows of a workgroup would be mostly padding
    ChainSum ka{};
    ChainSumS &a = ka.s;
    a.nb = c.nb; a.n = c.n; a.K1 = c.K1; a.A = c.A; a.A2 = c.A2; a.J = c.J;
    a.w_c = c.w_c; a.x_j = c.x_j; a.x_k = c.x_k; a.x_c = c.x_c; a.x_extent = c.x_extent;
    a.E = c.E;
    a.T = cc.Tint; a.t_b = cc.t_b; a.t_ld = cc.t_ld; a.t_extent = cc.t_extent;
    a.c_fast = c.x_c == 1 ? 1 : 0;
#ifdef TTSK_LAB
    { static int dg = [] { const char *e = getenv("TTSK_CS_DIAG"); return e ? atoi(e) : 0; }(); a.diag = dg; }
#endif
    const int JP = 4 * JS, KP = 4 * KB1;
    const int NAT = (c.A + 15) / 16;
    a.KB2 = (c.A + 3) / 4;
    // columns of Out: full tiles + up to two 4-wide strips (a remainder of 9..15 is a zero-padded full tile)
    {
        const int rem = c.A2 % 16;
        a.NNF = c.A2 / 16;
        a.NS = rem == 0 ? 0 : (rem <= 4 ? 1 : (rem <= 8 ? 2 : 0));
        if (rem > 8) a.NNF += 1;
    }
    a.A2P = std::max(c.A2 + (c.A2 & 1), 16 * a.NNF + 4 * a.NS);
    // terms per workgroup: as many as the LDS, the wave table and the W registers take
    int tpw = 0;
    size_t lds = 0;
    for (int t : {4, 2, 1}) {                           // (at least two waves per term: the G loader's share per lane)
        const int rows = t * JP, NRT = (rows + 15) / 16, RP = 16 * NRT + 2;
        const int64_t tl = (int64_t)4 * a.KB2 * RP;
        const int64_t units = (int64_t)2 * a.KB2 * a.A2P;                // 16-byte units of the E image
        const int64_t eun = cdiv(units, 64) * 64;
        const int64_t el = eun * 2, gl = (int64_t)t * KP * JP;
        const size_t need = (size_t)(tl + el + gl) * 8;
        if (need > 160 * 1024) continue;
        if ((NAT + 8 / t - 1) / (8 / t) > CS_NAMAX) continue;          // a-tiles per wave in phase A
        if (eun / 64 > 8 * CS_DMAMAX) continue;                         // E loader instructions per wave
        ChainSumRole tmp[8];
        if (!wave_table(NRT, a.NNF, a.NS, tmp)) continue;
        tpw = t; lds = need;
        a.RP = RP;
        a.ebase = (int)tl; a.gbase = (int)(tl + el); a.eunits = (int)eun;
        for (int wv = 0; wv < 8; ++wv) ka.role[wv] = tmp[wv];
        break;
    }
    if (!tpw) return 0;
    a.tpw = tpw;
    // phase A: wave w computes the a-tiles [at0, at0 + na) of local term w % tpw
    int na_run = 0;
    {
        const int wpt = 8 / tpw, run = (NAT + wpt - 1) / wpt;
        na_run = run;
        for (int wv = 0; wv < 8; ++wv) {
            const int part = wv / tpw, at0 = part * run;
            ka.role[wv].term = (unsigned char)(wv % tpw);
            ka.role[wv].at0 = (unsigned char)at0;
            ka.role[wv].na = (unsigned char)std::max(0, std::min(run, NAT - at0));
        }
    }
    a.ngroups = (c.nb + tpw - 1) / tpw;
    const int cus = cs_num_cu();
    // Slice ranges per term group = workgroups per group.  A workgroup costs ~20 k cycles before and after its slices
    // (set-up on a cold instruction cache, the switch, the partial results) during which its CU does nothing else -- 157 KB
    // of LDS and 2 x 256 registers per SIMD leave no room for a second one -- so the grid is NOT one workgroup per CU at
    // any price: (i) a workgroup gets at least ~32 k cycles of matrix-pipe time (one and a half times its fixed cost), (ii) a quarter of
    // the CUs is left to the kernel of the other chain, which the sketch drivers always have in flight beside this one.
    // (C5: 24 ranges of 5-6 slices for the right step and 16 of 8 for the left one instead of 32 of 4 each: 0.41 -> 0.39 ms
    // per sketch with two in flight, a single call unchanged.)  TTSK_CS_NR / TTSK_CS_NR_LEFT (lab): the count itself.
    static const int nr_env = [] { const char *e = getenv("TTSK_CS_NR"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    static const int nr_left_env = [] { const char *e = getenv("TTSK_CS_NR_LEFT"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    int nr = cus / a.ngroups;
    {
        const int NRT = (tpw * JP + 15) / 16;
        const double slice_cyc = 64.0 * (2.0 * na_run * (JS / 4 + 0.25 * (JS % 4)) * KB1 + NRT * (a.NNF + 0.25 * a.NS) * a.KB2 / 4.0);
        const int min_slices = (int)(32000.0 / slice_cyc) + 1;
        nr = std::min(std::max(1, 3 * cus / 4 / a.ngroups), std::max(1, c.n / min_slices));
    }
    if (a.T && nr_left_env) nr = nr_left_env;
    else if (!a.T && nr_env) nr = nr_env;
    if (nr >= 8) nr = nr / 8 * 8;
    if (nr < 1) nr = 1;
    if (nr > c.n) nr = c.n;
    a.nranges = nr;
    a.xcd_map = (nr % 8 == 0) ? 1 : 0;
    a.kbase = c.n / nr; a.krem = c.n % nr;
    a.inv_ng = (1 << 20) / a.ngroups + 1;
    if ((int64_t)a.ngroups * nr * a.ngroups >= (1 << 20)) return 0;
    a.wpt = 8 / tpw;
    a.e_inv = (uint32_t)(((1ull << 32) + (uint32_t)a.A2P - 1) / (uint32_t)a.A2P);
    a.per = (KP * JP + a.wpt - 1) / a.wpt;
    a.gu = (a.per + 63) / 64;
    // 32-bit byte offsets
    const int64_t lim32 = (1ll << 32) - 64;
    if ((c.x_extent + c.x_k) * 8 >= lim32) return 0;
    if ((int64_t)c.A * c.n * c.A2 * 8 >= lim32) return 0;
    if (a.T && a.t_extent * 8 >= lim32) return 0;
    if (((int64_t)(c.K1 - 1) * c.w_c + c.A) * 8 >= lim32) return 0;
    if ((c.J > 1 && c.x_j * 8 >= lim32) || (c.K1 > 1 && c.x_c * 8 >= lim32)) return 0;
    a.w_c8 = (uint32_t)(c.w_c * 8);
    a.x_j8 = c.J > 1 ? (uint32_t)(c.x_j * 8) : 0u;
    a.x_c8 = c.K1 > 1 ? (uint32_t)(c.x_c * 8) : 0u;
    if (a.T) {
        if ((c.nb > 1 && cc.t_b * 8 >= lim32) || (c.A > 1 && (int64_t)c.n * cc.t_ld * 8 >= lim32)) return 0;
        a.t_b8 = c.nb > 1 ? (uint32_t)(cc.t_b * 8) : 0u;
        a.t_a8 = c.A > 1 ? (uint32_t)((int64_t)c.n * cc.t_ld * 8) : 0u;
    }
    if ((int64_t)c.nb * nr * c.J * c.A2 * 8 >= lim32) return 0;
    a.slab_r8 = (uint32_t)((int64_t)c.J * c.A2 * 8);
    a.slab_t8 = (uint32_t)((int64_t)nr * c.J * c.A2 * 8);
    for (int b = 0; b < c.nb; ++b) {
        if ((uintptr_t)c.X[b] & 7) return 0;
        ka.W[b] = c.W[b];
        ka.X[b] = c.X[b];
    }
    const int64_t nslab = (int64_t)c.nb * nr;
    a.slab = (double *)scratch(stream, SCRATCH_GEMM, (size_t)nslab * c.J * c.A2 * 8 + 64);
    if (!a.slab) return TTSK_ERR_HIP;
    const bool prof = prof_on();
    if (prof) prof_open_named(st, -2, 2.0 * c.nb * (double)c.n * c.J * ((double)c.K1 * c.A + (double)c.A * c.A2),
                              a.T ? "chain_sum_kernel<5, 5, NA, true>" : "chain_sum_kernel<5, 5, NA, false>");
#ifdef TTSK_LAB
    static int stamps_on = [] { const char *e = getenv("TTSK_CS_STAMPS"); return e ? atoi(e) : 0; }();
    long long *stamps_dev = nullptr;
    if (stamps_on) {
        if (hipMalloc(&stamps_dev, 8 * 8 * 8 * 8) != hipSuccess) return TTSK_ERR_HIP;
        (void)hipMemset(stamps_dev, 0, 8 * 8 * 8 * 8);
        a.stamps = stamps_dev;
    }
#endif
    const int grid = a.ngroups * nr;
    int rc = na_run <= 2 ? launch_chain_sum_2(ka, a.T != nullptr, lds, grid, st) : launch_chain_sum_4(ka, a.T != nullptr, lds, grid, st);
    if (rc != TTSK_OK) set_error("chain_sum_kernel launch failed");
#ifdef TTSK_LAB
    if (stamps_on) {
        long long h[8 * 8 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, stamps_dev, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(stamps_dev);
        long long t0 = 0;
        for (int i = 0; i < 8 * 8 * 8; ++i) if (h[i] && (!t0 || h[i] < t0)) t0 = h[i];
        for (int w = 0; w < 8; ++w)
            fprintf(stderr, "[cs stamps] wave %d: kernel entry -> first slice %lld cycles, last barrier -> behind the stores %lld cycles\n", w,
                    h[w * 8 + 0] - h[w * 8 + 6], h[w * 8 + 7] - h[(3 * 8 + w) * 8 + 5]);
        for (int w = 0; w < 8; ++w) {
            const long long *r = h + (4 * 8 + w) * 8, e = h[w * 8 + 6];
            fprintf(stderr, "[cs stamps] prologue wave %d: picked %lld | G issued %lld | W issued %lld | E offsets %lld | lane offsets %lld | body entered %lld | G store begins %lld | G stored %lld | barrier passed %lld | first slice %lld\n", w,
                    r[6] - e, r[0] - e, r[1] - e, r[2] - e, r[3] - e, h[(5 * 8 + w) * 8] - e, r[4] - e, r[5] - e, r[7] - e, h[w * 8 + 0] - e);
        }
        fprintf(stderr, "[cs stamps] workgroup 0, cycles since its first stamp; slice, wave (body): start | endA | dma landed | afterB1 | endB | afterB2\n");
        for (int sl = 0; sl < 4; ++sl)
            for (int w = 0; w < 8; ++w) {
                const long long *r = h + (sl * 8 + w) * 8;
                fprintf(stderr, "  slice %d wave %d (%02x): %8lld %8lld %8lld %8lld %8lld %8lld\n", sl, w, ka.role[w].body, r[0] - t0, r[1] - t0, r[2] - t0,
                        r[3] - t0, r[4] - t0, r[5] - t0);
            }
    }
#endif
    if (rc == TTSK_OK) {
        ReduceOut ro{};
        for (int b = 0; b < c.nb; ++b) ro.C[b] = c.Out[b];
        rc = launch_r_reduce(st, a.slab, nr, c.J, c.A2, 1, (int64_t)c.J, ro, c.nb, (int64_t)c.A2, (int64_t)1, 1.0, 0);
    }
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : rc;
}

}  // namespace ttsk

using namespace ttsk;

// One chain step of nb low-rank tensor trains through the stacked-terms kernel (tests, tools).  T (optional): the
// intermediate T[b * t_b + (a * n + k) * t_ld + j], t_extent elements addressable.
extern "C" int ttsk_chain_step_sum(int nb, int n, int K1, int A, int A2, int J, const double *const *W, int64_t w_c,
                                   const double *const *X, int64_t x_j, int64_t x_k, int64_t x_c, int64_t x_extent,
                                   const double *E, double *T, int64_t t_b, int64_t t_ld, int64_t t_extent,
                                   double *const *Out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(W && X && E && Out && nb >= 1, "ttsk_chain_step_sum: NULL argument");
    ChainSumArgs c{{nb, n, K1, A, A2, J, W, w_c, X, x_j, x_k, x_c, x_extent, E, nullptr, Out}, T, t_b, t_ld, t_extent};
    const int rc = chain_sum_try(c, stream, st, true);
    if (rc == 0) {
        set_error("ttsk_chain_step_sum: shape (n=%d K1=%d A=%d A2=%d J=%d nb=%d) is not covered by the stacked-terms kernel", n,
                  K1, A, A2, J, nb);
        return TTSK_ERR_UNSUPPORTED;
    }
    return rc < 0 ? rc : TTSK_OK;
}
