// Host side of stream_small_kernel (stream_small.h) + its instantiations.
#include <cstdlib>
#include "stream_small.h"

namespace ttsk {

template <int NF, int STR, int UNR>
static int launch_ss_one(const StreamSmall &a, size_t lds, int grid, hipStream_t st)
{
    auto kern = stream_small_kernel<NF, STR, 5, UNR>;
    static PerInit attr_done;
    if (attr_done.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

static int launch_ss(const StreamSmall &a, int nf, int str, int unr, size_t lds, int grid, hipStream_t st)
{
#define TTSK_SS_CASE(NF, STR) if (nf == NF && str == STR) return unr == 25 ? launch_ss_one<NF, STR, 25>(a, lds, grid, st) : launch_ss_one<NF, STR, 5>(a, lds, grid, st);
    TTSK_SS_CASE(1, 0) TTSK_SS_CASE(1, 1) TTSK_SS_CASE(1, 2) TTSK_SS_CASE(2, 0) TTSK_SS_CASE(2, 1) TTSK_SS_CASE(2, 2)
    TTSK_SS_CASE(3, 0) TTSK_SS_CASE(3, 1) TTSK_SS_CASE(3, 2) TTSK_SS_CASE(4, 0) TTSK_SS_CASE(4, 1) TTSK_SS_CASE(4, 2)
    TTSK_SS_CASE(5, 0) TTSK_SS_CASE(5, 1) TTSK_SS_CASE(5, 2) TTSK_SS_CASE(6, 0) TTSK_SS_CASE(6, 1) TTSK_SS_CASE(6, 2)
    TTSK_SS_CASE(7, 0)
#undef TTSK_SS_CASE
    return 1;
}

static int ss_num_cu()
{
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

int stream_small_try(const StreamSmallArgs &c, int stream, hipStream_t st)
{
    (void)stream;
    static int on = [] { const char *e = getenv("TTSK_STREAM_SMALL"); return e ? atoi(e) : 1; }();
    if (!on || c.nb < 1 || c.nb > SK_MAXB) return 0;
    if (c.K1 < 1 || c.A < 4 || c.J < 1024) return 0;
    if (c.s_j < c.K1 || c.w_c < c.A || c.c_j < c.A) return 0;
    const int kb = (c.K1 + 3) / 4;
    const int pad25 = (kb + 24) / 25 * 25, pad5 = (kb + 4) / 5 * 5;
    const int unr = pad25 <= pad5 + 1 ? 25 : 5;         // straight-line runs of k-blocks (padded to whole runs)
    const int KB1 = unr == 25 ? pad25 : pad5;
    // column chunks: the fewest whose W image fits the LDS and whose structure (<= 7 tiles) is instantiated
    int nac = 0, nf = 0, str = 0, need = 0;
    size_t lds = 0;
    for (int t = 1; t <= 16 && !nac; ++t) {
        need = (int)((cdiv(c.A, t) + 3) / 4 * 4);
        nf = need / 16;
        const int rem = need % 16;
        if (rem == 0) str = 0;
        else if (rem <= 4) str = 1;
        else if (rem <= 8) str = 2;
        else { nf += 1; str = 0; }
        if (nf < 1 || nf + (str ? 1 : 0) > 7) continue;
        lds = (size_t)4 * (KB1 + 1) * (16 * nf + 4 * str) * 8;
        if (lds <= 160 * 1024) nac = t;
    }
    if (!nac) return 0;
    StreamSmall a{};
    a.nb = c.nb; a.J = c.J; a.K1 = c.K1; a.A = c.A;
    a.nac = nac; a.ac = nac == 1 ? c.A : need;
    a.s_j = c.s_j; a.w_c = c.w_c; a.c_j = c.c_j;
    a.accumulate = c.accumulate;
    a.AP = 16 * nf + 4 * str;
    const int ntiles = (c.J + 15) / 16;
    int wpp = ss_num_cu() / (c.nb * nac) > 0 ? ss_num_cu() / (c.nb * nac) : 1;
    if (wpp > (ntiles + 7) / 8) wpp = (ntiles + 7) / 8;
    a.wpp = wpp;
    a.s_extent = (int64_t)(c.J - 1) * c.s_j + c.K1;
    a.c_extent = (int64_t)(c.J - 1) * c.c_j + c.A;
    // 32-bit byte offsets incl. the look-ahead one visit past the last tile
    if (((int64_t)(ntiles + wpp * 8 + 2) * 16 * c.s_j + 256) * 8 >= (1ll << 32) - 64) return 0;
    if (((int64_t)(ntiles + 1) * 16 * c.c_j + 256) * 8 >= (1ll << 32) - 64) return 0;
    for (int b = 0; b < c.nb; ++b) {
        if (((uintptr_t)c.S[b] | (uintptr_t)c.W[b] | (uintptr_t)c.C[b]) & 7) return 0;
        a.S[b] = c.S[b]; a.W[b] = c.W[b]; a.C[b] = c.C[b];
    }
    const bool prof = prof_on();
    if (prof) prof_open(st, 2.0 * c.nb * (double)c.J * c.K1 * c.A, 7, nf * 10 + str, unr == 25, false);
    const int rc = launch_ss(a, nf, str, unr, lds, c.nb * wpp * nac, st);
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : (rc == 1 ? 0 : rc);
}

template <int NF, int STR, int UNR>
static int launch_sss_one(const StreamSmallSum &a, size_t lds, int grid, hipStream_t st)
{
    auto kern = stream_small_sum_kernel<NF, STR, 5, UNR>;
    static PerInit attr_done;
    if (attr_done.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

static int launch_sss(const StreamSmallSum &a, int nf, int str, int unr, size_t lds, int grid, hipStream_t st)
{
#define TTSK_SSS_CASE(NF, STR) if (nf == NF && str == STR) return unr == 25 ? launch_sss_one<NF, STR, 25>(a, lds, grid, st) : launch_sss_one<NF, STR, 5>(a, lds, grid, st);
    TTSK_SSS_CASE(1, 0) TTSK_SSS_CASE(1, 1) TTSK_SSS_CASE(1, 2) TTSK_SSS_CASE(2, 0) TTSK_SSS_CASE(2, 1) TTSK_SSS_CASE(2, 2)
    TTSK_SSS_CASE(3, 0) TTSK_SSS_CASE(3, 1) TTSK_SSS_CASE(3, 2)
#undef TTSK_SSS_CASE
    return 1;
}

int stream_small_sum_try(const StreamSmallSumArgs &c, int stream, hipStream_t st)
{
    (void)stream;
    static int on = [] { const char *e = getenv("TTSK_STREAM_SMALL_SUM"); return e ? atoi(e) : 1; }();
    if (!on || c.nb < 2 || c.K1 < 1 || c.A < 8 || c.J < 1024) return 0;
    if (c.s_j < c.K1 || c.w_c < c.A || c.c_j < c.A || c.s_b < 0 || c.w_b < 0) return 0;
    if (((uintptr_t)c.S | (uintptr_t)c.W | (uintptr_t)c.C) & 7) return 0;
    const int kb = (c.K1 + 3) / 4;
    const int pad25 = (kb + 24) / 25 * 25, pad5 = (kb + 4) / 5 * 5;
    const int unr = pad25 <= pad5 + 1 ? 25 : 5;
    const int KB1 = unr == 25 ? pad25 : pad5;
    const int ntiles = (c.J + 15) / 16, groups = (ntiles + 7) / 8, cus = ss_num_cu();
    // column chunks: at least two (two W images of <= 6144 doubles), as many as fill the chip in one round
    int nac = 0, nf = 0, str = 0, need = 0;
    auto plan = [&](int t, int &nf_, int &str_, int &need_) {
        need_ = (int)((cdiv(c.A, t) + 3) / 4 * 4);
        nf_ = need_ / 16;
        const int rem = need_ % 16;
        if (rem == 0) str_ = 0;
        else if (rem <= 4) str_ = 1;
        else if (rem <= 8) str_ = 2;
        else { nf_ += 1; str_ = 0; }
        if (nf_ < 1 || nf_ + (str_ ? 1 : 0) > 4 || (nf_ == 3 && str_ > 2) || nf_ > 3) return false;
        return (size_t)4 * (KB1 + 1) * (16 * nf_ + 4 * str_) <= 6144;
    };
    for (int t = 2; t <= 16 && !nac; ++t)
        if (plan(t, nf, str, need)) nac = t;
    if (!nac) return 0;
    while (groups * (nac + 1) <= cus + cus / 8 && (c.A + nac) / (nac + 1) >= 16) {
        int nf2, str2, need2;
        if (!plan(nac + 1, nf2, str2, need2)) break;
        ++nac; nf = nf2; str = str2; need = need2;
    }
    StreamSmallSum a{};
    a.S = c.S; a.W = c.W; a.C = c.C;
    a.nb = c.nb; a.groups = groups; a.nac = nac; a.ac = need;
    a.J = c.J; a.K1 = c.K1; a.A = c.A;
    a.s_j = c.s_j; a.w_c = c.w_c; a.c_j = c.c_j; a.s_b = c.s_b; a.w_b = c.w_b;
    a.accumulate = c.accumulate;
    a.AP = 16 * nf + 4 * str;
    a.s_extent = (int64_t)(c.nb - 1) * c.s_b + (int64_t)(c.J - 1) * c.s_j + c.K1;
    a.w_extent = (int64_t)(c.nb - 1) * c.w_b + (int64_t)(c.K1 - 1) * c.w_c + c.A;
    a.c_extent = (int64_t)(c.J - 1) * c.c_j + c.A;
    // 32-bit byte offsets incl. the prefetch into the term behind the last one
    if (((int64_t)c.nb * c.s_b + (int64_t)(ntiles + 1) * 16 * c.s_j + 4 * KB1 + 256) * 8 >= (1ll << 32) - 64) return 0;
    if (((int64_t)c.nb * c.w_b + (int64_t)c.K1 * c.w_c + 256) * 8 >= (1ll << 32) - 64) return 0;
    if (((int64_t)(ntiles + 1) * 16 * c.c_j + 256) * 8 >= (1ll << 32) - 64) return 0;
    const int units = groups * nac;
    a.xcd_map = units % 8 == 0 ? 1 : 0;
    const size_t lds = (size_t)2 * 4 * (KB1 + 1) * a.AP * 8;
    const bool prof = prof_on();
    if (prof) {
        char name[96];
        snprintf(name, sizeof(name), "stream_small_sum_kernel<%d, %d, 5, %d>", nf, str, unr);
        prof_open_named(st, -2, 2.0 * c.nb * (double)c.J * c.K1 * c.A, name);
    }
    const int rc = launch_sss(a, nf, str, unr, lds, units, st);
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : (rc == 1 ? 0 : rc);
}

}  // namespace ttsk
