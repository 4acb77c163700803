// Device pieces of the hash sampler shared by sampler.hip and sparse_fused.hip: the 64-bit mix, the flat
// index with the reference's 32-bit running product, the forced-exponent bits and the Cephes ndtri restatement
// (fast_lazy_gaussian.pyx:13-105,183-202).  One source: every kernel that samples produces the same bits.
#pragma once
#include "common.h"

namespace ttsk {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t r)
{
    r += 0x4BE98134A5976FD3ULL;
    r ^= r >> 30;
    r *= 0xBF58476D1CE4E5B9ULL;
    r ^= r >> 27;
    r *= 0x94D049BB133111EBULL;
    r ^= r >> 31;
    return r;
}

constexpr int MAX_MODES = 32;
struct IndexMap {
    int m;
    int row_order[MAX_MODES];
    uint64_t mult[MAX_MODES];  // Fortran-order multipliers with the reference's int32 wrap
    int64_t row_stride;
};

// fast_lazy_gaussian.pyx:60-71 -- `cdef int prod` is 32-bit and sign-extends.
static inline int make_index_map(const uint64_t *shape, int m, int64_t row_stride, const int *row_order,
                          IndexMap *im)
{
    TTSK_ARG(m >= 1 && m <= MAX_MODES, "hash sampler supports 1..%d index rows, got %d", MAX_MODES, m);
    im->m = m;
    im->row_stride = row_stride;
    int32_t prod = (int32_t)shape[0];
    for (int i = 0; i < m; ++i) {
        im->row_order[i] = row_order ? row_order[i] : i;
        if (i == 0) { im->mult[0] = 1; continue; }
        im->mult[i] = (uint64_t)(int64_t)prod;
        prod = (int32_t)((uint64_t)(int64_t)prod * shape[i]);
    }
    return TTSK_OK;
}

__device__ __forceinline__ uint64_t flat_index(const int64_t *idx, const IndexMap &im, size_t e)
{
    uint64_t f = 0;
    for (int i = 0; i < im.m; ++i)
        f += (uint64_t)idx[(int64_t)im.row_order[i] * im.row_stride + (int64_t)e] * im.mult[i];
    return f;
}

__device__ __forceinline__ uint64_t rand_bits(uint64_t flat, int col, uint64_t seed)
{
    uint64_t salt = mix64((uint64_t)col) + seed;
    uint64_t h = mix64(flat + salt);
    return (h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL;  // exponent field 001x..., pyx:91-101
}

// frexp(x)*2-1 for a normal positive double: the 52 mantissa bits as a fraction in [0,1)
__device__ __forceinline__ double mant_unit(uint64_t bits)
{
    return __longlong_as_double((bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL) - 1.0;
}

// ---- Cephes ndtri (fast_lazy_gaussian.pyx:183-202 -> scipy.special.cython_special.ndtri), restated for the vector ALU.
// The operations and their order are those of the C source, so the results are its bits wherever the host libm's log agrees with
// the device's (the 8-ulp bar of DESIGN section 3 is for that).  What is written by hand, because the samplers are bound by
// the vector ALU and each of these is instructions saved without touching a result bit:
//   * a Horner step p * z + c is ONE v_fma_f64 with c read from a scalar register pair (the compiler's form is a v_mov_b32
//     pair per constant -- a 64-bit literal does not exist -- plus v_fmac: 104 moves in 440 instructions of the tail);
//   * a / b is the reciprocal-Newton-residual sequence the compiler emits, without its v_div_scale / v_div_fixup frame: those
//     rescale operands near the ends of the exponent range and patch infinities, zeros and NaNs, none of which occur for the
//     polynomial values and x in [2, 38.6] divided here (same bits otherwise: the frame multiplies by 1);
//   * the far tail (x >= 8: y < exp(-32), one sample in 10^14) is a real branch, not both rational functions and a select.
__device__ __forceinline__ double nd_fma_c(double a, double b, double c)          // a * b + c, c a constant in scalar registers
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

__device__ __forceinline__ double nd_div(double a, double b)                      // a / b, correctly rounded, operands in range (above)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}

// log(x) for a positive normal x, error < 1 ulp: the classical reduction x = 2^k m, m in [sqrt(2)/2, sqrt(2)), f = m - 1,
// s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 minimax R of Sun's fdlibm (e_log.c), the
// two-part ln 2.  The compiler's library log is a double-double evaluation of ~110 instructions; two logarithms per tail
// sample were the largest single item of the samplers' instruction count.  s is formed by reciprocal + two Newton steps
// (its error enters scaled by f^2 / 2).  The host reference evaluates glibc's log (correctly rounded in nearly all cases):
// neither this nor the library log reproduces its last bit, which is what the 8-ulp bar on the normals covers.
__device__ __forceinline__ double nd_log(double x)
{
    const uint64_t bx = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(bx >> 32);
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t up = (hx + 0x95f64u) & 0x100000u;                 // mantissa >= sqrt(2): take m / 2, k + 1
    const double m = __longlong_as_double((long long)(((uint64_t)(hx | (up ^ 0x3ff00000u)) << 32) | (bx & 0xffffffffull)));
    k += (int)(up >> 20);
    const double f = m - 1.0, dk = (double)k;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double s = f * r, z = s * s, w = z * z;
    double t1 = w * 1.531383769920937332e-01 + 2.222219843214978396e-01;       // Lg6, Lg4
    t1 = nd_fma_c(t1, w, 3.999999999940941908e-01) * w;                       // Lg2
    double t2 = w * 1.479819860511658591e-01 + 1.818357216161805012e-01;       // Lg7, Lg5
    t2 = nd_fma_c(t2, w, 2.857142874366239149e-01);                           // Lg3
    t2 = nd_fma_c(t2, w, 6.666666666666735130e-01) * z;                       // Lg1
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return dk * ln2_hi - ((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// the central branch alone (exp(-2) < y0 <= 1 - exp(-2): 73 % of the samples), for callers that have made the case distinction
__device__ __forceinline__ double ndtri_central_dev(double y0)
{
    const double s2pi = 2.50662827463100050242E0;
    const double y = y0 - 0.5;
    const double y2 = y * y;
    double p = -5.99633501014107895267E1;
    p = nd_fma_c(p, y2, 9.80010754185999661536E1);
    p = nd_fma_c(p, y2, -5.66762857469070293439E1);
    p = nd_fma_c(p, y2, 1.39312609387279679503E1);
    p = nd_fma_c(p, y2, -1.23916583867381258016E0);
    double q = y2 + 1.95448858338141759834E0;
    q = nd_fma_c(q, y2, 4.67627912898881538453E0);
    q = nd_fma_c(q, y2, 8.63602421390890590575E1);
    q = nd_fma_c(q, y2, -2.25462687854119370527E2);
    q = nd_fma_c(q, y2, 2.00260212380060660359E2);
    q = nd_fma_c(q, y2, -8.20372256168333339912E1);
    q = nd_fma_c(q, y2, 1.59056225126211695515E1);
    q = nd_fma_c(q, y2, -1.18331621121330003142E0);
    const double x = __builtin_fma(y, nd_div(y2 * p, q), y);
    return x * s2pi;
}

// z p(z) / q(z) for x = 1 / z >= 8: never on a hot path
__device__ __attribute__((noinline)) double ndtri_far_tail_dev(double z)
{
    double p = 3.23774891776946035970E0;
    p = p * z + 6.91522889068984211695E0;
    p = p * z + 3.93881025292474443415E0;
    p = p * z + 1.33303460815807542389E0;
    p = p * z + 2.01485389549179081538E-1;
    p = p * z + 1.23716634817820021358E-2;
    p = p * z + 3.01581553508235416007E-4;
    p = p * z + 2.65806974686737550832E-6;
    p = p * z + 6.23974539184983293730E-9;
    double q = z + 6.02427039364742014255E0;
    q = q * z + 3.67983563856160859403E0;
    q = q * z + 1.37702099489081330271E0;
    q = q * z + 2.16236993594496635890E-1;
    q = q * z + 1.34204006088543189037E-2;
    q = q * z + 3.28014464682127739104E-4;
    q = q * z + 2.89247864745380683936E-6;
    q = q * z + 6.79019408009981274425E-9;
    return z * p / q;
}

// the tails alone: 0 < y0 <= exp(-2) or y0 > 1 - exp(-2)
__device__ __forceinline__ double ndtri_tail_dev(double y0)
{
    const double expm2 = 0.13533528323661269189;
    int code = 1;
    double y = y0;
    if (y > 1.0 - expm2) { y = 1.0 - y; code = 0; }
    double x = sqrt(-2.0 * nd_log(y));
    const double x0 = x - nd_div(nd_log(x), x);
    const double z = nd_div(1.0, x);
    double x1;
    if (x < 8.0) {
        double p = 4.05544892305962419923E0;
        p = nd_fma_c(p, z, 3.15251094599893866154E1);
        p = nd_fma_c(p, z, 5.71628192246421288162E1);
        p = nd_fma_c(p, z, 4.40805073893200834700E1);
        p = nd_fma_c(p, z, 1.46849561928858024014E1);
        p = nd_fma_c(p, z, 2.18663306850790267539E0);
        p = nd_fma_c(p, z, -1.40256079171354495875E-1);
        p = nd_fma_c(p, z, -3.50424626827848203418E-2);
        p = nd_fma_c(p, z, -8.57456785154685413611E-4);
        double q = z + 1.57799883256466749731E1;
        q = nd_fma_c(q, z, 4.53907635128879210584E1);
        q = nd_fma_c(q, z, 4.13172038254672030440E1);
        q = nd_fma_c(q, z, 1.50425385692907503408E1);
        q = nd_fma_c(q, z, 2.50464946208309415979E0);
        q = nd_fma_c(q, z, -1.42182922854787788574E-1);
        q = nd_fma_c(q, z, -3.80806407691578277194E-2);
        q = nd_fma_c(q, z, -9.33259480895457427372E-4);
        x1 = nd_div(z * p, q);
    } else {
        x1 = ndtri_far_tail_dev(z);
    }
    x = x0 - x1;
    return code ? -x : x;
}

__device__ inline double ndtri_dev(double y0)
{
    const double expm2 = 0.13533528323661269189;
    if (y0 == 0.0) return -INFINITY;
    if (y0 == 1.0) return INFINITY;
    if (y0 > expm2 && y0 <= 1.0 - expm2) return ndtri_central_dev(y0);
    return ndtri_tail_dev(y0);
}


}  // namespace ttsk
