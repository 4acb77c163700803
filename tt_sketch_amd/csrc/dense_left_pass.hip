// Dense tensor x DRM MATRICES (DenseGaussianDRM, plug-ins): the first four left products from ONE read of the tensor.
//
//   reference: dense_gaussian_drm.py:77-80 hands out the matrices A_mu (l x n_0 ... n_mu); dense_sketch.py:7-52 forms
//   Z_mu = A_mu X^{<mu+1>} inside Omega_mu and Psi_{mu+1} -- each of them a pass over the tensor (8.59 GB at C2; the generic
//   path reads it once per mu).
//
// X is viewed as (n0, n1, n2, C) with C = n3 * n4 (n4 = everything behind the fourth mode).  A workgroup owns ONE value of
// i2 and a tile of 512 columns c = (i3, i4) and walks (i1, i0) -- i1 outside, i0 inside, four values of i0 per k-block:
//
//   Z_0[a, (i1, i2, c)] = sum_{i0}      A_0[a, i0]              X[i0, i1, i2, c]    complete after every i1: stored, reset
//   Z_1[a, (i2, c)]     = sum_{i0, i1}  A_1[a, (i0, i1)]        X[...]              complete at the end of the walk
//   Z_2[a, c]          += sum_{i0, i1}  A_2[a, (i0, i1, i2)]    X[...]              partial over i2: slab per workgroup
//   E_3[a, c]          += sum_{i0, i1}  A_3[a, (i0, i1, i2, i3(c))] X[...]          partial over i2: slab; Z_3[a, i4] = sum_{i3} E_3
//
// -- four products with the SAME right-hand operand (the X tile, K = i0) and four different small left-hand ones.  Eight
// waves x 64 columns each x four accumulator sets of (16 + 4) rows: 160 registers of accumulators per lane.  The X tile
// (4 rows x 512 columns, 16 KB) comes by global_load_lds, two images; the A slices of a k-block (A_0, A_1, A_2: 20 x 4
// each; A_3: 20 x 4 x (i3 values of the tile)) are gathered by the whole workgroup, one or two 8-byte loads per thread,
// through registers into two LDS images.  A_1 and A_2 are read from TRANSPOSED copies (i0 fastest: the k-block's four
// values contiguous; 0.6 / 42 MB at C2, made per sketch); A_3 (2.7 GB) is read where it lies, in 64-byte pieces that the
// neighbouring column tile -- same XCD, same moment -- completes to whole lines.
// Work: 4 x 2 l n0 n1 n2 C flops = 172 GF at C2 (1024 k-blocks of 2560 matrix-pipe cycles per SIMD and workgroup, two
// workgroups per CU in turn), bytes: X once + A_3 once + Z_0 written = 14 GB.
#include <type_traits>
#include "common.h"
#include "skinny.h"

namespace ttsk {

namespace {

constexpr int LP_CT = 512;       // columns per workgroup
constexpr int LP_XP = 528;       // row pitch of the X image (16 mod 32: the two k rows a half wave reads hit different banks)
constexpr int LP_MAXI3 = 9;      // i3 values a column tile may span: (3 + that) x 80 gathered A elements <= 1024 = two per thread

struct LeftPass {
    const double *X;
    const double *A0, *A1t, *A2t, *A3p;
    double *Z0, *Z1, *slab2, *slab3;
    int n0, n1, n2, n4, l;
    int64_t C;                   // n3 * n4
    int ni3;                     // i3 values per column tile = 512 / n4 (>= 1)
    int nct;                     // column tiles = C / 512
    int xcd_map;
};

__device__ __forceinline__ double lp_mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// STRIP: l > 16 (rows 16..19 as a 4-row strip)
template <bool STRIP>
__global__ __launch_bounds__(512, 2) void dense_left_pass_kernel(LeftPass a)
{
    extern __shared__ double lp_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x16 = lane & 15, kq = lane >> 4;
    int i2, ct;
    if (a.xcd_map) {
        // blocks b and b + 8 share an XCD: the column tiles of one i2 on the same one (they complete each other's A_3 lines)
        const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
        ct = y % a.nct;
        i2 = x + 8 * (y / a.nct);
    } else {
        ct = blockIdx.x % a.nct;
        i2 = blockIdx.x / a.nct;
    }
    const int n0 = a.n0, n1 = a.n1, n2 = a.n2, n4 = a.n4, l = a.l, ni3 = a.ni3;
    const int64_t C = a.C;
    const int nsets = 3 + ni3;                       // A images per k-block
    const int aimg = (nsets * 80 + 127) / 128 * 128 + 128;   // doubles per A buffer: [set][a (20)][k (4)], whole instructions + one of padding
    double *XB = lp_lds;                             // [4][4][LP_XP]
    double *AB = lp_lds + 4 * 4 * LP_XP;             // [4][nsets][20][4]
    const int kpb = n0 >> 2;                         // k-blocks per i1
    const int nkb = n1 * kpb;

    // ---- the A images by LDS-DMA as well (a register-staged gather makes the compiler wait for every load in flight in front
    // of each LDS-DMA instruction -- measured: the prefetch pipeline collapses to one k-block).  All four arrays are read with
    // i0 fastest (A_0 as it is, the transposed copies of A_1, A_2 and A_3), so a k-block's four values are 32 contiguous bytes:
    // 16-byte unit (set, a, kp) = elements k = 2 kp, 2 kp + 1 of row a lands at AB[set][a][k] -- lane-linear in the unit
    // number, 64 units per instruction, wave m issues instruction m (at most 7 of them: (3 + 8 sets) x 40 units).
    const int i3_0 = (int)(((int64_t)ct * LP_CT) / n4);
    const int64_t n3 = C / n4;
    const int nunits = nsets * 40, ninstr = (nunits + 63) >> 6;
    const int am = w < ninstr ? w : ninstr - 1;      // (a wave without an instruction repeats the last one: the waits count three per wave)
    const char *asrc;                                // this lane's unit of the k-block (0, 0)
    int akind;                                       // how it moves with (i1, i0): 0: i0, 1: i1 n0 + i0, 2: i1 n3 n0 + i0
    {
        int U = 64 * am + lane;
        if (U >= nunits) U = nunits - 1;             // (the tail lanes of the last instruction repeat its last unit; their slots are padding)
        const int set = U / 40, aa = (U >> 1) % 20, kp = U & 1;
        const int ar = aa < l ? aa : l - 1;          // rows beyond l: any finite values (their results are never stored)
        const double *p;
        if (set == 0) { p = a.A0 + (int64_t)ar * n0; akind = 0; }
        else if (set == 1) { p = a.A1t + (int64_t)ar * n1 * n0; akind = 1; }
        else if (set == 2) { p = a.A2t + ((int64_t)ar * n2 + i2) * n1 * n0; akind = 1; }
        else { p = a.A3p + ((((int64_t)ar * n2 + i2) * n1) * n3 + (i3_0 + set - 3)) * n0; akind = 2; }
        asrc = (const char *)(p + 2 * kp);
    }
    auto a_fill = [&](int i1, int i0, int buf) {
        const int64_t o0 = i0, o1 = (int64_t)i1 * n0 + i0, o2 = (int64_t)i1 * n3 * n0 + i0;
        const int64_t o = akind == 0 ? o0 : (akind == 1 ? o1 : o2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(asrc + o * 8),
                                         (__attribute__((address_space(3))) void *)(AB + buf * aimg + am * 128), 16, 0, 0);
    };
    // ---- the X tile: 16 instructions of 1 KB per k-block, wave w issues (row r, segment seg) = (2 w + j) / 4, % 4
    const int64_t rowstep = (int64_t)n1 * n2 * C;    // elements between consecutive i0
    const char *xbase = (const char *)(a.X + (int64_t)i2 * C + (int64_t)ct * LP_CT);
    uint32_t xvoff[2];
    int xdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = 2 * w + j, r = idx >> 2, seg = idx & 3;
        xvoff[j] = (uint32_t)(((int64_t)r * rowstep + seg * 128 + 2 * lane) * 8);       // (3 rowstep < 2^29 elements: checked by the host)
        xdst[j] = r * LP_XP + seg * 128;
    }
    auto x_fill = [&](int i1, int i0, int buf) {
        const char *src = xbase + uniform_i64(((int64_t)i0 * rowstep + (int64_t)i1 * n2 * C) * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + xvoff[j]),
                                             (__attribute__((address_space(3))) void *)(XB + buf * 4 * LP_XP + xdst[j]), 16, 0, 0);
    };

    // ---- accumulators: set s (0: Z_0, 1: Z_1, 2: Z_2, 3: E_3) x column tile t of this wave's 64 columns
    v4d acc[4][4];
    double accs[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][t][r] = 0.0;
            accs[s][t] = 0.0;
        }
    // the A_3 image of this wave's columns: their i3 relative to the tile's first
    const int j3 = (64 * w) / n4;                    // (n4 >= 64: the wave's 64 columns share one i3)
    const int a_lane = 4 * x16 + kq;                 // tile fragment: lane (a = x16, k = kq)
    const int a_strip = 4 * (16 + (x16 & 3)) + kq;   // strip fragment: rows 16 .. 19 replicated over the lane blocks
    const int b_lane = kq * LP_XP + 64 * w + x16;    // + 16 t

    // Z_0 of one i1: register r of tile t is row a = 4 r + kq, column 64 w + 16 t + x16; the strip: row 16 + kq
    const int64_t z0_row = (int64_t)n1 * n2 * C;
    auto z0_flush = [&](int i1) {
        double *dst = a.Z0 + ((int64_t)i1 * n2 + i2) * C + (int64_t)ct * LP_CT + 64 * w + x16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int aa = 4 * r + kq;
                if (aa < l) dst[(int64_t)aa * z0_row + 16 * t] = acc[0][t][r];
                acc[0][t][r] = 0.0;
            }
            if constexpr (STRIP) {
                if (16 + kq < l) dst[(int64_t)(16 + kq) * z0_row + 16 * t] = accs[0][t];
                accs[0][t] = 0.0;
            }
        }
    };

    // ---- the walk: X tile and A images of k-block kb + 2 requested while kb is worked on (four X images, four A images; the
    // loop is unrolled by four so that the images are compile-time); per k-block and wave 3 LDS-DMA instructions, counted by vmcnt
    int li1 = 0, li0 = 0;                            // (i1, i0) of the next k-block to request
    auto advance = [&]() { li0 += 4; if (li0 >= n0) { li0 = 0; ++li1; } };
    x_fill(li1, li0, 0); a_fill(li1, li0, 0); advance();
    if (nkb > 1) { x_fill(li1, li0, 1); a_fill(li1, li0, 1); advance(); }
    auto step = [&](int kb, auto jc) {
        constexpr int J = decltype(jc)::value;       // kb % 4: X image J, A image J
        // k-block kb has landed: everything but the three requests of kb + 1 (the stores of a Z_0 flush are younger still and
        // are waited for with them, once per i1)
        if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kb + 2 < nkb) {
            x_fill(li1, li0, (J + 2) & 3);           // the images k-block kb - 2 was read from
            a_fill(li1, li0, (J + 2) & 3);
            advance();
        }
        {
            const double *Ab = AB + J * aimg, *Xb = XB + J * 4 * LP_XP;
            // fragments of the k-block: A sets 0 .. 2 (+ strips), then per column tile the X fragment and its A_3 set
            double af[3], as[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                af[s] = Ab[s * 80 + a_lane];
                if constexpr (STRIP) as[s] = Ab[s * 80 + a_strip];
            }
            double bf[4], a3s = 0.0;
            const double a3f = Ab[(3 + j3) * 80 + a_lane];
            if constexpr (STRIP) a3s = Ab[(3 + j3) * 80 + a_strip];
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = Xb[b_lane + 16 * t];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    acc[s][t] = mfma16(af[s], bf[t], acc[s][t]);
                    if constexpr (STRIP) accs[s][t] = lp_mfma4(as[s], bf[t], accs[s][t]);
                }
                acc[3][t] = mfma16(a3f, bf[t], acc[3][t]);
                if constexpr (STRIP) accs[3][t] = lp_mfma4(a3s, bf[t], accs[3][t]);
            }
        }
        if ((kb + 1) % kpb == 0) z0_flush(kb / kpb);
    };
    for (int kb = 0; kb < nkb; kb += 4) {
        step(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < nkb) step(kb + 1, std::integral_constant<int, 1>{});
        if (kb + 2 < nkb) step(kb + 2, std::integral_constant<int, 2>{});
        if (kb + 3 < nkb) step(kb + 3, std::integral_constant<int, 3>{});
    }

    // ---- Z_1 (complete), the partial Z_2 and E_3 of this i2
    {
        double *z1 = a.Z1 + (int64_t)i2 * C + (int64_t)ct * LP_CT + 64 * w + x16;
        double *s2 = a.slab2 + (int64_t)i2 * l * C + (int64_t)ct * LP_CT + 64 * w + x16;
        double *s3p = a.slab3 + (int64_t)i2 * l * C + (int64_t)ct * LP_CT + 64 * w + x16;
        const int64_t z1_row = (int64_t)n2 * C;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int aa = 4 * r + kq;
                if (aa < l) {
                    z1[(int64_t)aa * z1_row + 16 * t] = acc[1][t][r];
                    s2[(int64_t)aa * C + 16 * t] = acc[2][t][r];
                    s3p[(int64_t)aa * C + 16 * t] = acc[3][t][r];
                }
            }
            if constexpr (STRIP) {
                const int aa = 16 + kq;
                if (aa < l) {
                    z1[(int64_t)aa * z1_row + 16 * t] = accs[1][t];
                    s2[(int64_t)aa * C + 16 * t] = accs[2][t];
                    s3p[(int64_t)aa * C + 16 * t] = accs[3][t];
                }
            }
        }
    }
}

}  // namespace

}  // namespace ttsk

using namespace ttsk;

// Z_0 .. Z_3 = A_mu X^{<mu+1>} (dense_sketch.py:15-16, :40-51 with the matrices of dense_gaussian_drm.py:77-80) from one read
// of X (n0, n1, n2, n3 * n4 = C columns; n4 = the extent Z_3 keeps).  A1t (l, n1, n0) and A2t (l, n2, n1, n0) are the
// transposed copies of A_1 (l, n0 n1) and A_2 (l, n0 n1 n2), A3p (l, n2, n1, n3, n0) that of A_3 (l, n0 n1 n2 n3) -- i0 fastest
// in all of them; A0 (l, n0) as the DRM holds it.
// Z0 (l, n1 n2 C), Z1 (l, n2 C), Z2 (l, C), E3 (l, C): E3[a, (i3, i4)] still holds i3 -- Z_3 = its sum over i3 (the caller's,
// a product of l C flops).  TTSK_ERR_UNSUPPORTED outside the cover: n0 % 4 == 0, C % 512 == 0, n4 in {64, 128, 256, 512}
// (a column tile of 512 spans at most eight values of i3), l <= 20.
extern "C" int ttsk_dense_left_pass(const double *X, int64_t n0, int64_t n1, int64_t n2, int64_t C, int64_t n4, int l,
                                    const double *A0, const double *A1t, const double *A2t, const double *A3p, double *Z0,
                                    double *Z1, double *Z2, double *E3, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(X && A0 && A1t && A2t && A3p && Z0 && Z1 && Z2 && E3, "ttsk_dense_left_pass: NULL argument");
    TTSK_ARG(n0 >= 1 && n1 >= 1 && n2 >= 1 && C >= 1 && n4 >= 1 && l >= 1, "ttsk_dense_left_pass: bad shape");
    const bool ok = n0 % 4 == 0 && C % LP_CT == 0 && C % n4 == 0 && n4 % 16 == 0 && LP_CT % n4 == 0 && LP_CT / n4 <= LP_MAXI3 &&
                    l <= 20 && n0 <= (1 << 20) && n1 <= (1 << 20) && n2 <= (1 << 16) && n2 * (C / LP_CT) < (1ll << 30) &&
                    (double)l * n0 * n1 * n2 * (C / n4) * 8 < 4.0e9 && (double)4 * n1 * n2 * C * 8 < 4.0e9;      // 32-bit lane offsets
    if (!ok) {
        set_error("ttsk_dense_left_pass: shape (%lld, %lld, %lld, C = %lld, n4 = %lld), l = %d is outside the kernel's cover",
                  (long long)n0, (long long)n1, (long long)n2, (long long)C, (long long)n4, l);
        return TTSK_ERR_UNSUPPORTED;
    }
    LeftPass a{};
    a.X = X; a.A0 = A0; a.A1t = A1t; a.A2t = A2t; a.A3p = A3p;
    a.Z0 = Z0; a.Z1 = Z1;
    a.n0 = (int)n0; a.n1 = (int)n1; a.n2 = (int)n2; a.n4 = (int)n4; a.l = l;
    a.C = C;
    a.ni3 = (int)(LP_CT / n4);
    a.nct = (int)(C / LP_CT);
    a.xcd_map = (n2 % 8 == 0) ? 1 : 0;
    // the partial Z_2 / E_3 of the n2 workgroup rows: [i2][l][C] each, summed below
    const size_t slab = (size_t)n2 * l * C;
    double *ws = (double *)scratch(stream, SCRATCH_GEMM, 2 * slab * 8 + 64);
    if (!ws) return TTSK_ERR_HIP;
    a.slab2 = ws;
    a.slab3 = ws + slab;
    const size_t lds = ((size_t)4 * 4 * LP_XP + (size_t)4 * (((3 + a.ni3) * 80 + 127) / 128 * 128 + 128)) * 8;
    static PerInit attr;
    if (attr.first()) {
        if (hipFuncSetAttribute((const void *)dense_left_pass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)dense_left_pass_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("ttsk_dense_left_pass: cannot raise the dynamic LDS limit");
            return TTSK_ERR_HIP;
        }
    }
    const unsigned grid = (unsigned)(n2 * a.nct);
    const bool prof = prof_on();
    if (prof) prof_open_named(st, -2, 8.0 * l * (double)n0 * n1 * n2 * C, "dense_left_pass_kernel");
    if (l > 16) hipLaunchKernelGGL(dense_left_pass_kernel<true>, dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(dense_left_pass_kernel<false>, dim3(grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    // Z_2 = sum_{i2} slab2[i2], E_3 likewise: the slabs are [chunk = i2][M = l][N = C]
    ReduceOut r2{}, r3{};
    r2.C[0] = Z2;
    r3.C[0] = E3;
    int rc = launch_r_reduce(st, a.slab2, (int)n2, l, (int)C, 1, (int64_t)l, r2, 1, C, 1, 1.0, 0);
    if (rc == TTSK_OK) rc = launch_r_reduce(st, a.slab3, (int)n2, l, (int)C, 1, (int64_t)l, r3, 1, C, 1, 1.0, 0);
    if (prof) prof_close(st);
    if (rc != TTSK_OK) { set_error("ttsk_dense_left_pass: reduce launch failed"); return TTSK_ERR_HIP; }
    return TTSK_OK;
}
