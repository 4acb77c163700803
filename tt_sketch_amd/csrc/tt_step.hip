// Fused chain step of the TT x TT-DRM sketch: one workgroup per mode index k.
//
//   T_k[a,q]     = sum_b  X_k[a,b] * Cin[b,q]                    (phase 1, K = s_in)
//   Out_k[a,q']  = sum_q  T_k[a,q] * D[q,k,q']                   (phase 2, K = rho)   -> slab k
//   Psi[q,k,c]   = sum_a  T_k[a,q] * R[a,c]                      (phase 3, K = s_out, left chain only)
//
// with X_k[a,b] = X[p''=a, k, p=b] for the right chain (the transposed tensor through strides) and
// X[p=b, k, p'=a] for the left chain.  This is one step of TensorTrainDRM.sketch_tt
// (reference tt_sketch/drm/tensor_train_drm.py:81-88) and, for the left chain, the interior
// sketch_psi_tt (tt_sketch/sketching_methods/tensor_train_sketch.py:28-34), which shares T_k.
// The chain matrix of the next step is sum_k Out_k: the slabs are written in accumulator order
// and summed (and un-permuted) by splitk_reduce_kernel -- the same reduction the unfused path uses.
//
// Why per-k: the three products of a step share T_k, which never leaves the CU.  T_k lives in LDS
// ((s_out) x (rho+2) doubles, <= 102 KB), the K-tiles of X_k / Cin / D_k / R are staged through
// two 30 KB buffers with the buffer_load machinery of gemm_kernel.h, and every wave keeps
// (all M tiles) x (one N tile) accumulators (8 waves, two per SIMD), so one k-step is 7 + 1 LDS reads and
// 6 DPP moves for up to 28 MFMAs.  Limits (LDS): s_out, rho, rhop, r, q_cnt <= 112; larger steps use ttsk_gemm.
#include "gemm_kernel.h"
#include "tt_step.h"

namespace ttsk {

constexpr int SMAX = 112;                // max extent of any M / N dimension of a phase
constexpr int TMAX = SMAX / 16;          // 7 tiles
constexpr int LDB = SMAX;                // staged B tile [k][n], ld = 112 == 16 (mod 32)

// Staging of a 112 x 32 (k-fast: [x][34]) or 32 x 112 (x-fast: [k][112]) tile by 512 threads:
// 1792 sixteen-byte pairs, 3.5 per thread -- the 4th pair index is clamped, so a few pairs are
// loaded and stored twice with identical values instead of branching.
template <bool KF>
struct Stage512 {
    uint32_t goff[4], lds[4];
    int kk[4];
    __device__ __forceinline__ void init(int tid, uint32_t xs8, uint32_t ski8)
    {
        constexpr int half = KF ? BK / 2 : SMAX / 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int idx = tid + 512 * e;
            idx = idx < SMAX * BK / 2 ? idx : SMAX * BK / 2 - 1;
            int x, k;
            if (KF) { k = 2 * (idx % half); x = idx / half; }
            else    { x = 2 * (idx % half); k = idx / half; }
            goff[e] = (uint32_t)x * xs8 + (uint32_t)k * ski8;
            lds[e] = (uint32_t)(KF ? x * LDKF + k : k * SMAX + x) * 8u;
            kk[e] = k;
        }
    }
    __device__ __forceinline__ void load(double2 (&r)[4], __amdgpu_buffer_rsrc_t rs, uint32_t soff, bool vec,
                                         uint32_t second_off) const
    {
        if (vec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ld16(rs, goff[e], soff);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[e].x = ld8(rs, goff[e], soff);
                r[e].y = ld8(rs, goff[e] + second_off, soff);
            }
        }
    }
    __device__ __forceinline__ void store(const double2 (&r)[4], char *S, bool mask_k, int kcount) const
    {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double2 v = r[e];
            if (mask_k) {
                if (kk[e] >= kcount) v.x = 0.0;
                if ((KF ? kk[e] + 1 : kk[e]) >= kcount) v.y = 0.0;
            }
            *reinterpret_cast<double2 *>(S + lds[e]) = v;
        }
    }
};

// 512 threads = 8 waves = two per SIMD: wave w owns N tile w (of <= 7) and all (<= 7) M tiles.  The
// two waves of a SIMD cover each other's LDS / barrier waits, so the loop needs no hand pipelining.
template <bool XKF>
__global__ __launch_bounds__(512) void tt_step_kernel(StepArgs g)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane >> 4, fj = lane & 15;
    const int li = lane >> 4, beta = (lane >> 2) & 3, jj = lane & 3;
    const int64_t k = blockIdx.x;
    const int TA = (g.s_out + 15) / 16, TQ = (g.rho + 15) / 16, TQP = (g.rhop + 15) / 16;
    const int ldt = TQ * 16 + 2;                       // T image [a][ldt]: 2*ldt == 4 (mod 8) -> conflict-free k-fast reads
    double *Timg = smem;
    double *As = smem + (size_t)TA * 16 * ldt;         // phase-1 A tile: k-fast [a][34] or x-fast [k][112]
    double *Bs = As + (XKF ? SMAX * LDKF : BK * SMAX); // staged B tile [k][112]
    char *Asb = reinterpret_cast<char *>(As), *Bsb = reinterpret_cast<char *>(Bs);

    TTSK_STAMP(40);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses every arbitration otherwise
    double acc[TMAX][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = 0.0;
    };
    // k-steps of one staged K-tile: A fragments through `afrag(i, ks)`, the B fragment of this wave's
    // N tile from Bs; tm M-tiles, tn N-tiles in total
    // k-steps of one staged K-tile: A fragments through `afrag(i, ks)`, the B fragment of this wave's
    // N tile from Bs; tm M-tiles, tn N-tiles in total.  Kept deliberately simple: variants with
    // hand software pipelining, compile-time tile counts or rotated B reads from LDS made hipcc
    // hoist every LDS read of the K-tile and spill (measured 60 us per step instead of 42).
    auto mma_tile = [&](auto afrag, int tm, int tn, int kcount) {
        if (wave >= tn) return;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            if (ks >= kcount) break;
            double a[TMAX], br[4];
#pragma unroll
            for (int i = 0; i < TMAX; ++i) a[i] = afrag(i < tm ? i : 0, ks);
            rot4(Bs[(ks + fi) * LDB + wave * 16 + fj], br);
#pragma unroll
            for (int i = 0; i < TMAX; ++i) {
                if (i < tm) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma4(a[i], br[t], acc[i][t]);
                }
            }
        }
    };

    // ------------------------------------------------------------------ phase 1: T = X_k * Cin
    {
        Stage512<XKF> sa;                  // 112 x 32 tile of X_k
        Stage512<false> sb;                // 32 x 112 tile of Cin
        sa.init(tid, (uint32_t)(g.x_a * 8), (uint32_t)(g.x_b * 8));
        sb.init(tid, 8u, (uint32_t)(g.ldc * 8));
        const __amdgpu_buffer_rsrc_t rsa = make_rsrc(g.X + k * g.x_k, (g.x_extent - k * g.x_k) * 8);
        const __amdgpu_buffer_rsrc_t rsb = make_rsrc(g.Cin, g.c_extent * 8);
        const uint32_t a2 = (uint32_t)((XKF ? g.x_b : g.x_a) * 8);
        double2 ra[4], rb[4];
        zero_acc();
        sa.load(ra, rsa, 0u, g.avec, a2);
        sb.load(rb, rsb, 0u, g.cvec, 8u);
        TTSK_STAMP(41);
        for (int k0 = 0; k0 < g.s_in; k0 += BK) {
            const int kcount = g.s_in - k0 < BK ? g.s_in - k0 : BK;
            TTSK_STAMP(48 + 4 * (k0 / BK));
            __syncthreads();
            TTSK_STAMP(80 + (k0 / BK));
            sa.store(ra, Asb, kcount < BK, kcount);
            sb.store(rb, Bsb, kcount < BK, kcount);
            TTSK_STAMP(49 + 4 * (k0 / BK));
            __syncthreads();
            TTSK_STAMP(50 + 4 * (k0 / BK));
            if (k0 + BK < g.s_in) {
                sa.load(ra, rsa, (uint32_t)((k0 + BK) * g.x_b * 8), g.avec, a2);
                sb.load(rb, rsb, (uint32_t)((k0 + BK) * g.ldc * 8), g.cvec, 8u);
            }
#ifdef TTSK_STAMPS
            if (blockIdx.x == 0 && lane == 0 && k0 == BK) g_stamps[100 + wave * 2] = __builtin_amdgcn_s_memtime();
#endif
            mma_tile([&](int i, int ks) { const int x = i * 16 + fj;
                                          return As[XKF ? x * LDKF + ks + fi : (ks + fi) * SMAX + x]; },
                     TA, TQ, kcount);
#ifdef TTSK_STAMPS
            if (blockIdx.x == 0 && lane == 0 && k0 == BK) { g_stamps[101 + wave * 2] = __builtin_amdgcn_s_memtime(); g_stamps[120 + wave] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 4) ; }
#endif
            TTSK_STAMP(51 + 4 * (k0 / BK));
        }
        TTSK_STAMP(42);
        // accumulators -> T image (zero outside s_out x rho so that it is a clean operand)
        if (wave < TQ) {
#pragma unroll
            for (int i = 0; i < TMAX; ++i)
                if (i < TA) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int a = i * 16 + 4 * beta + li, q = wave * 16 + 4 * ((beta + t) & 3) + jj;
                        Timg[a * ldt + q] = (a < g.s_out && q < g.rho) ? acc[i][t] : 0.0;
                    }
                }
        }
    }
    TTSK_STAMP(43);
    __syncthreads();
    TTSK_STAMP(44);

    // ------------------------------------------------------------------ phase 2: Out_k = T * D_k
    {
        Stage512<false> sb;                // 32 x 112 tile of D[:, k, :]
        sb.init(tid, 8u, (uint32_t)(g.d_q * 8));
        const __amdgpu_buffer_rsrc_t rsb = make_rsrc(g.D + k * g.d_k, (g.d_extent - k * g.d_k) * 8);
        double2 rb[4];
        zero_acc();
        sb.load(rb, rsb, 0u, g.dvec, 8u);
        for (int k0 = 0; k0 < g.rho; k0 += BK) {
            const int kcount = g.rho - k0 < BK ? g.rho - k0 : BK;
            __syncthreads();
            sb.store(rb, Bsb, kcount < BK, kcount);
            __syncthreads();
            if (k0 + BK < g.rho) sb.load(rb, rsb, (uint32_t)((k0 + BK) * g.d_q * 8), g.dvec, 8u);
            mma_tile([&](int i, int ks) { return Timg[(i * 16 + fj) * ldt + k0 + ks + fi]; }, TA, TQP, kcount);
        }
        TTSK_STAMP(45);
        if (wave < TQP) {
            double *pz = g.partial + (size_t)k * TA * TQP * 256;
#pragma unroll
            for (int i = 0; i < TMAX; ++i)
                if (i < TA) {
                    double *pt = pz + ((size_t)i * TQP + wave) * 256;
#pragma unroll
                    for (int t = 0; t < 4; ++t) pt[t * 64 + lane] = acc[i][t];
                }
        }
    }
    TTSK_STAMP(46);
    if (g.r == 0) return;

    // ------------------------------------------------------------------ phase 3: Psi[:,k,:] = T^T * R
    {
        const int TL = (g.q_cnt + 15) / 16, TC = (g.r + 15) / 16;
        Stage512<false> sb;                // 32 x 112 tile of R
        sb.init(tid, 8u, (uint32_t)(g.ldr * 8));
        const __amdgpu_buffer_rsrc_t rsb = make_rsrc(g.R, g.r_extent * 8);
        double2 rb[4];
        zero_acc();
        sb.load(rb, rsb, 0u, g.rvec, 8u);
        for (int k0 = 0; k0 < g.s_out; k0 += BK) {
            const int kcount = g.s_out - k0 < BK ? g.s_out - k0 : BK;
            __syncthreads();
            sb.store(rb, Bsb, kcount < BK, kcount);
            __syncthreads();
            if (k0 + BK < g.s_out) sb.load(rb, rsb, (uint32_t)((k0 + BK) * g.ldr * 8), g.rvec, 8u);
            // A[m = q][k = a] = T[a][q_lo + q]
            mma_tile([&](int i, int ks) { return Timg[(k0 + ks + fi) * ldt + g.q_lo + i * 16 + fj]; }, TL, TC,
                     kcount);
        }
        if (wave < TC) {
            double *pk = g.Psi + k * g.psi_k;
#pragma unroll
            for (int i = 0; i < TMAX; ++i)
                if (i < TL) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int q = i * 16 + 4 * beta + li, c = wave * 16 + 4 * ((beta + t) & 3) + jj;
                        if (q < g.q_cnt && c < g.r) {
                            double *p = pk + q * g.psi_q + c * g.psi_c;
                            *p = g.accumulate_psi ? *p + acc[i][t] : acc[i][t];
                        }
                    }
                }
        }
    }
}

// defined in gemm.hip
int launch_splitk_reduce(const ttsk_gemm_desc &d, const double *partial, double *C, int splits, int64_t tiles_m,
                         int64_t tiles_n, int rota, hipStream_t st);

static bool even64(int64_t v) { return (v & 1) == 0; }
static bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

bool tt_step_fits(int64_t s_in, int64_t s_out, int64_t rho, int64_t rhop, int64_t r, int64_t q_cnt,
                  int64_t x_span_elems)
{
    return s_in >= 1 && s_out <= SMAX && rho <= SMAX && rhop <= SMAX && r <= SMAX && q_cnt <= SMAX &&
           x_span_elems * 8 < (1ll << 31);
}

// One fused step.  xkf: the contracted rank index of X is the contiguous one (right chain).
// Out (s_out x rhop, row stride ld_out) = sum_k slabs; Psi optional.
int tt_step_launch(bool xkf, int64_t n, StepArgs g, double *Out, int64_t ld_out, int stream)
{
    hipStream_t st = stream_of(stream);
    if (!st) return TTSK_ERR_ARG;
    const int TA = (g.s_out + 15) / 16, TQ = (g.rho + 15) / 16, TQP = (g.rhop + 15) / 16;
    g.partial = (double *)scratch(stream, SCRATCH_GEMM, (size_t)n * TA * TQP * 256 * 8 + 64);
    if (!g.partial) return TTSK_ERR_HIP;
    // 16-byte loads: the contiguous index pairs up, even strides, aligned bases, even extents
    if (xkf) g.avec = g.x_b == 1 && even64(g.x_a) && even64(g.x_k) && even64(g.s_in) && al16(g.X);
    else g.avec = g.x_a == 1 && even64(g.x_b) && even64(g.x_k) && al16(g.X);
    g.cvec = even64(g.ldc) && al16(g.Cin);
    g.dvec = even64(g.d_q) && even64(g.d_k) && al16(g.D);
    g.rvec = g.r > 0 && even64(g.ldr) && al16(g.R);
    const int ldt = TQ * 16 + 2;
    const size_t lds = 8 * ((size_t)TA * 16 * ldt + (xkf ? SMAX * LDKF : BK * SMAX) + BK * LDB);
    static bool attr_set = false;
    if (!attr_set) {   // > 64 KB of dynamic LDS needs the opt-in once per kernel
        TTSK_HIP(hipFuncSetAttribute((const void *)tt_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)tt_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        attr_set = true;
    }
    if (xkf) hipLaunchKernelGGL(tt_step_kernel<true>, dim3((unsigned)n), dim3(512), lds, st, g);
    else hipLaunchKernelGGL(tt_step_kernel<false>, dim3((unsigned)n), dim3(512), lds, st, g);
    TTSK_LAUNCH_CHECK();
    ttsk_gemm_desc d{};
    d.batch = 1; d.M = g.s_out; d.N = g.rhop; d.Ko = 1; d.Ki = 1;
    d.c_m = ld_out; d.c_n = 1; d.alpha = 1.0; d.accumulate = 0;
    return launch_splitk_reduce(d, g.partial, Out, (int)n, TA, TQP, 0, st);
}

}  // namespace ttsk
