// The two DRM chains of a tensor train and the Omega products WITHOUT the Psi cores: the first phase of the
// orthogonalising sketches (tt_orth.hip), on the kernels of the streaming sketch (tt_fused.hip).
#pragma once
#include <cstdint>

namespace ttsk {

struct TTChains {
    int want_left;             // 0: right chain only (hmt_sketch has no left DRM)
    double *const *omega;      // want_left: (d - 1) outputs per tensor, tensor-major: Omega_mu (lt[mu+1] x rt[d-1-mu]) contiguous
    const double *Rc[64];      // out: right contraction j (s[d-1-j] x rt[j+1]) of tensor 0 -- lives in `stream`'s DRIVER workspace
    const double *Lc[64];      // out: left contraction mu (s[mu+1] x lt[mu+1]) of tensor 0, or nullptr
    int64_t r_stride[64], l_stride[64];   // out: tensor b's contraction sits b * stride doubles behind tensor 0's
};

// full rank ranges only; lt / DL may be nullptr when want_left == 0
int tt_chains(int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt, const double *const *X,
              const double *const *DL, const double *const *DR, TTChains *out, int stream);
// the same for nb <= 32 tensors of one signature (X: nb * d pointers, tensor-major): every chain step one launch over all of them
int tt_chains_batch(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt, const double *const *X,
                    const double *const *DL, const double *const *DR, TTChains *out, int stream);

}  // namespace ttsk
