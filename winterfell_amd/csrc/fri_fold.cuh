// One row of folding::apply_drp (fri/src/folding/mod.rs:86-118) in registers: the N evaluations of the row (D components each) are
// interpolated (N-point inverse DFT), coefficient k scaled by (1/N) (offset^-1 g^-row)^k, and evaluated at alpha in the extension
// field.  Shared by the fold kernel (fri.hip) and the fold + next-layer commit kernel (fri_rows.hip).
#pragma once
#include "dft_regs.cuh"

// comp[d][j]: component d of element j, destroyed; io = offset^-1 * g^-(row); al = alpha; acc = the folded element
template <class F, int LOG_NF, int D>
__device__ __forceinline__ void fri_fold_row(typename F::T (&comp)[D][1 << LOG_NF], typename F::T io, typename F::T inv_n,
                                             const typename F::T (&al)[D], const typename F::T *w16, typename F::T (&acc)[D]) {
    typedef typename F::T T;
    constexpr int N = 1 << LOG_NF;
    // forward DFT per component (bit-reversed registers); inverse coefficient k = X[(N - k) mod N]
#pragma unroll
    for (int d = 0; d < D; d++) dft_dif<F, LOG_NF>(comp[d], w16);
    T scale[N];
    scale[0] = inv_n;
#pragma unroll
    for (int k = 1; k < N; k++) scale[k] = F::mul(scale[k - 1], io);
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = F::zero();
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        T tmp[D];
        F::template ext_mul<D>(acc, al, tmp);
        const int src = brev((N - k) & (N - 1), LOG_NF);
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = F::add(tmp[d], F::mul(comp[d][src], scale[k]));
    }
}
