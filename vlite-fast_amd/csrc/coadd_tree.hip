// The incoherent sum in a FIXED order (pb_coadd_tree / pb_coadd_local_tree): what makes the coadded filterbank of
// BASELINE configs[3] the same bytes on 1, 2, 4 or 8 GPUs.
//
// The reference sums nothing itself: one process_baseband per antenna host feeds its codes to the co ring
// (src/process_baseband.cu:1416-1422) and the external MPI coadder scripts/start_coadd:16,20-58 starts (one rank per
// antenna ring; source not in the repository) adds them in whatever order its reduction takes.  A floating-point sum
// whose order depends on the collective's topology cannot be checked byte for byte, so the order is DEFINED here
// (DESIGN.md section 6):
//
//     S(o, s) = plane of antenna o                      when o + s >= N_ant   (the set {o, o+s, ...} is one antenna)
//             = S(o, 2 s) + S(o + s, 2 s)               otherwise             (even members + odd members)
//     coadded = S(0, 1)
//
// i.e. the antennas are split by the parity of their index, recursively.  Listing a node's leaves left to right
// ("in order") makes its shape a function of the leaf count alone: T_1(x) = x,
// T_n(x_0 .. x_{n-1}) = T_ceil(n/2)(x_0 ..) + T_floor(n/2)(x_ceil(n/2) ..).  That is the only thing the kernel below
// knows; which planes are the leaves is the host's business (coadd.py: tree_order).  With antenna a on rank
// a mod W and W a power of two, rank r's antennas are exactly the node S(r, W): every rank evaluates T over its own
// planes, ships ONE plane, and the root evaluates T_W over the partial sums in bit-reversed rank order -- the same
// additions in the same association as one GPU holding all antennas.  (Any other W ships the antennas' planes
// themselves and the root evaluates the whole tree.)
//
// HBM-bound streaming: n planes read once, one written, float4 per lane, grid-stride over 1024 workgroups (4 per CU:
// every XCD's L2 sees a contiguous quarter-megabyte stripe per sweep).  -ffp-contract=off: plain IEEE additions.
#include "pb_internal.h"

struct CoaddLeaves {
    const float4 *p[PB_COADD_MAX_LEAVES];
};

template <int N, int O>
__device__ __forceinline__ float4 tree_sum(const CoaddLeaves &L, size_t i)
{
    if constexpr (N == 1) {
        return L.p[O][i];
    } else {
        const float4 a = tree_sum<(N + 1) / 2, O>(L, i);
        const float4 b = tree_sum<N / 2, O + (N + 1) / 2>(L, i);
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

template <int N>
__global__ __launch_bounds__(256) void k_coadd_tree(CoaddLeaves L, float4 *__restrict__ dst, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = tree_sum<N, 0>(L, i);
}

template <int N>
static void launch_n(const CoaddLeaves &L, float4 *dst, size_t n4, hipStream_t st)
{
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    k_coadd_tree<N><<<grid ? grid : 1, 256, 0, st>>>(L, dst, n4);
}

template <int N>
static void dispatch(int n, const CoaddLeaves &L, float4 *dst, size_t n4, hipStream_t st)
{
    if (n == N) launch_n<N>(L, dst, n4, st);
    else if constexpr (N > 1) dispatch<N - 1>(n, L, dst, n4, st);
}

// leaves[0 .. n): device planes of nfloat floats each (16-byte aligned, nfloat a multiple of 4), in tree order
hipError_t launch_coadd_tree(const float *const *leaves, int n, float *d_dst, size_t nfloat, hipStream_t st)
{
    if (n < 1 || n > PB_COADD_MAX_LEAVES || (nfloat & 3)) return hipErrorInvalidValue;
    CoaddLeaves L;
    for (int i = 0; i < PB_COADD_MAX_LEAVES; ++i) {
        const float *p = leaves[i < n ? i : 0];
        if (!p || ((uintptr_t)p & 15)) return hipErrorInvalidValue;
        L.p[i] = (const float4 *)p;
    }
    if (!d_dst || ((uintptr_t)d_dst & 15)) return hipErrorInvalidValue;
    if (nfloat == 0) return hipSuccess;
    dispatch<PB_COADD_MAX_LEAVES>(n, L, (float4 *)d_dst, nfloat / 4, st);
    return hipGetLastError();
}
