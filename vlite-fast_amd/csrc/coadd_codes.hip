// The incoherent sum taken from the QUANTISED filterbank instead of the fp32 planes (pb_coadd_local_codes).
//
// The reference's coadder (scripts/start_coadd:16, source not in the repository) read the per-antenna co rings,
// i.e. the bytes writer.c:343-352 copies there: already 8/4/2-bit codes.  Whatever it summed had been through
// sel_and_dig (src/pb_kernels.cu:517-582) once per antenna.  pb_coadd_local sums the fp32 planes before that
// step, which is the better arithmetic; this mode exists so that the two can be compared on the same data:
//   d_sum[i] (+)= sum over this handle's antennas of level(code_a[i])
// with level() the centre of the quantiser's cell (8 bit: (q - 127) * 0.02957, 4 bit: (q - 7) * 0.3188, 2 bit:
// the mid-points of the thresholds -0.6109 / 0.3970 / 1.4050 and one cell width beyond the outer two).
// d_sum has the layout of the fp32 planes (compact [time][4096], or [pol][time][4096] with two polarisations) so
// the cross-GPU reduce and pb_coadd_finish are the ones of the fp32 mode.  HBM-bound byte work: one byte in per
// antenna and sample, one float out.
#include "pb_internal.h"

__device__ __forceinline__ float code_level(unsigned q, int nbit)
{
    if (nbit == 8) return (float)(((double)q - 127.0) * 0.02957);
    if (nbit == 4) return (float)(((double)q - 7.0) * 0.3188);
    // cell width (1.4050 + 0.6109) / 2 = 1.00795
    const float lv[4] = {(float)(-0.6109 - 0.503975), (float)(0.5 * (-0.6109 + 0.3970)),
                         (float)(0.5 * (0.3970 + 1.4050)), (float)(1.4050 + 0.503975)};
    return lv[q & 3];
}

// one thread per output float; i runs over the plane layout, n over sel_and_dig's sample order
__global__ __launch_bounds__(256) void k_coadd_local_codes(const uint8_t *__restrict__ codes, size_t ant_stride, int A,
                                                           float *__restrict__ sum, size_t ave_per_seg, size_t trim,
                                                           int nseg, int npol, int nbit, int ntime, int accumulate)
{
    const size_t total = (size_t)nseg * ave_per_seg;
    const unsigned mask = (1u << nbit) - 1u;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t seg = i / ave_per_seg, src = i % ave_per_seg;
        size_t n = src;
        if (npol == 2) {
            const size_t pol = src / ((size_t)ntime * PB_NCHANOUT), rem = src % ((size_t)ntime * PB_NCHANOUT);
            const size_t trow = rem / PB_NCHANOUT, c = rem % PB_NCHANOUT;
            n = (trow * 2 + pol) * PB_NCHANOUT + c;
        }
        const size_t bit = n * nbit;
        const size_t at = seg * trim + (bit >> 3);
        const unsigned sh = (unsigned)(bit & 7);
        float s = accumulate ? sum[i] : 0.f;
        for (int a = 0; a < A; ++a) s += code_level((codes[(size_t)a * ant_stride + at] >> sh) & mask, nbit);
        sum[i] = s;
    }
}

hipError_t launch_coadd_local_codes(pb_handle *h, int nseg, float *d_sum, int accumulate, hipStream_t st)
{
    const int stream = h->cfg.rfi_mode == 0 ? 0 : 1;
    const uint8_t *codes = h->d_codes + (size_t)stream * h->S * h->trim;
    k_coadd_local_codes<<<1024, 256, 0, st>>>(codes, (size_t)2 * h->S * h->trim, h->A, d_sum, h->ave_per_seg, h->trim,
                                              nseg, h->cfg.npol, h->cfg.nbit, h->R / PB_NSCRUNCH, accumulate);
    return hipGetLastError();
}
