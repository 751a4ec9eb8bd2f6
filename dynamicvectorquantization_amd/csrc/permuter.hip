// Dual-grain code <-> sequence permutation of the stage-2 transformer input (integer work, bit-exact):
//   DualGrainSeperatePermuter.forward / forward_back   modules/dynamic_modules/permuter.py:50-135
// forward: per image, compact the codes of the coarse cells (their top-left code) and of the fine cells (all hw2 x hw2
// codes; "region-first" = cell by cell, "row-first" = raster order of the fine grid) into EOS-terminated, PAD-filled rows
// together with their position ids.  One workgroup per image: flags -> block-wide exclusive scan -> scatter.
// forward_back: scatter the sequences back into the [fine_hw, fine_hw] code map with the reference's exact semantics
// (coarse codes are broadcast only if the coarse EOS is present; entries after the first EOS are ignored; a position
// written twice keeps the LAST write).
#include "dvq_common.h"

namespace {

constexpr int PT = 256;

// exclusive scan of per-thread counts; returns the thread's offset, *total = block total
__device__ int block_excl_scan(int v, int* lds /* [PT] */, int* total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < PT; off <<= 1) {
        const int add = t >= off ? lds[t - off] : 0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const int incl = lds[t];
    *total = lds[PT - 1];
    __syncthreads();
    return incl - v;
}

struct PermParams {
    const int64_t* idx;     // [B, fine_hw, fine_hw]
    const int64_t* grain;   // [B, hw1, hw1]
    int hw1, hw2, order;    // order 0 region-first, 1 row-first
    int64_t content_pad, content_eos, cpos_pad, cpos_eos, fpos_pad, fpos_eos;
    int64_t *cc, *cp, *fc, *fp;   // [B, ncell+1], [B, ncell+1], [B, npix+1], [B, npix+1]
    int* counts;                  // [B][2]: coarse cells, fine codes
};

__global__ __launch_bounds__(PT) void permute_dual_kernel(PermParams p) {
    __shared__ int lds[PT];
    const int b = blockIdx.x, t = threadIdx.x;
    const int ncell = p.hw1 * p.hw1, fhw = p.hw1 * p.hw2, npix = fhw * fhw, q = p.hw2 * p.hw2;
    const int64_t* idx = p.idx + (int64_t)b * npix;
    const int64_t* gr = p.grain + (int64_t)b * ncell;
    int64_t* cc = p.cc + (int64_t)b * (ncell + 1);
    int64_t* cp = p.cp + (int64_t)b * (ncell + 1);
    int64_t* fc = p.fc + (int64_t)b * (npix + 1);
    int64_t* fp = p.fp + (int64_t)b * (npix + 1);
    // ---- coarse stream + region-first fine stream: scan over cells (each thread owns a contiguous run of cells)
    const int per = (ncell + PT - 1) / PT;
    const int c0 = t * per, c1 = min(ncell, c0 + per);
    int n0 = 0, n1 = 0;
    for (int c = c0; c < c1; ++c) {
        n0 += gr[c] == 0;
        n1 += gr[c] == 1;
    }
    int tot0, tot1;
    int o0 = block_excl_scan(n0, lds, &tot0);
    int o1 = block_excl_scan(n1, lds, &tot1);
    for (int c = c0; c < c1; ++c) {
        const int h1 = c / p.hw1, w1 = c - h1 * p.hw1;
        if (gr[c] == 0) {
            cc[o0] = idx[(int64_t)(h1 * p.hw2) * fhw + w1 * p.hw2];
            cp[o0] = c;
            ++o0;
        } else if (gr[c] == 1 && p.order == 0) {
            for (int j = 0; j < q; ++j) {
                const int h2 = j / p.hw2, w2 = j - h2 * p.hw2;
                const int pos = (h1 * p.hw2 + h2) * fhw + w1 * p.hw2 + w2;
                fc[(int64_t)o1 * q + j] = idx[pos];
                fp[(int64_t)o1 * q + j] = pos;
            }
            ++o1;
        }
    }
    int nfine = tot1 * q;
    if (p.order == 1) {
        // ---- row-first fine stream: scan over the raster of the fine grid
        const int perp = (npix + PT - 1) / PT;
        const int p0 = t * perp, p1 = min(npix, p0 + perp);
        int nf = 0;
        for (int e = p0; e < p1; ++e) nf += gr[(e / fhw / p.hw2) * p.hw1 + (e % fhw) / p.hw2] == 1;
        int totf;
        int of = block_excl_scan(nf, lds, &totf);
        for (int e = p0; e < p1; ++e)
            if (gr[(e / fhw / p.hw2) * p.hw1 + (e % fhw) / p.hw2] == 1) {
                fc[of] = idx[e];
                fp[of] = e;
                ++of;
            }
        nfine = totf;
    }
    // ---- EOS + padding
    for (int e = tot0 + t; e <= ncell; e += PT) {
        cc[e] = e == tot0 ? p.content_eos : p.content_pad;
        cp[e] = e == tot0 ? p.cpos_eos : p.cpos_pad;
    }
    for (int e = nfine + t; e <= npix; e += PT) {
        fc[e] = e == nfine ? p.content_eos : p.content_pad;
        fp[e] = e == nfine ? p.fpos_eos : p.fpos_pad;
    }
    if (t == 0) {
        p.counts[2 * b] = tot0;
        p.counts[2 * b + 1] = nfine;
    }
}

struct BackParams {
    const int64_t *cc, *fc, *cp, *fp;   // [B,Lc], [B,Lf], [B,Lc], [B,Lf]
    int Lc, Lf, hw1, hw2;
    int64_t cpos_eos, fpos_eos;
    int64_t* out;                       // [B, fine_hw, fine_hw]
};

__global__ __launch_bounds__(PT) void permute_dual_back_kernel(BackParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, t = threadIdx.x;
    const int ncell = p.hw1 * p.hw1, fhw = p.hw1 * p.hw2, npix = fhw * fhw;
    int* last_c = reinterpret_cast<int*>(smem);         // [ncell] last sequence index writing a coarse cell
    int* last_f = last_c + ncell;                       // [npix]
    __shared__ int eos_c, eos_f;
    const int64_t* cc = p.cc + (int64_t)b * p.Lc;
    const int64_t* cp = p.cp + (int64_t)b * p.Lc;
    const int64_t* fc = p.fc + (int64_t)b * p.Lf;
    const int64_t* fp = p.fp + (int64_t)b * p.Lf;
    if (t == 0) {
        eos_c = p.Lc;
        eos_f = p.Lf;
    }
    for (int c = t; c < ncell; c += PT) last_c[c] = -1;
    for (int e = t; e < npix; e += PT) last_f[e] = -1;
    __syncthreads();
    for (int k = t; k < p.Lc; k += PT)
        if (cp[k] == p.cpos_eos) atomicMin(&eos_c, k);
    for (int k = t; k < p.Lf; k += PT)
        if (fp[k] == p.fpos_eos) atomicMin(&eos_f, k);
    __syncthreads();
    const bool has_eos = eos_c < p.Lc;
    for (int k = t; k < eos_c; k += PT) {
        const int64_t pos = cp[k];
        if (pos >= 0 && pos < ncell) atomicMax(&last_c[(int)pos], k);
    }
    for (int k = t; k < eos_f; k += PT) {
        const int64_t pos = fp[k];
        if (pos >= 0 && pos < npix) atomicMax(&last_f[(int)pos], k);
    }
    __syncthreads();
    int64_t* out = p.out + (int64_t)b * npix;
    for (int e = t; e < npix; e += PT) {
        int64_t v = 0;
        if (has_eos) {
            const int cell = (e / fhw / p.hw2) * p.hw1 + (e % fhw) / p.hw2;
            const int lc = last_c[cell];
            v = lc >= 0 ? cc[lc] : 0;
        }
        const int lf = last_f[e];
        if (lf >= 0) v = fc[lf];
        out[e] = v;
    }
}

}  // namespace

extern "C" {

int dvq_permute_dual(const int64_t* indices, const int64_t* grain, int64_t B, int hw1, int hw2, int order, int64_t content_pad,
                     int64_t content_eos, int64_t cpos_pad, int64_t cpos_eos, int64_t fpos_pad, int64_t fpos_eos,
                     int64_t* coarse_content, int64_t* coarse_position, int64_t* fine_content, int64_t* fine_position,
                     int* counts, dvq_stream_t stream) {
    DVQ_REQUIRE(indices && grain && coarse_content && coarse_position && fine_content && fine_position && counts, DVQ_EINVAL,
                "dvq_permute_dual: null pointer");
    DVQ_REQUIRE(B > 0 && B < (1 << 30) && hw1 > 0 && hw2 > 0 && hw1 * hw2 <= 256 && (order == 0 || (order == 1 && hw2 == 2)),
                DVQ_ESHAPE, "dvq_permute_dual: bad geometry (row-first needs hw2 == 2 like the reference)");
    PermParams p{};
    p.idx = indices; p.grain = grain; p.hw1 = hw1; p.hw2 = hw2; p.order = order;
    p.content_pad = content_pad; p.content_eos = content_eos; p.cpos_pad = cpos_pad; p.cpos_eos = cpos_eos;
    p.fpos_pad = fpos_pad; p.fpos_eos = fpos_eos;
    p.cc = coarse_content; p.cp = coarse_position; p.fc = fine_content; p.fp = fine_position; p.counts = counts;
    permute_dual_kernel<<<dim3((unsigned)B), dim3(PT), 0, (hipStream_t)stream>>>(p);
    DVQ_CHECK_LAUNCH("permute_dual");
    return DVQ_OK;
}

int dvq_permute_dual_back(const int64_t* coarse_content, const int64_t* fine_content, const int64_t* coarse_position,
                          const int64_t* fine_position, int64_t B, int64_t Lc, int64_t Lf, int hw1, int hw2, int64_t cpos_eos,
                          int64_t fpos_eos, int64_t* out, dvq_stream_t stream) {
    DVQ_REQUIRE(coarse_content && fine_content && coarse_position && fine_position && out, DVQ_EINVAL,
                "dvq_permute_dual_back: null pointer");
    DVQ_REQUIRE(B > 0 && B < (1 << 30) && Lc > 0 && Lf > 0 && Lc < (1 << 30) && Lf < (1 << 30) && hw1 > 0 && hw2 > 0 &&
                    hw1 * hw2 <= 128,
                DVQ_ESHAPE, "dvq_permute_dual_back: bad geometry");
    BackParams p{};
    p.cc = coarse_content; p.fc = fine_content; p.cp = coarse_position; p.fp = fine_position;
    p.Lc = (int)Lc; p.Lf = (int)Lf; p.hw1 = hw1; p.hw2 = hw2; p.cpos_eos = cpos_eos; p.fpos_eos = fpos_eos; p.out = out;
    const int fhw = hw1 * hw2;
    const int lds = (hw1 * hw1 + fhw * fhw) * (int)sizeof(int);
    dvq_ensure_dynamic_lds((const void*)permute_dual_back_kernel, lds);
    permute_dual_back_kernel<<<dim3((unsigned)B), dim3(PT), lds, (hipStream_t)stream>>>(p);
    DVQ_CHECK_LAUNCH("permute_dual_back");
    return DVQ_OK;
}

}  // extern "C"
