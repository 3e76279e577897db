// hist.hip -- HBM-bound half of the hot path for gfx950:
//   lookup_regular          grid -> event gather          (translation.py:417-501)
//   histogram_regular       weighted N-D histogram        (translation.py:90-205)
//   reweight_hist           fused prob3.apply + aeff.apply + hist.apply(sumw2)
//   hist_finalize           fixed point -> fp64 maps
//
// Accumulation is ORDER INDEPENDENT: every summand is converted to 192-bit
// fixed point (LSB 2^-116, top 2^76) split into six 32-bit-payload limbs held
// in 64-bit integers, so plain integer atomics (LDS ds_add_u64, no carries)
// give exact sums whatever the event order, workgroup count or GPU count.  A
// block's partial sums go to a slab with plain coalesced stores (no global
// float atomics), a second tiny kernel adds the slabs and normalises carries;
// the result can be SUM-all-reduced across ranks as int64 and is finally
// rounded ONCE (round-to-nearest-even) to fp64.
//
// Roofline: HBM.  Algorithmic traffic of the fused kernel is
// 8 B x (2 lookup coords + 2 flux + aeff + w0 + D sample columns) per event
// = 72 B (D=3) / 64 B (D=2); the probability tables (<= 5.8 MB) stay in L2.
#include "common.hpp"

namespace pisa {

constexpr int NL = PISA_HIP_ACC_LIMBS;  // limbs per accumulator
constexpr int FX_LSB = 116;             // value = sum limb_k * 2^(32k - 116)
constexpr int MAX_CONT = 16;            // containers per launch (kernarg budget)
constexpr int HIST_THREADS = 256;
constexpr int64_t LDS_ACC_BYTES_MAX = 64 * 1024;

// ---------------------------------------------------------------------------
// double -> signed 6-limb fixed point.  Returns false for NaN/Inf/|x| >= 2^76.
// limb values are in (-2^32, 2^32); q is the index of the lowest touched limb.
struct Fx {
    long long v[3];
    int q;
};

__device__ __forceinline__ bool to_fixed(double x, Fx &f) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff);
    unsigned long long m = bits & 0xfffffffffffffULL;
    bool neg = (bits >> 63) != 0;
    if (e == 0x7ff) return false;
    if (e == 0) {  // zero / subnormal: far below 2^-116
        f.q = 0; f.v[0] = f.v[1] = f.v[2] = 0;
        return true;
    }
    m |= (1ULL << 52);
    int shift = e - (1075 - FX_LSB);  // F = m * 2^shift
    if (shift > 192 - 53) return false;
    unsigned int l0, l1, l2;
    int q;
    if (shift < 0) {
        int s = -shift;
        unsigned long long t = (s >= 64) ? 0ULL : (m >> s);
        q = 0;
        l0 = (unsigned int)t;
        l1 = (unsigned int)(t >> 32);
        l2 = 0;
    } else {
        q = shift >> 5;
        int r = shift & 31;
        unsigned long long lo = m << r;                   // low 64 bits
        unsigned long long hi = r ? (m >> (64 - r)) : 0;  // spill (<= 20 bits)
        l0 = (unsigned int)lo;
        l1 = (unsigned int)(lo >> 32);
        l2 = (unsigned int)hi;
    }
    f.q = q;
    f.v[0] = neg ? -(long long)l0 : (long long)l0;
    f.v[1] = neg ? -(long long)l1 : (long long)l1;
    f.v[2] = neg ? -(long long)l2 : (long long)l2;
    return true;
}

// accumulator layout: acc[(limb * n_entries) + entry], entry = bin*2 + quantity
template <bool LDS_ACC>
__device__ __forceinline__ void acc_add(unsigned long long *acc, int64_t n_entries, int64_t entry,
                                        const Fx &f) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int limb = f.q + k;
        if (f.v[k] != 0 && limb < NL)
            atomicAdd(&acc[(int64_t)limb * n_entries + entry], (unsigned long long)f.v[k]);
    }
}

__device__ __forceinline__ bool bin_index(const DevBinning &b, double x, double y, double z,
                                          int64_t &flat) {
    // half-open [min, max) per dimension, bin = (int)((x - min) * norm)
    // (fast_histogram rule; translation.py:417-456 for lookups)
    if (!(x >= b.mins[0] && x < b.maxs[0])) return false;
    int ix = (int)((x - b.mins[0]) * b.norm[0]);
    ix = ix < b.nb[0] ? ix : b.nb[0] - 1;
    flat = ix;
    if (b.ndim > 1) {
        if (!(y >= b.mins[1] && y < b.maxs[1])) return false;
        int iy = (int)((y - b.mins[1]) * b.norm[1]);
        iy = iy < b.nb[1] ? iy : b.nb[1] - 1;
        flat = flat * b.nb[1] + iy;
    }
    if (b.ndim > 2) {
        if (!(z >= b.mins[2] && z < b.maxs[2])) return false;
        int iz = (int)((z - b.mins[2]) * b.norm[2]);
        iz = iz < b.nb[2] ? iz : b.nb[2] - 1;
        flat = flat * b.nb[2] + iz;
    }
    return true;
}

// ---------------------------------------------------------------------------
struct ContDev {
    int64_t n;
    const double *gx, *gy, *flux, *aeff, *w0;
    const double *s[3];
    double scale;
    int32_t flav, side;
};

struct HistArgs {
    int32_t n_cont;
    int32_t fused;        // 1: reweight chain, quantities (w, w^2); 0: (w, 1)
    int64_t n_bins;
    int64_t chunk;        // events per workgroup
    DevBinning grid;      // calc grid (lookup)
    DevBinning outb;      // output binning
    const double *prob[2];
    ContDev cont[MAX_CONT];
    int32_t blk_start[MAX_CONT + 1];
    int32_t slab_base;    // first slab row used by this launch
};

template <bool FUSED, bool LDS_ACC>
__global__ void __launch_bounds__(HIST_THREADS)
hist_accumulate_kernel(const HistArgs a, unsigned long long *__restrict__ slab_or_acc,
                       int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];
    // locate this workgroup's container (workgroup-uniform scalar loop)
    int c = 0;
    const int bid = blockIdx.x;
    while (c + 1 < a.n_cont && bid >= a.blk_start[c + 1]) c++;
    const ContDev &C = a.cont[c];
    const int64_t lb = bid - a.blk_start[c];
    const int64_t start = lb * a.chunk;
    int64_t end = start + a.chunk;
    if (end > C.n) end = C.n;
    const int64_t n_entries = a.n_bins * 2;

    unsigned long long *acc;
    if (LDS_ACC) {
        acc = s_acc;
        for (int64_t k = threadIdx.x; k < n_entries * NL; k += HIST_THREADS) acc[k] = 0ULL;
        __syncthreads();
    } else {
        acc = slab_or_acc + (int64_t)c * n_entries * NL;  // global accumulators per container
    }

    const double *prob = FUSED ? a.prob[C.side] : nullptr;
    const int po_e = 0 * 3 + C.flav;  // P[e  -> flav]
    const int po_mu = 1 * 3 + C.flav; // P[mu -> flav]
    bool bad = false;

    for (int64_t i = start + threadIdx.x; i < end; i += HIST_THREADS) {
        double w;
        if (FUSED) {
            // grid -> event lookup of prob_e, prob_mu (container.py:981-1012,
            // translation.py:427-438): 0 outside the grid
            double pe = 0.0, pmu = 0.0;
            int64_t node;
            if (bin_index(a.grid, C.gx[i], a.grid.ndim > 1 ? C.gy[i] : 0.0, 0.0, node)) {
                pe = prob[9 * node + po_e];
                pmu = prob[9 * node + po_mu];
            }
            double2 f = reinterpret_cast<const double2 *>(C.flux)[i];
            w = C.w0[i];
            w = w * ((f.x * pe) + (f.y * pmu));   // prob3.py:622
            w = w * (C.aeff[i] * C.scale);        // aeff.py:87
        } else {
            w = C.w0 ? C.w0[i] : 1.0;
        }
        int64_t bin;
        double x = C.s[0][i];
        double y = a.outb.ndim > 1 ? C.s[1][i] : 0.0;
        double z = a.outb.ndim > 2 ? C.s[2][i] : 0.0;
        if (!bin_index(a.outb, x, y, z, bin)) continue;
        Fx f0, f1;
        bool ok = to_fixed(w, f0);
        ok = to_fixed(FUSED ? w * w : 1.0, f1) && ok;
        if (!ok) { bad = true; continue; }
        acc_add<LDS_ACC>(acc, n_entries, bin * 2 + 0, f0);
        acc_add<LDS_ACC>(acc, n_entries, bin * 2 + 1, f1);
    }
    if (bad && status) atomicOr(status, 1);

    if (LDS_ACC) {
        __syncthreads();
        unsigned long long *row = slab_or_acc + (int64_t)(a.slab_base + bid) * n_entries * NL;
        for (int64_t k = threadIdx.x; k < n_entries * NL; k += HIST_THREADS) row[k] = acc[k];
    }
}

// Adds the slab rows of each container and normalises carries so that every
// limb is in [0, 2^32) (top limb keeps the sign).  out[c][entry][limb] int64.
struct ReduceArgs {
    int32_t n_cont;
    int32_t from_global;  // 1: accumulators already summed in global memory
    int64_t n_entries;    // n_bins * 2
    int32_t blk_start[MAX_CONT + 1];
    int32_t slab_base;
    int32_t cont_base;    // first container index of this batch in the output
};

__global__ void __launch_bounds__(256)
hist_reduce_kernel(const ReduceArgs a, const unsigned long long *__restrict__ slab,
                   long long *__restrict__ out, int32_t *__restrict__ status) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = blockIdx.y;
    if (e >= a.n_entries) return;
    long long limb[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) limb[k] = 0;
    if (a.from_global) {
        const unsigned long long *acc = slab + (int64_t)c * a.n_entries * NL;
#pragma unroll
        for (int k = 0; k < NL; k++) limb[k] = (long long)acc[(int64_t)k * a.n_entries + e];
    } else {
        for (int b = a.blk_start[c]; b < a.blk_start[c + 1]; b++) {
            const unsigned long long *row = slab + (int64_t)(a.slab_base + b) * a.n_entries * NL;
#pragma unroll
            for (int k = 0; k < NL; k++) limb[k] += (long long)row[(int64_t)k * a.n_entries + e];
        }
    }
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        long long v = limb[k] + carry;
        carry = v >> 32;  // arithmetic shift = floor division
        limb[k] = v & 0xffffffffLL;
    }
    limb[NL - 1] += carry;  // signed top limb
    if (limb[NL - 1] >= (1LL << 31) || limb[NL - 1] < -(1LL << 31))
        if (status) atomicOr(status, 1);
    long long *o = out + ((int64_t)(a.cont_base + c) * a.n_entries + e) * NL;
#pragma unroll
    for (int k = 0; k < NL; k++) o[k] = limb[k];
}

// limbs (possibly summed over ranks) -> fp64, rounded once (RNE).
__device__ __forceinline__ double limbs_to_double(const long long *in) {
    long long L[NL + 1];
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        long long v = in[k] + carry;
        carry = v >> 32;
        L[k] = v & 0xffffffffLL;
    }
    L[NL] = carry;  // signed remainder
    bool neg = L[NL] < 0;
    if (neg) {
        // two's complement negate of the (NL+1)-limb number
        long long c2 = 1;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            long long v = (0xffffffffLL - L[k]) + c2;
            c2 = v >> 32;
            L[k] = v & 0xffffffffLL;
        }
        L[NL] = ~L[NL] + c2;
    }
    int t = -1;
#pragma unroll
    for (int k = 0; k <= NL; k++)
        if (L[k] != 0) t = k;
    if (t < 0) return 0.0;
    // 96-bit window from limbs t, t-1, t-2; everything below is sticky
    unsigned long long hi = (unsigned long long)L[t];  // < 2^32 (or small carry)
    unsigned long long mid = t >= 1 ? (unsigned long long)L[t - 1] : 0ULL;
    unsigned long long low = t >= 2 ? (unsigned long long)L[t - 2] : 0ULL;
    bool sticky = false;
    for (int k = 0; k + 3 <= t; k++) sticky = sticky || (L[k] != 0);
    // value = (hi*2^64 + mid*2^32 + low) * 2^(32*(t-2) - FX_LSB)
    int lz = __clzll(hi);              // hi != 0
    int hb = 64 - lz;                  // significant bits of hi (<= 33)
    // top 64 bits of the (hb+64)-bit number hi:mid:low
    unsigned long long ml = (mid << 32) | low;
    unsigned long long top = (hi << (64 - hb)) | (hb < 64 ? (ml >> hb) : 0ULL);
    unsigned long long lost = hb < 64 ? (ml << (64 - hb)) : ml;
    if (hb == 0) { top = ml; lost = 0; }
    if (lost != 0 || sticky) top |= 1ULL;
    double d = (double)top;  // u64 -> f64 is round-to-nearest-even
    int exp2 = hb + 32 * (t - 2) - FX_LSB;
    d = ldexp(d, exp2);
    return neg ? -d : d;
}

__global__ void __launch_bounds__(256)
hist_finalize_kernel(const long long *__restrict__ limbs, int64_t n_total_bins,
                     double *__restrict__ hist, double *__restrict__ q1) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_total_bins) return;
    if (hist) hist[b] = limbs_to_double(limbs + (b * 2 + 0) * NL);
    if (q1) q1[b] = limbs_to_double(limbs + (b * 2 + 1) * NL);
}

__global__ void __launch_bounds__(256)
hist_average_kernel(int64_t n_bins, double *__restrict__ hist, const double *__restrict__ cnt) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    double v = hist[b] / cnt[b];  // translation.py:118-127: x/0 -> nan -> 0
    if (!(v == v)) v = 0.0;
    if (isinf(v)) v = v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    hist[b] = v;
}

__global__ void __launch_bounds__(256)
lookup_regular_kernel(const DevBinning b, const double *__restrict__ x,
                      const double *__restrict__ y, const double *__restrict__ z, int64_t n,
                      const double *__restrict__ flat_hist, int width, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t node;
    bool in = bin_index(b, x[i], b.ndim > 1 ? y[i] : 0.0, b.ndim > 2 ? z[i] : 0.0, node);
    for (int w = 0; w < width; w++) out[i * width + w] = in ? flat_hist[node * width + w] : 0.0;
}

__global__ void __launch_bounds__(256)
apply_osc_weights_kernel(const double *__restrict__ flux, const double *__restrict__ pe,
                         const double *__restrict__ pmu, int64_t n, double *__restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double2 f = reinterpret_cast<const double2 *>(flux)[i];
    w[i] = w[i] * ((f.x * pe[i]) + (f.y * pmu[i]));
}

__global__ void __launch_bounds__(256)
apply_aeff_kernel(const double *__restrict__ aeff, double scale, int64_t n,
                  double *__restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    w[i] = w[i] * (aeff[i] * scale);
}

// ---------------------------------------------------------------- host side
static int64_t acc_entries_bytes(int64_t n_bins) { return n_bins * 2 * NL * 8; }

static int plan_blocks(const int64_t *n_events, int n_cont, int64_t &chunk, int32_t *blk_start) {
    int64_t total = 0;
    for (int c = 0; c < n_cont; c++) total += n_events[c];
    // ~4 workgroups per CU; a chunk is a whole number of 256-event sweeps
    const int64_t target_blocks = 1024;
    chunk = (total + target_blocks - 1) / target_blocks;
    if (chunk < 2048) chunk = 2048;
    chunk = ((chunk + HIST_THREADS - 1) / HIST_THREADS) * HIST_THREADS;
    blk_start[0] = 0;
    for (int c = 0; c < n_cont; c++) {
        int64_t nb = (n_events[c] + chunk - 1) / chunk;
        blk_start[c + 1] = blk_start[c] + (int32_t)nb;
    }
    return blk_start[n_cont];
}

constexpr int64_t MAX_SLAB_ROWS = 1024 + 4 * MAX_CONT;

// optional hipEvent pair recorded around the accumulate kernel of the next
// hist launch (bench.py measures the dominant kernel with them)
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

static int run_hist(const ContDev *conts, int n_cont, bool fused, const DevBinning *grid,
                    const double *prob_nu, const double *prob_nubar, const DevBinning &outb,
                    int64_t n_bins, long long *d_limbs, void *d_workspace, int32_t *d_status,
                    hipStream_t s) {
    const int64_t row_bytes = acc_entries_bytes(n_bins);
    const bool lds = row_bytes <= LDS_ACC_BYTES_MAX;
    unsigned long long *ws = reinterpret_cast<unsigned long long *>(d_workspace);
    for (int base = 0; base < n_cont; base += MAX_CONT) {
        int nc = n_cont - base < MAX_CONT ? n_cont - base : MAX_CONT;
        HistArgs a;
        a.n_cont = nc;
        a.fused = fused ? 1 : 0;
        a.n_bins = n_bins;
        if (grid) a.grid = *grid; else a.grid = outb;
        a.outb = outb;
        a.prob[0] = prob_nu;
        a.prob[1] = prob_nubar;
        a.slab_base = 0;
        int64_t nev[MAX_CONT];
        for (int c = 0; c < nc; c++) { a.cont[c] = conts[base + c]; nev[c] = conts[base + c].n; }
        int nblocks = plan_blocks(nev, nc, a.chunk, a.blk_start);
        if (nblocks > MAX_SLAB_ROWS) return PISA_HIP_ERR_INVALID;
        ReduceArgs r;
        r.n_cont = nc;
        r.from_global = lds ? 0 : 1;
        r.n_entries = n_bins * 2;
        for (int c = 0; c <= nc; c++) r.blk_start[c] = a.blk_start[c];
        r.slab_base = 0;
        r.cont_base = base;
        if (!lds) PISA_TRY_HIP(hipMemsetAsync(ws, 0, (size_t)nc * row_bytes, s));
        if (nblocks > 0) {
            dim3 grid_dim((unsigned)nblocks), block(HIST_THREADS);
            size_t shmem = lds ? (size_t)row_bytes : 0;
            if (g_prof_start) PISA_TRY_HIP(hipEventRecord(g_prof_start, s));
            if (fused) {
                if (lds) hipLaunchKernelGGL((hist_accumulate_kernel<true, true>), grid_dim, block, shmem, s, a, ws, d_status);
                else hipLaunchKernelGGL((hist_accumulate_kernel<true, false>), grid_dim, block, shmem, s, a, ws, d_status);
            } else {
                if (lds) hipLaunchKernelGGL((hist_accumulate_kernel<false, true>), grid_dim, block, shmem, s, a, ws, d_status);
                else hipLaunchKernelGGL((hist_accumulate_kernel<false, false>), grid_dim, block, shmem, s, a, ws, d_status);
            }
            PISA_CHECK_LAUNCH("hist_accumulate_kernel");
            if (g_prof_stop) PISA_TRY_HIP(hipEventRecord(g_prof_stop, s));
        }
        dim3 rgrid((unsigned)((r.n_entries + 255) / 256), (unsigned)nc), rblock(256);
        hipLaunchKernelGGL(hist_reduce_kernel, rgrid, rblock, 0, s, r, ws, d_limbs, d_status);
        PISA_CHECK_LAUNCH("hist_reduce_kernel");
    }
    return PISA_HIP_OK;
}

}  // namespace pisa

using namespace pisa;

PISA_API int pisa_hip_profile_events(void *start_event, void *stop_event) {
    g_prof_start = reinterpret_cast<hipEvent_t>(start_event);
    g_prof_stop = reinterpret_cast<hipEvent_t>(stop_event);
    return PISA_HIP_OK;
}

PISA_API int64_t pisa_hip_hist_workspace_bytes(int32_t n_containers, int64_t n_bins) {
    if (n_containers < 1 || n_bins < 1) return PISA_HIP_ERR_INVALID;
    int64_t row = acc_entries_bytes(n_bins);
    if (row <= LDS_ACC_BYTES_MAX) return MAX_SLAB_ROWS * row;
    int nc = n_containers < MAX_CONT ? n_containers : MAX_CONT;
    return (int64_t)nc * row;
}

PISA_API int pisa_hip_reweight_hist(const pisa_hip_container *h_containers, int32_t n_containers,
                                    const pisa_hip_binning *h_calc_grid, const double *d_prob_nu,
                                    const double *d_prob_nubar,
                                    const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                                    void *d_workspace, int32_t *d_status, void *stream) {
    if (!h_containers || n_containers < 1 || n_containers > 1024 || !d_limbs || !d_workspace)
        return PISA_HIP_ERR_INVALID;
    DevBinning grid, outb;
    int64_t n_nodes, n_bins;
    int rc = make_dev_binning(h_calc_grid, grid, n_nodes);
    if (rc) return rc;
    if (grid.ndim > 2) return PISA_HIP_ERR_INVALID;
    if ((rc = make_dev_binning(h_out_binning, outb, n_bins))) return rc;
    ContDev *conts = new ContDev[n_containers];
    for (int c = 0; c < n_containers; c++) {
        const pisa_hip_container &h = h_containers[c];
        ContDev &d = conts[c];
        bool bad = h.n_events < 0 || h.flav < 0 || h.flav > 2 || (h.nubar != 1 && h.nubar != -1);
        if (h.n_events > 0) {
            bad = bad || !h.d_grid_x || (grid.ndim > 1 && !h.d_grid_y) || !h.d_nu_flux ||
                  !h.d_weighted_aeff || !h.d_initial_weights;
            for (int k = 0; k < outb.ndim; k++) bad = bad || !h.d_sample[k];
            bad = bad || (h.nubar > 0 ? !d_prob_nu : !d_prob_nubar);
        }
        if (bad) { delete[] conts; return PISA_HIP_ERR_INVALID; }
        d.n = h.n_events;
        d.gx = h.d_grid_x; d.gy = h.d_grid_y; d.flux = h.d_nu_flux; d.aeff = h.d_weighted_aeff;
        d.w0 = h.d_initial_weights;
        for (int k = 0; k < 3; k++) d.s[k] = h.d_sample[k];
        d.scale = h.scale;
        d.flav = h.flav;
        d.side = h.nubar > 0 ? 0 : 1;
    }
    rc = run_hist(conts, n_containers, true, &grid, d_prob_nu, d_prob_nubar, outb, n_bins,
                  (long long *)d_limbs, d_workspace, d_status, as_stream(stream));
    delete[] conts;
    return rc;
}

PISA_API int pisa_hip_hist_finalize(const int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                                    double *d_hist, double *d_sumw2, void *stream) {
    if (!d_limbs || n_containers < 1 || n_bins < 1) return PISA_HIP_ERR_INVALID;
    int64_t total = (int64_t)n_containers * n_bins;
    dim3 block(256), grid((unsigned)((total + 255) / 256));
    hipLaunchKernelGGL(hist_finalize_kernel, grid, block, 0, as_stream(stream),
                       (const long long *)d_limbs, total, d_hist, d_sumw2);
    PISA_CHECK_LAUNCH("hist_finalize_kernel");
    return PISA_HIP_OK;
}

// Generic histogram: needs scratch; allocated per call (setup-time use only:
// container translations, hist_transform construction), not on the hot loop.
PISA_API int pisa_hip_histogram_regular(const pisa_hip_binning *h_binning,
                                        const double *const *h_d_sample, int64_t n,
                                        const double *d_weights, int32_t averaged, double *d_hist,
                                        void *stream) {
    DevBinning outb;
    int64_t n_bins;
    int rc = make_dev_binning(h_binning, outb, n_bins);
    if (rc) return rc;
    if (n < 0 || !d_hist || !h_d_sample) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < outb.ndim; k++)
        if (n > 0 && !h_d_sample[k]) return PISA_HIP_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    int64_t ws_bytes = pisa_hip_hist_workspace_bytes(1, n_bins);
    int64_t limb_bytes = n_bins * 2 * NL * 8;
    void *ws = nullptr;
    long long *limbs = nullptr;
    double *cnt = nullptr;
    int32_t *st = nullptr;
    PISA_TRY_HIP(hipMalloc(&ws, (size_t)ws_bytes));
    rc = check_hip(hipMalloc(&limbs, (size_t)limb_bytes), "hipMalloc");
    if (!rc) rc = check_hip(hipMalloc(&cnt, (size_t)n_bins * 8), "hipMalloc");
    if (!rc) rc = check_hip(hipMalloc(&st, 4), "hipMalloc");
    if (!rc) rc = check_hip(hipMemsetAsync(st, 0, 4, s), "memset");
    if (!rc) {
        ContDev c;
        c.n = n; c.gx = c.gy = c.flux = c.aeff = nullptr; c.w0 = d_weights;
        for (int k = 0; k < 3; k++) c.s[k] = k < outb.ndim ? h_d_sample[k] : nullptr;
        c.scale = 1.0; c.flav = 0; c.side = 0;
        rc = run_hist(&c, 1, false, nullptr, nullptr, nullptr, outb, n_bins, limbs, ws, st, s);
    }
    if (!rc) rc = pisa_hip_hist_finalize((const int64_t *)limbs, 1, n_bins, d_hist, cnt, s);
    if (!rc && averaged) {
        dim3 block(256), grid((unsigned)((n_bins + 255) / 256));
        hipLaunchKernelGGL(hist_average_kernel, grid, block, 0, s, n_bins, d_hist, cnt);
        rc = check_hip(hipGetLastError(), "hist_average_kernel");
    }
    int32_t h_st = 0;
    if (!rc) rc = check_hip(hipMemcpyAsync(&h_st, st, 4, hipMemcpyDeviceToHost, s), "d2h");
    if (!rc) rc = check_hip(hipStreamSynchronize(s), "sync");
    if (ws) (void)hipFree(ws);
    if (limbs) (void)hipFree(limbs);
    if (cnt) (void)hipFree(cnt);
    if (st) (void)hipFree(st);
    if (!rc && h_st) rc = PISA_HIP_ERR_OVERFLOW;
    return rc;
}

PISA_API int pisa_hip_lookup_regular(const pisa_hip_binning *h_binning,
                                     const double *const *h_d_sample, int64_t n,
                                     const double *d_flat_hist, int32_t width, double *d_out,
                                     void *stream) {
    DevBinning b;
    int64_t n_bins;
    int rc = make_dev_binning(h_binning, b, n_bins);
    if (rc) return rc;
    if (n < 0 || width < 1 || !h_d_sample) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_flat_hist || !d_out) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < b.ndim; k++)
        if (!h_d_sample[k]) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(lookup_regular_kernel, grid, block, 0, as_stream(stream), b, h_d_sample[0],
                       b.ndim > 1 ? h_d_sample[1] : nullptr, b.ndim > 2 ? h_d_sample[2] : nullptr,
                       n, d_flat_hist, (int)width, d_out);
    PISA_CHECK_LAUNCH("lookup_regular_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_apply_osc_weights(const double *d_nu_flux, const double *d_prob_e,
                                        const double *d_prob_mu, int64_t n, double *d_weights,
                                        void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_nu_flux || !d_prob_e || !d_prob_mu || !d_weights) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(apply_osc_weights_kernel, grid, block, 0, as_stream(stream), d_nu_flux,
                       d_prob_e, d_prob_mu, n, d_weights);
    PISA_CHECK_LAUNCH("apply_osc_weights_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_apply_aeff(const double *d_weighted_aeff, double scale, int64_t n,
                                 double *d_weights, void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_weighted_aeff || !d_weights) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(apply_aeff_kernel, grid, block, 0, as_stream(stream), d_weighted_aeff,
                       scale, n, d_weights);
    PISA_CHECK_LAUNCH("apply_aeff_kernel");
    return PISA_HIP_OK;
}
