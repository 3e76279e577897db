/*
 * pisa_hip.h -- C ABI of libpisa_hip.so, the MI355X (gfx950) implementation of
 * PISA's per-event hot path:  prob3 oscillation -> flux x osc x aeff reweight
 * -> weighted N-D histogram -> LLH / chi2.
 *
 * Conventions
 *   - every entry point returns an int status: 0 = OK, negative = error
 *     (pisa_hip_strerror() gives the text).  No exceptions cross the ABI.
 *   - `d_` pointers are DEVICE pointers (hipMalloc'ed or torch-allocated
 *     tensor.data_ptr()); `h_` pointers are host pointers.  Small parameter
 *     blocks are passed by host pointer and travel in the kernel-argument
 *     segment.
 *   - `stream` is a hipStream_t cast to void* (NULL = default stream).  All
 *     calls are asynchronous w.r.t. the host unless the name ends in _host.
 *   - fp64 throughout (PISA FTYPE=float64); complex numbers are interleaved
 *     (re, im) doubles, matrices are row-major 3x3 -- i.e. exactly the memory
 *     layout of the numpy arrays the reference passes to its numba kernels.
 *   - the library is stateless apart from an optional scratch context.
 *
 * Each declaration names the reference interface it replaces (file:line
 * relative to the icecube/pisa tree).
 */
#ifndef PISA_HIP_H
#define PISA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PISA_HIP_OK 0
#define PISA_HIP_ERR_INVALID -1     /* bad argument (shape, NULL, range)            */
#define PISA_HIP_ERR_LAYERS -2      /* >120 layers (numba_osc_kernels.py:227)       */
#define PISA_HIP_ERR_HIP -3         /* HIP runtime error (see pisa_hip_last_hip_error) */
#define PISA_HIP_ERR_GEOMETRY -4    /* Earth model/detector geometry unsupported    */
#define PISA_HIP_ERR_NEGATIVE -5    /* negative counts passed to a metric (stats.py:231-240) */
#define PISA_HIP_ERR_OVERFLOW -6    /* weight outside the fixed-point accumulator range */
#define PISA_HIP_ERR_NOMEM -7

#define PISA_HIP_MAX_LAYERS 120     /* numba_osc_kernels.py:227 */
#define PISA_HIP_MAX_SHELLS 64      /* PREM-59 + atmosphere = 61 */
#define PISA_HIP_MAX_DIMS 3         /* translation.py:252 "can only do up to 3D" */
#define PISA_HIP_ACC_LIMBS 6        /* 6 x 32-bit payload limbs, see DESIGN.md */
#define PISA_HIP_MAX_POINTS 16      /* parameter points per multi-point call */

const char *pisa_hip_strerror(int status);
const char *pisa_hip_last_hip_error(void);
int pisa_hip_version(void);
/* number of visible GPUs (<=0: none). Does not initialise a device context. */
int pisa_hip_device_count(void);

/* ------------------------------------------------------------------ prob3 */

/* The scalar arguments of `propagate_array`
 * (pisa/stages/osc/prob3numba/numba_osc_hostfuncs.py:56-70), as prepared by
 * prob3.compute_function / calc_probs (pisa/stages/osc/prob3.py:429-450, 539-578). */
typedef struct {
    double dm[9];         /* dm_matrix  f8[3,3]   (osc_params.py:265-292)        */
    double mix[18];       /* PMNS       c16[3,3]  (osc_params.py:174-211)        */
    double mat_pot[18];   /* generalised matter potential c16[3,3] (prob3.py:539-557) */
    double mat_decay[18]; /* decay matrix c16[3,3] (prob3.py:559-563)            */
    double lri_pot[9];    /* LRI potential f8[3,3] (prob3.py:565-578)            */
    int64_t decay_flag;   /* +1 decay on, -1 off  (prob3.py:226-229)             */
} pisa_hip_prob3_params;

/* Replaces the gufunc `propagate_array` for one container
 * (numba_osc_hostfuncs.py:56-70 -> osc_probs_layers_kernel,
 * numba_osc_kernels.py:121-345).
 *   d_energy[n]; d_densities / d_distances: [n][n_layers] if
 *   layers_per_element != 0, else one shared row [n_layers];
 *   d_probability[n][3][3] (P[init][final]).
 * nubar = +1 / -1 (container aux datum). */
int pisa_hip_propagate_array(const pisa_hip_prob3_params *h_params, int64_t nubar,
                             const double *d_energy, const double *d_densities,
                             const double *d_distances, int64_t n, int32_t n_layers,
                             int32_t layers_per_element, double *d_probability, void *stream);

/* Same contract with HOST buffers (numpy arrays): allocates, copies, runs,
 * copies back, synchronises.  This is the call a reference-side binding would
 * use in place of the numba gufunc. */
int pisa_hip_propagate_array_host(const pisa_hip_prob3_params *h_params, int64_t nubar,
                                  const double *h_energy, const double *h_densities,
                                  const double *h_distances, int64_t n, int32_t n_layers,
                                  int32_t layers_per_element, double *h_probability);

/* Grid fast path used when calc_mode is a 2-D (true_energy x true_coszen)
 * binning (prob3.py:452-459 links the 12 containers into 'nu' and 'nubar'):
 * one launch evaluates nu AND nubar on every node.
 *   d_energy[n_e]       node energies along the energy axis
 *   d_densities/d_distances[n_cz][n_layers]   one row per coszen node
 *   node index = e_major ? iE*n_cz + jcz : jcz*n_e + iE   (container.py:769-773)
 *   d_prob_nu / d_prob_nubar [n_e*n_cz][3][3]  (either may be NULL)
 *   d_pepmu [2][3][n_e*n_cz][2] (may be NULL): compact gather tables
 *       pepmu[side][flav][node] = (P[e->flav], P[mu->flav]), side 0 nu / 1 nubar,
 *       i.e. prob_e / prob_mu of `fill_probs` (prob3.py:593-608) for every
 *       container class, laid out for one 16-byte gather per event. */
int pisa_hip_prob3_grid(const pisa_hip_prob3_params *h_params, const double *d_energy,
                        int32_t n_e, const double *d_densities, const double *d_distances,
                        int32_t n_cz, int32_t n_layers, int32_t e_major, double *d_prob_nu,
                        double *d_prob_nubar, double *d_pepmu, void *stream);

/* Planned form of the same computation.  A plan is built once per set of
 * layer rows (i.e. at setup, and again only when Ye / tomography parameters
 * change, prob3.py:461-475): it resolves the reference's layer-matrix cache
 * (numba_osc_kernels.py:230-249) per coszen row, gives mirrored layers of a row one
 * matrix and lists the distinct shell densities.  Per evaluation H(E, rho) is then
 * diagonalised per (energy, distinct density) instead of per node and layer, and
 * each row's chain is multiplied in parts.  Results equal pisa_hip_prob3_grid to
 * rounding (<= 3e-13 absolute on the probabilities), not bit for bit.
 * plan_create synchronises (it reads the rows back); the planned call is async. */
typedef struct pisa_hip_grid_plan pisa_hip_grid_plan;
int pisa_hip_grid_plan_create(const double *d_densities, const double *d_distances, int32_t n_cz,
                              int32_t n_layers, pisa_hip_grid_plan **out);
int pisa_hip_grid_plan_destroy(pisa_hip_grid_plan *plan);
int pisa_hip_prob3_grid_planned(const pisa_hip_prob3_params *h_params, pisa_hip_grid_plan *plan,
                                const double *d_energy, int32_t n_e, int32_t e_major,
                                double *d_prob_nu, double *d_prob_nubar, double *d_pepmu,
                                void *stream);

/* Earth model as held by `Layers` (pisa/stages/osc/layers.py:216-335, 411-439):
 * shells ordered from the production sphere inwards. */
typedef struct {
    int32_t n_shell;
    double r_detector;
    double radii[PISA_HIP_MAX_SHELLS];
    double rhos[PISA_HIP_MAX_SHELLS];         /* electron-fraction weighted */
    double coszen_limit[PISA_HIP_MAX_SHELLS]; /* layers.py:308-335 */
} pisa_hip_earth;

/* Replaces `extCalcLayers` (layers.py:38-169).  Outputs [n][max_layers],
 * zero padded; max_layers >= 2*n_shell (layers.py:244).  d_n_layers may be
 * NULL.  d_status (int32, may be NULL) is set non-zero if any coszen hit the
 * geometry the reference cannot handle (it raises a broadcast error there). */
int pisa_hip_calc_layers(const pisa_hip_earth *h_earth, const double *d_coszen, int64_t n,
                         int32_t max_layers, double *d_n_layers, double *d_densities,
                         double *d_distances, int32_t *d_status, void *stream);

/* Event-by-event prob3 (calc_mode = "events", prob3.py:406-409 + 581-588)
 * WITHOUT materialising densities/distances[n][L]: the layer path of every
 * event is rebuilt in-kernel from its coszen with the shell table in LDS. */
int pisa_hip_prob3_events(const pisa_hip_prob3_params *h_params, const pisa_hip_earth *h_earth,
                          int64_t nubar, const double *d_energy, const double *d_coszen,
                          int64_t n, double *d_probability, int32_t *d_status, void *stream);

/* The same for several containers in ONE launch (prob3.py:581-588 loops over the
 * 12 containers), optionally writing only the two probabilities the container
 * needs: d_pepmu[n][2] = (P[e->flav], P[mu->flav]) = (prob_e, prob_mu) of
 * `fill_probs` (prob3.py:593-608). */
typedef struct {
    int64_t n_events;
    const double *d_energy;
    const double *d_coszen;
    double *d_probability;   /* [n][3][3] or NULL */
    double *d_pepmu;         /* [n][2] or NULL    */
    int32_t nubar;           /* +1 / -1 */
    int32_t flav;            /* 0, 1, 2 */
} pisa_hip_event_set;
int pisa_hip_prob3_events_multi(const pisa_hip_prob3_params *h_params,
                                const pisa_hip_earth *h_earth, const pisa_hip_event_set *h_sets,
                                int32_t n_sets, int32_t *d_status, void *stream);

/* `fill_probs` (numba_osc_hostfuncs.py:206-221): out[i] = P[i][init_flav][flav]. */
int pisa_hip_fill_probs(const double *d_probability, int64_t init_flav, int64_t flav, int64_t n,
                        double *d_out, void *stream);

/* ------------------------------------------------------------ translation */

typedef struct {
    int32_t ndim;                         /* 1..3 */
    int64_t nbins[PISA_HIP_MAX_DIMS];
    double mins[PISA_HIP_MAX_DIMS];       /* regularised (linear) domain:      */
    double maxs[PISA_HIP_MAX_DIMS];       /* ln(lo), ln(hi) for log dimensions */
} pisa_hip_binning;

/* `lookup_regular_{1,2,3}d` and `_array` variants
 * (pisa/core/translation.py:417-501): nearest-bin gather, 0 outside
 * [min,max).  d_flat_hist[n_bins][width], d_out[n][width]. */
int pisa_hip_lookup_regular(const pisa_hip_binning *h_binning, const double *const *h_d_sample,
                            int64_t n, const double *d_flat_hist, int32_t width, double *d_out,
                            void *stream);

/* `histogram(sample, weights, binning, averaged)` for regular linear
 * binnings (translation.py:90-129, 171-205 -> fast_histogram.histogramdd).
 * d_weights may be NULL (counts).  Sums are accumulated in 192-bit fixed
 * point (order independent, bit-reproducible) and rounded once to fp64.
 * averaged != 0 divides by the per-bin count with NaN->0 (translation.py:118-127). */
int pisa_hip_histogram_regular(const pisa_hip_binning *h_binning,
                               const double *const *h_d_sample, int64_t n,
                               const double *d_weights, int32_t averaged, double *d_hist,
                               void *stream);

/* ------------------------------------- fused reweight + histogram (hot loop) */

/* Digitise event coordinates once: d_index[i] = flat C-order bin of the regular
 * binning, or -1 outside [min,max) / NaN (same rule as the lookups and
 * histograms above).  Event coordinates do not change between evaluations, so
 * the engine stores the calc-grid node and the output bin of every event as
 * int32 columns (the reference pre-digitises irregular dimensions the same way,
 * utils/hist.py:100-113). */
int pisa_hip_event_indices(const pisa_hip_binning *h_binning, const double *const *h_d_sample,
                           int64_t n, int32_t *d_index, void *stream);

/* One PISA container (pisa/core/container.py:451) as seen by the fused
 * kernel: device columns + the aux scalars the stages read.  Either the
 * coordinate columns (d_grid_*, d_sample) or the pre-digitised index columns
 * (d_node, d_bin) must be given; the indexed form moves 40 B/event, the
 * coordinate form 48 + 8*D B/event. */
typedef struct {
    int64_t n_events;
    const double *d_grid_x;          /* lookup coordinate on calc-grid dim 0 (ln E if log) */
    const double *d_grid_y;          /* lookup coordinate on calc-grid dim 1               */
    const double *d_nu_flux;         /* [n][2]  (nue, numu) flux  (barr_simple.py:100)     */
    const double *d_weighted_aeff;   /* [n]                                              */
    const double *d_initial_weights; /* [n]   (toy_event_generator.py:101-104)           */
    const double *d_sample[PISA_HIP_MAX_DIMS]; /* output-binning coordinates, regularised */
    const int32_t *d_node;           /* [n] calc-grid node of each event, -1 outside (optional) */
    const int32_t *d_bin;            /* [n] output bin of each event, -1 outside (optional)     */
    const int32_t *d_node_bin;       /* [n][2] the two above interleaved (optional, fastest)    */
    const double *d_aeff_w0;         /* [n][2] (weighted_aeff, initial_weights) interleaved (opt.) */
    const double *d_pepmu;           /* [n_nodes][2] this container's own (P_e, P_mu) table,
                                        overrides the grid tables (event-mode prob3: node = event) */
    int32_t flav;                    /* 0 e, 1 mu, 2 tau  (aux 'flav')                     */
    int32_t nubar;                   /* +1 / -1           (aux 'nubar')                    */
    double scale;                    /* aeff_scale*livetime_s*norms (aeff.py:78-86)        */
    const double *d_weighted_flux;   /* [n][2] optional COMPACT form, used together with d_node_bin:
                                        initial_weights*weighted_aeff*(nu_flux_e, nu_flux_mu), i.e. the
                                        factors of the weight that do not depend on the oscillation
                                        parameters multiplied once (24 B per event instead of 40).
                                        w = ((g_e*P_e) + (g_mu*P_mu)) * scale: the reference's product
                                        with the static factors associated first (differs from the
                                        40-B form by rounding, <= 3 ulp per weight).  The caller
                                        refreshes it when nu_flux changes (flux systematics). */
    /* 16-BIT INDEX form of the compact columns (20 B per event), for calc grids below 65535
     * nodes and output binnings below 65535 bins.
     * Given for every container it is the form used; it stands alone (d_node_bin and
     * d_weighted_flux may be NULL).  For a grid / binning it does not apply to these two
     * columns are ignored: the call uses the other forms if they are given as well and is
     * PISA_HIP_ERR_INVALID if not.  Both arrays are padded to a multiple of 256 events with events
     * outside the binning (index word 0xffffffff, flux pair 0). */
    const uint32_t *d_node_bin16;    /* [n_pad] node | bin << 16, 0xffff in a half = outside */
    const double *d_weighted_flux_q; /* [n_pad / 256][4][64][2]: d_weighted_flux with the pairs of a
                                        quad of consecutive events e = 4q + k stored at
                                        [q / 64][k][q % 64], so that the 64 lanes of a wavefront
                                        (one quad each) read 16 contiguous bytes per lane */
    /* PARTITIONED resident order (optional; 16-bit index form with an output binning beyond the LDS
     * accumulators, pisa_hip_hist_window_bins(n_bins) = W > 0): the caller has ordered the events so that
     * partition p = events [256 d_part_start[p], 256 d_part_start[p+1]) deposits only into bins
     * [p W, (p+1) W) (events outside the binning may sit anywhere), d_part_start[0] = 0,
     * d_part_start[n_part] = n_pad / 256.  The kernel then keeps every deposit in LDS and scans nothing.
     * Ignored (the general window path is used) unless part_width == W.  The histogram does not depend on
     * the order of the events: the same bits either way. */
    const int32_t *d_part_start;     /* DEVICE array [n_part + 1], units of 256 events, or NULL */
    int32_t n_part;
    int32_t part_width;
} pisa_hip_container;

/* 0 if the fused kernels keep all n_bins accumulators of a container in LDS, else the number of
 * consecutive bins W their LDS window holds (a multiple of 32). */
int pisa_hip_hist_window_bins(int64_t n_bins);
/* Workgroups the fused kernels give each container of a launch on the current device (a hint for a caller
 * that lays out the partitioned order: partitions that are whole numbers of a container's per-workgroup share
 * keep every workgroup inside one partition; any other layout is handled, only slower). */
int pisa_hip_hist_workgroups(const int64_t *h_n_events, int32_t n_containers, int32_t *h_workgroups);
/* The resident order of one container's events for the 16-bit index form (d_node_bin16 / d_weighted_flux_q): the
 * permutation that gathers the events that can deposit (d_node >= 0 and d_bin >= 0: the digitised calc-grid node and output
 * bin of every event, DEVICE int32[n]) into whole blocks of 256 -- sorted by node, dealt over the 32 LDS bank pairs inside
 * windows of 4 096, the blocks spread evenly among the blocks of events that deposit nothing.  No counterpart in the
 * reference (its events stay in file order; `container.py:981-1012` looks them up one by one): the sums are exact, so the
 * order is this build's to choose, and it is part of the set-up a caller pays once per event sample.
 * d_perm: DEVICE int64[n], out: position i of the resident order holds input event d_perm[i].  n_nodes: size of the calc
 * grid (d_node < n_nodes).  Asynchronous on `stream`; d_work: DEVICE scratch of pisa_hip_deposit_block_order_workspace(n)
 * bytes. */
int64_t pisa_hip_deposit_block_order_workspace(int64_t n);
int pisa_hip_deposit_block_order(const int32_t *d_node, const int32_t *d_bin, int64_t n, int64_t n_nodes,
                                 int64_t *d_perm, void *d_work, int64_t work_bytes, void *stream);
/* The resident order for a binning BEYOND the LDS accumulators (pisa_hip_hist_window_bins(n_bins) = W > 0): partition p holds
 * the events that deposit into bins [p W, (p + 1) W) -- sorted by node, whole 4 096-event windows of a partition of >= 8 192
 * events in the bank order --, topped up with idle events to whole blocks of 256 and interleaved with further idle blocks
 * (pisa_hip_container::d_part_start is the table the fused kernel then walks).  Two calls around the caller's block accounting:
 *   _sort      one key per event, one stable radix sort; h_counts[0 .. n_part) = depositing events of every partition,
 *              h_counts[n_part] = idle events (HOST, out; the call synchronises `stream`);
 *   _assemble  h_n_dep = those counts, h_dep_blocks[p] = ceil(h_n_dep[p] / 256), h_idle_blocks[p] = idle blocks partition p
 *              is interleaved with (depositing block q of its nb goes to block floor(q (nb + nf) / nb) of the partition):
 *              d_perm DEVICE int64[n], out, position i of the resident order holds input event d_perm[i]; whatever idle
 *              events the blocks do not use follow the last partition.  Synchronises `stream`.
 * Both take the SAME d_work (DEVICE, pisa_hip_partition_order_workspace(n) bytes: the sorted sequence lives in it between the
 * calls).  n_part <= 255, (n_part + 1) (n_nodes + 1) < 2^32.  No counterpart in the reference (its events stay in file
 * order): the sums are exact, so the order is this build's to choose. */
int64_t pisa_hip_partition_order_workspace(int64_t n);
int pisa_hip_partition_order_sort(const int32_t *d_node, const int32_t *d_bin, int64_t n, int64_t n_nodes, int32_t width,
                                  int32_t n_part, int64_t *h_counts, void *d_work, int64_t work_bytes, void *stream);
int pisa_hip_partition_order_assemble(const int32_t *d_bin, int64_t n, int32_t n_part, const int64_t *h_n_dep,
                                      const int64_t *h_dep_blocks, const int64_t *h_idle_blocks, int64_t *d_perm,
                                      void *d_work, int64_t work_bytes, void *stream);
/* The resident copies of one container's event columns in the order `d_perm` (the permutation above), in ONE launch: the
 * permuted columns themselves and the interleaved / folded forms the fused kernel reads -- what pisa_amd/engine.py built
 * with ~25 tensor operations per container (6 ms of a 24 ms set-up at 1e7 events, bound by their dispatch on the host).
 * out[k] = in[d_perm[k]] for every column; d_node_bin[k] = (node, bin); d_aeff_w0[k] = (aeff, w0); d_static_w[k] =
 * w0 * aeff; d_node_bin16[k] = node | bin << 16 with 0xFFFF for a negative index, 0xFFFFFFFF for n <= k < n_pad (n_pad = n
 * rounded up to whole blocks of 256 events).  All pointers DEVICE; columns double[n], d_nu_flux double[n][2], d_node /
 * d_bin int32[n], d_perm int64[n]; d_sample: up to three output-binning coordinate columns (n_sample of them).  No
 * counterpart in the reference (its containers keep file order, `container.py:933-1012`).  Asynchronous on `stream`. */
typedef struct {
    int64_t n, n_pad;
    const int64_t *d_perm;
    const double *d_grid_x, *d_grid_y, *d_nu_flux, *d_weighted_aeff, *d_initial_weights;
    const double *d_sample[3];
    const int32_t *d_node, *d_bin;
    double *o_grid_x, *o_grid_y, *o_nu_flux, *o_weighted_aeff, *o_initial_weights;
    double *o_sample[3];
    int32_t *o_node, *o_bin, *o_node_bin;
    double *o_aeff_w0, *o_static_w;
    int32_t *o_node_bin16;
    int32_t n_sample, reserved;
} pisa_hip_pack_set;
int pisa_hip_pack_resident_columns(const pisa_hip_pack_set *set, void *stream);

/* Fused  prob3.apply (prob3.py:621-622, with the grid->event lookup of
 * container.py:981-1012 / translation.py:427-438)  +  aeff.apply
 * (aeff.py:78-88)  +  hist.apply with error_method='sumw2'
 * (utils/hist.py:163-218)  over all containers in ONE pass over the events:
 *     w  = w0 * (f_e*P[e->flav] + f_mu*P[mu->flav]) * (aeff*scale)
 *     hist[c][bin] += w ;  sumw2[c][bin] += w*w
 * Result: d_limbs[n_containers][n_bins][2][PISA_HIP_ACC_LIMBS] int64 (zeroed by
 * the call) -- exact sums in 192-bit fixed point spread over six 32-bit-spaced
 * limbs, safe to SUM-all-reduce across ranks (integer addition is associative
 * => bit-reproducible for any workgroup schedule and any GPU count).
 * d_pepmu: gather tables of pisa_hip_prob3_grid (required for the indexed forms; the coordinate
 * form reads them too when they are given -- one 16-byte gather per event instead of two 8-byte ones
 * from the 72-byte records of d_prob_nu / d_prob_nubar, 25 % of that kernel -- and the full matrices
 * otherwise; both hold the same numbers).
 * d_status: int32 flag, set non-zero if a weight is not finite or >= 2^76. */
int pisa_hip_reweight_hist(const pisa_hip_container *h_containers, int32_t n_containers,
                           const pisa_hip_binning *h_calc_grid, const double *d_prob_nu,
                           const double *d_prob_nubar, const double *d_pepmu,
                           const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                           int32_t *d_status, void *stream);

/* Measurement hook: hipEvent_t handles (cast to void*, NULL to disable) that the
 * next pisa_hip_reweight_hist / pisa_hip_histogram_regular calls of this host
 * thread record immediately before and after the accumulate kernel on `stream`. */
int pisa_hip_profile_events(void *start_event, void *stop_event);

/* Unfused stage-by-stage variants (same arithmetic, one stage each) so that a
 * pipeline with other services interleaved still runs on the device. */
int pisa_hip_apply_osc_weights(const double *d_nu_flux, const double *d_prob_e,
                               const double *d_prob_mu, int64_t n, double *d_weights,
                               void *stream); /* prob3.py:621-622 */
/* the same with prob_e / prob_mu read at an element stride (columns of one table: the (P_e, P_mu) pairs of the
 * gather tables have stride 2) -- a stage that publishes such columns as views need not compact them */
int pisa_hip_apply_osc_weights_strided(const double *d_nu_flux, const double *d_prob_e,
                                       const double *d_prob_mu, int64_t prob_stride, int64_t n,
                                       double *d_weights, void *stream); /* prob3.py:621-622 */
int pisa_hip_apply_aeff(const double *d_weighted_aeff, double scale, int64_t n,
                        double *d_weights, void *stream); /* aeff.py:87 */

/* The whole weight chain of several containers in ONE launch (what the three stages' apply_functions do to
 * `weights` one after the other, on events or on the bins of a map):
 *   weights = copy(initial_weights)                   toy_event_generator.py:101-104 and the other loaders
 *   weights *= flux[:,0]*prob_e + flux[:,1]*prob_mu   prob3.py:621-622   (skipped if d_nu_flux == NULL)
 *   weights *= weighted_aeff * aeff_scale             aeff.py:87         (skipped if d_weighted_aeff == NULL)
 * with every step rounded to fp64 as the one-step calls above round it: the same bits as
 * copy -> pisa_hip_apply_osc_weights[_strided] -> pisa_hip_apply_aeff. */
typedef struct {
    int64_t n;
    const double *d_initial_weights; /* [n] */
    const double *d_nu_flux;         /* [n][2] or NULL */
    const double *d_prob_e;          /* read at element stride prob_stride */
    const double *d_prob_mu;
    int64_t prob_stride;
    const double *d_weighted_aeff;   /* [n] or NULL */
    double aeff_scale;
    double *d_weights;               /* [n] out */
} pisa_hip_chain_set;
int pisa_hip_weight_chain_multi(const pisa_hip_chain_set *h_sets, int32_t n_sets, void *stream);

/* Converts all-reduced limbs to fp64 maps: d_hist / d_sumw2 [n_containers][n_bins]
 * (either may be NULL). `errors` = sqrt(sumw2) is left to the caller (hist.py:215). */
int pisa_hip_hist_finalize(const int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                           double *d_hist, double *d_sumw2, int32_t *d_status, void *stream);

/* pisa_hip_reweight_hist without the initial clear of d_limbs: ADDS this call's
 * events to whatever the limbs hold (e.g. the zeros left behind by
 * pisa_hip_finalize_metric(clear_limbs=1), or the sums of another batch). */
int pisa_hip_reweight_hist_acc(const pisa_hip_container *h_containers, int32_t n_containers,
                               const pisa_hip_binning *h_calc_grid, const double *d_prob_nu,
                               const double *d_prob_nubar, const double *d_pepmu,
                               const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                               int32_t *d_status, void *stream);

/* Largest n_containers*n_bins pisa_hip_finalize_metric accepts (one workgroup). */
#define PISA_HIP_FINALIZE_METRIC_MAX 4096

/* pisa_hip_hist_finalize + pisa_hip_metric(kind, d_actual, d_hist, d_sumw2,
 * n_maps = n_containers) in ONE launch, for the tail of a template evaluation
 * (hist.py:215 -> distribution_maker.py:274-281 -> map.py:1572-1604): same
 * arithmetic in the same order as the two separate calls, bit for bit.
 * d_hist / d_sumw2 [n_containers][n_bins] are written (required).
 * total[1] may be device memory or device-mapped pinned host memory (the host
 * then only waits for the stream; no copy kernel).
 * clear_limbs != 0: d_limbs is zero again when the kernel ends, ready for
 * pisa_hip_reweight_hist_acc.
 * d_status: overflow flag of the limbs (as pisa_hip_hist_finalize);
 * d_metric_status: PISA_HIP_ERR_NEGATIVE for negative inputs (as pisa_hip_metric).
 * PISA_HIP_ERR_INVALID if n_containers*n_bins > PISA_HIP_FINALIZE_METRIC_MAX. */
int pisa_hip_finalize_metric(int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                             double *d_hist, double *d_sumw2, int32_t kind,
                             const double *d_actual, double *total, int32_t *d_status,
                             int32_t *d_metric_status, int32_t clear_limbs, void *stream);

/* pisa_hip_finalize_metric for a template that is not just the sum of the histogrammed maps:
 * d_scale [n_containers][n_bins] (or NULL): per-bin factors of a stage that follows the histogram --
 *   discr_sys.hypersurfaces, `weights = clip(weights * s, 0, inf)`, `errors *= s`
 *   (hypersurfaces.py:251-259): the expectation of bin b is sum_c max(w_cb s_cb, 0), its variance
 *   sum_c sumw2_cb s_cb^2.  d_hist / d_sumw2 are written UNscaled.
 * d_extra [2][n_bins] (or NULL): expectation and variance of maps added to the template after the
 *   containers, in that order (the other pipelines of a DistributionMaker, distribution_maker.py:274-281). */
int pisa_hip_finalize_metric_scaled(int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                                    double *d_hist, double *d_sumw2, int32_t kind,
                                    const double *d_actual, const double *d_scale,
                                    const double *d_extra, double *total, int32_t *d_status,
                                    int32_t *d_metric_status, int32_t clear_limbs, void *stream);

/* ------------------------------------------------ several parameter points in one sweep
 * What a fit loop asks of the path: the minimiser settings of the reference are finite-difference
 * methods (settings/minimizer/l-bfgs-b_*.json, slsqp_*.json, driven from
 * pisa/analysis/analysis.py:2493-2670), i.e. n + 1 INDEPENDENT parameter points per gradient, and its
 * own timing protocol evaluates independent points as well
 * (pisa/scripts/benchmark_pipeline_performance.py:196-223).  The three calls below evaluate K points
 * where the reference -- and the single-point calls above -- would run K whole template evaluations
 * one after the other; per point the results are BIT-IDENTICAL to the single-point calls
 * (tests/test_gpu_multipoint.py). */

/* pisa_hip_prob3_grid_planned (osc.prob3 compute_function, prob3.py:581-608) for n_points parameter
 * blocks h_params[n_points] in one pair of launches.  Only the gather tables are written, the points
 * interleaved:  d_pepmu_points[2][3][n_e*n_cz][n_points][2]  (n_points == 1: the layout of d_pepmu).
 * All points must agree in decay_flag.  n_points <= PISA_HIP_MAX_POINTS. */
int pisa_hip_prob3_grid_planned_multi(const pisa_hip_prob3_params *h_params, int32_t n_points,
                                      pisa_hip_grid_plan *plan, const double *d_energy, int32_t n_e,
                                      int32_t e_major, double *d_pepmu_points, void *stream);

/* pisa_hip_reweight_hist (prob3.py:621-622 + aeff.py:78-88 + utils/hist.py:163-218) for n_points
 * parameter points in ONE pass over the event columns: the 20 B per event are read once, an event's
 * (P_e, P_mu) pairs of all points come from one contiguous run of d_pepmu_points (layout above) and
 * each point has its own accumulators.  Containers must carry the 16-bit index form (d_node_bin16,
 * d_weighted_flux_q) and use the shared grid tables (d_pepmu == NULL); PISA_HIP_ERR_INVALID otherwise
 * (the caller then evaluates point by point).
 * h_scales[n_points][n_containers] (or NULL = every point uses container.scale): aeff.py:78-86 scale
 * of each container at each point.
 * d_limbs[n_points][n_containers][n_bins][2][PISA_HIP_ACC_LIMBS]; clear_first != 0: zeroed by the
 * call, else the sums are ADDED to what they hold (the zeros pisa_hip_finalize_metric_multi leaves
 * behind with clear_limbs = 1, as pisa_hip_reweight_hist_acc).
 * A batch larger than pisa_hip_multi_points_per_pass(n_bins) is split into several sweeps. */
int pisa_hip_reweight_hist_multi(const pisa_hip_container *h_containers, int32_t n_containers,
                                 const pisa_hip_binning *h_calc_grid, const double *d_pepmu_points,
                                 int32_t n_points, const double *h_scales,
                                 const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                                 int32_t clear_first, int32_t *d_status, void *stream);
int pisa_hip_multi_points_per_pass(int64_t n_bins);

/* pisa_hip_finalize_metric_scaled for n_points sets of limbs, one workgroup per point:
 * d_limbs as pisa_hip_reweight_hist_multi leaves them, d_hist / d_sumw2 [n_points][n_containers][n_bins],
 * total[n_points] (device or device-mapped pinned host memory); d_actual and d_extra are shared by the
 * points, d_scale (may be NULL) is read at d_scale + point * scale_point_stride (0: shared). */
int pisa_hip_finalize_metric_multi(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                   int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                   const double *d_actual, const double *d_scale,
                                   int64_t scale_point_stride, const double *d_extra, double *total,
                                   int32_t *d_status, int32_t *d_metric_status, int32_t clear_limbs,
                                   void *stream);

/* pisa_hip_finalize_metric_multi with FOUR workgroups per point, for a caller that reads the value from
 * device-mapped pinned host memory in a fit loop (the conversions of the limbs are what the one-workgroup form
 * spends its time on).  Workgroup k takes the bins b = k (mod 4) of every container and leaves the sum of the
 * metric over its bins in partial[4 * point + k]; the caller adds
 *       total = (partial[0] + partial[2]) + (partial[1] + partial[3])
 * which is pisa_hip_finalize_metric_multi's value bit for bit (the same reduction tree, cut before its last two
 * levels).  Maps, limb clearing and status words as there; a negative input makes the partial of its workgroup NaN.
 * kind = PISA_HIP_METRIC_CHI2 is refused (stats.py:160-161: its all-bins-equal rule needs every bin). */
int pisa_hip_finalize_metric_split(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                   int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                   const double *d_actual, const double *d_scale,
                                   int64_t scale_point_stride, const double *d_extra, double *partial,
                                   int32_t *d_status, int32_t *d_metric_status, int32_t clear_limbs,
                                   void *stream);

/* pisa_hip_finalize_metric_split with n_parts = 4 or 16 workgroups per point: workgroup k takes the bins
 * b = k (mod n_parts) and leaves its sum in partial[n_parts * point + k]; the caller joins them along the
 * kernel's own reduction tree, cut n_parts wide:
 *      for (w = n_parts / 2; w >= 1; w /= 2) for (i = 0; i < w; i++) partial[i] += partial[i + w];   total = partial[0]
 * (n_parts = 4: (p0 + p2) + (p1 + p3)) -- pisa_hip_finalize_metric_multi's value bit for bit.  Sixteen workgroups
 * leave every thread at most one conversion of a limb sum: the tail of a 3 072-accumulator evaluation drops
 * from 6.4 to ~3.5 us. */
int pisa_hip_finalize_metric_parts(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                   int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                   const double *d_actual, const double *d_scale,
                                   int64_t scale_point_stride, const double *d_extra, double *partial,
                                   int32_t n_parts, int32_t *d_status, int32_t *d_metric_status,
                                   int32_t clear_limbs, void *stream);

/* ------------------------------------------------ one template evaluation in ONE call
 * What `Pipeline.get_outputs()` followed by `Map.metric_total` amount to for the chain
 * osc.prob3 (calc grid) -> aeff.aeff -> utils.hist -> metric (pisa/core/pipeline.py:537-558 running
 * stage.py:584-586 per stage; the minimiser's callable pisa/analysis/analysis.py:2493-2670 calls exactly
 * that per point).  An evaluator is made once per pipeline from everything that does not change between
 * parameter points (the event columns, the binnings, the grid plan, the output buffers); per point
 *      pisa_hip_evaluator_eval(ev, params, kind, d_actual, ...)
 * enqueues  pisa_hip_prob3_grid_planned (gather tables only) -> pisa_hip_reweight_hist[_acc] ->
 * [all-reduce of the limbs] -> pisa_hip_finalize_metric_parts (16 workgroups, clear_limbs = 1)  -- the same
 * launches, the same bits as those four calls -- and, if asked to, waits for the value: the metric's sixteen
 * partial sums arrive in device-mapped pinned host memory and are joined as pisa_hip_finalize_metric_parts
 * documents.  A host binding crosses its FFI once per evaluation instead of three or four times.
 * kind = PISA_HIP_METRIC_CHI2 goes through the one-workgroup pisa_hip_finalize_metric.
 *
 * allreduce / comm (both NULL on one rank): the in-place int64 SUM of d_limbs over the ranks, called as
 *      allreduce(d_limbs, d_limbs, count, 4 (ncclInt64), 0 (ncclSum), comm, stream)
 * -- ncclAllReduce's own signature, so a binding passes RCCL's entry point and its communicator; the
 * library itself does not link RCCL. */
typedef struct pisa_hip_evaluator pisa_hip_evaluator;
typedef int (*pisa_hip_allreduce_fn)(const void *sendbuf, void *recvbuf, size_t count, int datatype,
                                     int op, void *comm, void *stream);
typedef struct {
    const pisa_hip_container *h_containers; /* copied */
    int32_t n_containers;
    int32_t n_e;                            /* energies of the calc grid */
    int32_t e_major;
    int32_t reserved;
    const pisa_hip_binning *h_calc_grid;    /* copied */
    const pisa_hip_binning *h_out_binning;  /* copied */
    pisa_hip_grid_plan *plan;               /* borrowed: must outlive the evaluator */
    const double *d_energy;                 /* [n_e] */
    double *d_pepmu;                        /* [2][3][nodes][2] gather tables (written per point) */
    int64_t *d_limbs;                       /* [n_containers][n_bins][2][PISA_HIP_ACC_LIMBS] */
    double *d_hist, *d_sumw2;               /* [n_containers][n_bins] (written per point) */
    double *partial;                        /* [16] device-mapped pinned host memory */
    int32_t *d_status, *d_metric_status;
    pisa_hip_allreduce_fn allreduce;
    void *comm;
} pisa_hip_evaluator_desc;
int pisa_hip_evaluator_create(const pisa_hip_evaluator_desc *desc, pisa_hip_evaluator **out);
int pisa_hip_evaluator_destroy(pisa_hip_evaluator *ev);
/* aeff.py:78-86: the scale of one container (aeff_scale * livetime * norms) for the points to come */
int pisa_hip_evaluator_set_scale(pisa_hip_evaluator *ev, int32_t container, double scale);
/* limbs_zero != 0: d_limbs holds zeros (left by the previous evaluation's tail): no clearing pass.
 * wait_us > 0: poll `partial` for up to that many microseconds, then synchronise the stream; *value
 *   receives the metric.  wait_us == 0: enqueue only; the caller reads `partial` / synchronises itself.
 * On return with PISA_HIP_OK the limbs are zero again (after the stream has passed the tail). */
int pisa_hip_evaluator_eval(pisa_hip_evaluator *ev, const pisa_hip_prob3_params *h_params, int32_t kind,
                            const double *d_actual, int32_t limbs_zero, int64_t wait_us, double *value,
                            void *stream);

/* --------------------------------------------------------------------- KDE */

/* The density estimator behind the KDE stage.  pisa/utils/kde_hist.py:110-120 calls
 *     k = kde.gaussian_kde(x[D,N], weights=, bw_method=, adaptive=, alpha=);  k(points[D,M])
 * from the external, un-vendored `kde` package (parity of this core is UNPINNED); the
 * estimator object below is what that binding would be replaced by.
 *
 * All-pairs Gaussian kernel sums, no cut-off (the plain O(N*M) double loop):
 *   out[j] = sum_i coef[i] * exp(-0.5 * s2[i] * (q_j - x_i)^T inv_cov (q_j - x_i))
 * d_src[dim][n_src], d_qry[dim][n_qry] (dimension-major), h_inv_cov[dim*dim]
 * row-major symmetric, dim <= 3. */
int pisa_hip_kde_eval(int32_t dim, const double *d_src, const double *d_coef, const double *d_s2,
                      int64_t n_src, const double *d_qry, int64_t n_qry, const double *h_inv_cov,
                      double *d_out, void *stream);

#define PISA_HIP_KDE_SILVERMAN 0   /* factor = (n (d+2) / 4)^(-1/(d+4)) */
#define PISA_HIP_KDE_SCOTT 1       /* factor = n^(-1/(d+4))             */

typedef struct pisa_hip_kde pisa_hip_kde;   /* opaque (host object; device data live in the caller's workspace) */

typedef struct pisa_hip_kde_info_t {
    int32_t dim, cells[3];
    int64_t n_src, n_cells;
    double factor;          /* Silverman / Scott factor                         */
    double norm;            /* sqrt(det(2 pi covariance))                       */
    double sum_w;           /* sum of the weights as given                      */
    double mean[3];         /* weighted mean                                    */
    double covariance[9];   /* weighted data covariance * factor^2 (3x3 row-major, zero padded) */
    double inv_cov[9];
    double r_cut;           /* cut-off radius in kernel sigmas (inf: none)      */
    double cell;            /* cell side in kernel sigmas                       */
    int64_t pairs_pilot;    /* kernel evaluations of the pilot estimate (cells summed through their
                             * Hermite series count P^2/23 + 1 per target: same instruction count) */
    int64_t pairs_eval;     /* kernel evaluations of the last pisa_hip_kde_evaluate */
    int32_t n_dense;        /* cells summed through a Hermite series in the pilot estimate */
} pisa_hip_kde_info_t;

/* `gaussian_kde(x, weights, bw_method, adaptive, alpha)`: weighted mean / covariance, bandwidth
 * matrix, and -- if adaptive -- the pilot densities at the sources and the local bandwidths
 * lambda_i = (pilot_i / geometric mean)^-alpha.  d_x[dim][n] dimension-major; d_w[n] may be NULL
 * (equal weights).  `tol` > 0: kernel values below tol are dropped (cell list, cost O(N k));
 * tol = 0: all pairs.  The caller owns `d_work` (>= pisa_hip_kde_workspace_bytes(dim, n) bytes)
 * and must keep its first pisa_hip_kde_resident_bytes(k) bytes untouched while the estimator
 * is in use; the rest may be reused after the call returns.  Synchronises the stream. */
int64_t pisa_hip_kde_workspace_bytes(int32_t dim, int64_t n_src);
int pisa_hip_kde_create(int32_t dim, const double *d_x, const double *d_w, int64_t n,
                        int32_t bw_method, int32_t adaptive, double alpha, double tol,
                        void *d_work, int64_t work_bytes, pisa_hip_kde **out, void *stream);
int64_t pisa_hip_kde_resident_bytes(const pisa_hip_kde *k);
/* `k(points)`: densities (normalised to integral 1) at d_qry[dim][m] -> d_out[m].  d_work: scratch of
 * >= pisa_hip_kde_eval_workspace_bytes(k, m) bytes, free again on return. */
int64_t pisa_hip_kde_eval_workspace_bytes(const pisa_hip_kde *k, int64_t n_qry);
int pisa_hip_kde_evaluate(pisa_hip_kde *k, const double *d_qry, int64_t m, void *d_work,
                          int64_t work_bytes, double *d_out, void *stream);
/* `k(points)` for the points of a lattice  x[d] = h_origin[d] + i_d h_step[d], 0 <= i_d < h_count[d]
 * (host arrays of length dim), d_out[(i_0 n_1 + i_1) n_2 + i_2] -- numpy.meshgrid(indexing="ij")
 * order, the shape in which get_hist evaluates a map (kde_hist.py:122-190: oversampled bin centres
 * of a regular binning, coszen reflection included).  Same values as pisa_hip_kde_evaluate on the
 * written-out points to rounding (<= 1e-12 relative); in 2-D with a cut-off the kernel values along
 * a lattice line come from a two-multiplication recurrence instead of one exponential each.
 * d_work >= pisa_hip_kde_lattice_workspace_bytes(k, h_step, h_count) bytes. */
int64_t pisa_hip_kde_lattice_workspace_bytes(const pisa_hip_kde *k, const double *h_step,
                                             const int64_t *h_count);
int pisa_hip_kde_evaluate_lattice(pisa_hip_kde *k, const double *h_origin, const double *h_step,
                                  const int64_t *h_count, void *d_work, int64_t work_bytes,
                                  double *d_out, void *stream);
int pisa_hip_kde_info(const pisa_hip_kde *k, pisa_hip_kde_info_t *info);
/* device pointers into the resident workspace, in the estimator's (cell-sorted) source order:
 * whitened coordinates ys[dim][n], kernel coefficients coef[n], squared inverse local bandwidths s2[n] */
int pisa_hip_kde_arrays(const pisa_hip_kde *k, const double **d_ys, const double **d_coef,
                        const double **d_s2);
int pisa_hip_kde_destroy(pisa_hip_kde *k);
/* The Hermite / local-expansion coefficients and split partial sums live in a grow-only scratch buffer of
 * the library, one per HOST THREAD (an evaluation with several worker threads holds one per worker, up to a
 * few GB each at 1e7 events).  A thread that changes streams waits for the previous stream before the
 * buffer is reused.  This call frees the calling thread's buffer. */
int pisa_hip_kde_release_scratch(void);

/* The estimators of one KDE-stage evaluation in one call (pisa/stages/utils/kde.py:154-293 loops over the
 * containers, kde_hist.py:303-372 over the pid channels: 24 independent estimators at the C3 size, all
 * evaluated on the same lattice of oversampled bin centres).  Job by job the result is that of
 * pisa_hip_kde_create(d_x, w, ...) + pisa_hip_kde_evaluate_lattice(...) with
 * w[k] = nan_to_num(d_weights[d_index ? d_index[k] : k]) (kde_hist.py:104-108); the jobs run side by side on
 * `n_threads` host threads of the library (each with its own stream and grow-only workspace; <= 0: 8), longest
 * first, after everything queued on `stream`; on return all d_out are complete.  Returns the first job error. */
typedef struct pisa_hip_kde_job {
    const double *d_x;        /* [dim][n] sample, dimension-major */
    const double *d_weights;  /* weights of the parent sample, or NULL (unit weights) */
    const int64_t *d_index;   /* n indices into d_weights, or NULL (d_weights has n entries) */
    int64_t n;
    double *d_out;            /* densities at the lattice points, numpy.meshgrid(indexing="ij") order */
    double sum_w;             /* out: sum of the weights used (the map's normalisation, kde_hist.py:107) */
    int64_t pairs_pilot;      /* out: as pisa_hip_kde_info_t */
    int64_t pairs_eval;
    int32_t status;           /* out: PISA_HIP_OK or the job's error */
    int32_t reserved;
} pisa_hip_kde_job;
int pisa_hip_kde_lattice_batch(pisa_hip_kde_job *jobs, int32_t n_jobs, int32_t dim, int32_t bw_method,
                               int32_t adaptive, double alpha, double tol, const double *h_origin,
                               const double *h_step, const int64_t *h_count, int32_t n_threads, void *stream);
/* The same in two steps, for a caller that produces the inputs sample by sample (the stage materialises the
 * event weights of one container while the estimators of the previous ones run): _submit queues the jobs and
 * returns (the job array must stay alive and untouched), _wait returns when every job submitted so far is done;
 * then each job's `status` says how it went. */
int pisa_hip_kde_lattice_submit(pisa_hip_kde_job *jobs, int32_t n_jobs, int32_t dim, int32_t bw_method,
                                int32_t adaptive, double alpha, double tol, const double *h_origin,
                                const double *h_step, const int64_t *h_count, int32_t n_threads, void *stream);
int pisa_hip_kde_lattice_wait(void);
/* returns when THESE jobs (a range of an array handed to _submit) are done -- their `status` and outputs may be read --,
 * whatever else is still queued: the stage copies and folds the densities of one container's estimators while the later
 * containers' are still being built (a job that was never submitted counts as done). */
int pisa_hip_kde_lattice_wait_jobs(const pisa_hip_kde_job *jobs, int32_t n_jobs);
/* The pool threads keep their stream, workspaces and estimator scratch between calls (grow-only, a few hundred MB per
 * thread at C3 sizes).  This call waits for the queued jobs, has every pool thread free what it holds and returns when
 * all have; the next submission allocates again. */
int pisa_hip_kde_pool_release(void);

/* How the 2-D fixed-bandwidth pilot estimate sums its cells (fast Gauss transform; truncation error below
 * the cut-off tolerance): 2 (default) = Hermite series of ALL non-empty source cells translated into one
 * local expansion per target cell, 1 = Hermite series of the cells of >= 24 sources evaluated target by target,
 * 0 = every pair directly; < 0 = query.  Returns the previous setting. */
int pisa_hip_kde_configure(int32_t use_expansion);

/* ------------------------------------------------------------------ metric */

#define PISA_HIP_METRIC_LLH 0          /* stats.py:169-253 */
#define PISA_HIP_METRIC_POISSON_LLH 1  /* stats.py:255-326 */
#define PISA_HIP_METRIC_CHI2 2         /* stats.py:98-167  */
#define PISA_HIP_METRIC_MOD_CHI2 3     /* stats.py:651-695 */
/* pisa_hip_metric only (maps in HBM; the fused tails of the evaluation take the four above) */
#define PISA_HIP_METRIC_CORRECT_CHI2 4          /* stats.py:697-730 */
#define PISA_HIP_METRIC_SIGNED_SQRT_MOD_CHI2 5  /* stats.py:762-786 */
#define PISA_HIP_METRIC_MCLLH_MEAN 6            /* stats.py:328-382, likelihood_functions.py:22-63 (a = 0) */
#define PISA_HIP_METRIC_MCLLH_EFF 7             /* stats.py:384-438 (a = 1) */
#define PISA_HIP_METRIC_CONV_LLH 8              /* stats.py:440-596 */

/* Map.metric / metric_total (pisa/core/map.py:1572-1604): per-bin metric of
 * (actual, expected[, sigma2]) and its nansum.  If n_maps > 1 the expectation
 * and variance are first summed over maps in index order
 * (distribution_maker.py:274-281; map.py:1811-1838 adds variances):
 *     d_expected[n_maps][n_bins], d_sigma2[n_maps][n_bins] (may be NULL).
 * d_per_bin[n_bins] may be NULL.  d_total[1] receives the sum; d_status[1]
 * (int32) is set to PISA_HIP_ERR_NEGATIVE for negative inputs. */
int pisa_hip_metric(int32_t kind, const double *d_actual, const double *d_expected,
                    const double *d_sigma2, int32_t n_maps, int64_t n_bins, double *d_per_bin,
                    double *d_total, int32_t *d_status, void *stream);

/* ----------------------------------------------- binned post-histogram stages */

/* d_out[i] = d_x[i] * d_scale[i] * scalar  (d_scale may be NULL = 1), then, if
 * has_floor, v < floor ? floor : v  (NaN stays NaN, as np.clip / the reference's
 * apply_floor_gufunc).  d_out may alias d_x.  Used by
 *   discr_sys.hypersurfaces  weights = clip(weights*hs_scales, 0, inf), errors *= hs_scales
 *                            (pisa/stages/discr_sys/hypersurfaces.py:236-257)
 *   utils.set_variance       manual_variance = weights*scale [floored]
 *                            (pisa/stages/utils/set_variance.py:88-104)
 *   data.csv_icc_hist        weights = count*atm_muon_scale (pisa/stages/data/csv_icc_hist.py:79-84) */
int pisa_hip_bin_scale(const double *d_x, const double *d_scale, double scalar, int32_t has_floor,
                       double floor_value, int64_t n, double *d_out, void *stream);
/* d_out[i] = sqrt(d_x[i])  (set_variance.py:84-86: errors = sqrt(manual_variance)) */
int pisa_hip_bin_sqrt(const double *d_x, int64_t n, double *d_out, void *stream);

/* ------------------------------------------------- services around the path */

#define PISA_HIP_MAX_POLY_TERMS 8

/* Replaces `lookup_indices_vectorized_{1,2,3}d` (pisa/core/bin_indexing.py:46-101): flat bin number (C order) of every
 * event among the bin EDGES of 1-3 dimensions, each dimension by the rule of `find_index`
 * (pisa/core/translation.py:504-553: half-open bins, the last edge inside; NaN counts as below); any dimension below
 * -> -1, else any dimension above -> n_bins.  h_d_sample[d]: device column [n]; h_d_edges[d]: device array of
 * h_n_edges[d] ascending edges. */
int pisa_hip_lookup_indices(const double *const *h_d_sample, const double *const *h_d_edges,
                            const int32_t *h_n_edges, int32_t ndim, int64_t n, int64_t *d_index, void *stream);

/* Replaces the gufunc `apply_probs_vectorized` (pisa/stages/osc/two_nu_osc.py:122-127; `calc_probs` :101-110) for one
 * container: d_weights[i] *= flux[i][1]*(1 - P) (flav 1), flux[i][1]*P (flav 2), flux[i][0] (flav 0), with
 * P = theta * sin^2(1.267 deltam31 L(coszen) / E) -- `theta` enters as the reference passes it (the angle itself).
 * d_nu_flux [n][2]. */
int pisa_hip_two_nu_osc(const double *d_nu_flux, double theta, double deltam31, const double *d_energy,
                        const double *d_coszen, int32_t flav, int64_t n, double *d_weights, void *stream);

/* d_out[i] = norm * d_nominal[i] * (d_energy[i] / pivot)^index  (d_nominal may be NULL = 1): the nominal flux and
 * `apply_sys_loop` of pisa/stages/flux/astrophysical.py:69-72, 121-149. */
int pisa_hip_power_law(const double *d_energy, double pivot, double index, double norm, const double *d_nominal,
                       int64_t n, double *d_out, void *stream);

/* d_out[i] = x + (target - x) * fraction, then np.clip(lo, hi) if has_clip; target = d_target[i] or, where d_target
 * is NULL, target_value (pisa/stages/reco/resolutions.py:74-96).  d_out may alias d_x. */
int pisa_hip_shift_toward(const double *d_x, const double *d_target, double target_value, double fraction,
                          int32_t has_clip, double lo, double hi, int64_t n, double *d_out, void *stream);

/* Replaces `apply_genie_sys` (pisa/stages/xsec/genie_sys.py:103-113), `apply_dis_sys` (xsec/dis_sys.py:196-206) and
 * the weight update of background/atm_muons.py:95-101:
 * d_weights[i] *= max(0, scale * prod_k (1 + (lin_k[i] + quad_k[i] p_k) p_k)), k < n_terms <= PISA_HIP_MAX_POLY_TERMS;
 * h_d_quad, or single entries of it, may be NULL (= 0); scale = 1 where the reference has none. */
int pisa_hip_poly_scale(const double *const *h_d_linear, const double *const *h_d_quad, const double *h_params,
                        int32_t n_terms, double scale, int64_t n, double *d_weights, void *stream);

/* numpy.interp(d_x, x_knots, y_knots) -- what scipy's `interp1d(kind='linear')` evaluates for 1-D float data
 * (background/atm_muons.py:82-87 with the spline of :159-164) -- for ascending knots in device memory.  A finite
 * d_x[i] outside [x_knots[0], x_knots[last]] sets *d_status (int32, may be NULL) and gives NaN: interp1d's
 * bounds_error, which the caller raises. */
int pisa_hip_interp_linear(const double *d_x_knots, const double *d_y_knots, int32_t n_knots, const double *d_x,
                           int64_t n, double *d_out, int32_t *d_status, void *stream);

/* d_out[i] = f(sum_g h_coef[g] * h_d_columns[g][i]), the sum from 0.0 in the order of the columns (the loop over
 * `grad_shift_inplace`, pisa/stages/discr_sys/ultrasurfaces.py:339-356, 362-365); mode 0: exp (us_scales), 1: 1 + sum
 * (`approx_exponential`), 2: the sum itself.  n_columns <= PISA_HIP_MAX_COMBINATION. */
#define PISA_HIP_MAX_COMBINATION 64
int pisa_hip_column_combination(const double *const *h_d_columns, const double *h_coef, int32_t n_columns, int32_t mode,
                                int64_t n, double *d_out, void *stream);

/* The element-wise helpers of pisa/utils/vectorizer.py:44-209 (its gufuncs `scale_gufunc` ... `replace_where_counts_gt_gufunc`):
 *   SCALE out = a*scalar;  MUL out = a*b;  IMUL out *= a;  IMUL_AND_SCALE out *= a*scalar;
 *   ITRUEDIV out /= a (0 where a == 0);  ASSIGN out = a;  POW out = a**scalar;  SQRT out = sqrt(a);
 *   REPLACE_WHERE_COUNTS_GT out = a where b > scalar.   d_out may alias d_a. */
#define PISA_HIP_VEC_SCALE 0
#define PISA_HIP_VEC_MUL 1
#define PISA_HIP_VEC_IMUL 2
#define PISA_HIP_VEC_IMUL_AND_SCALE 3
#define PISA_HIP_VEC_ITRUEDIV 4
#define PISA_HIP_VEC_ASSIGN 5
#define PISA_HIP_VEC_POW 6
#define PISA_HIP_VEC_SQRT 7
#define PISA_HIP_VEC_REPLACE_WHERE_COUNTS_GT 8
int pisa_hip_vector_op(int32_t op, const double *d_a, const double *d_b, double scalar, int64_t n, double *d_out,
                       void *stream);

/* Replaces `decoherence.calc_probs` (pisa/stages/osc/decoherence.py:449-466 over `calc_decoherence_probs` :66-106):
 * d_probability[n][3][3], rows (1, 0, 0), (0, 1 - D, D), (0, D, 1 - D) with the numu disappearance D(E, L) of
 * `_calc_numu_disappearance_prob_3flav` (:229-269; h_coef[k] = |U[2][j]|^2 |U[2][k]|^2, h_gamma[k] in GeV, h_delta[k]
 * in eV^2 for the pairs (1,0), (2,0), (2,1)) or, two_flavor != 0, of `_calc_numu_disappearance_prob_2flav`
 * (:112-139; h_coef[0] = 0.5 sin^2(2 theta23), h_gamma[0] = gamma32 in eV, h_delta[0] = dm32).  d_energy in GeV,
 * d_baseline in km. */
int pisa_hip_decoherence_probs(const double *h_coef, const double *h_gamma, const double *h_delta, int32_t two_flavor,
                               const double *d_energy, const double *d_baseline, int64_t n, double *d_probability,
                               void *stream);

/* -------------------------------------------------------------------- flux */

/* 2-D (azimuth-averaged) Honda flux table prepared for `pisa_hip_flux_2d`
 * (host side: scipy splrep exactly as pisa/utils/flux_weights.py:50-131 builds the
 * integral-preserving band splines).  All pointers are device memory. */
typedef struct pisa_hip_flux_table {
    int32_t n_bands;          /* coszen bands of the table (20)                               */
    int32_t n_knots_e;        /* knots of a band spline in log10(E) (106)                     */
    const double *d_knots_e;  /* [n_knots_e]                                                  */
    const double *d_coef_e;   /* [4][n_bands][n_knots_e-4] cubic B-spline coefficients of the
                                 running integrals; primaries (nue, numu, nuebar, numubar),
                                 bands in ascending coszen                                   */
    int32_t n_knots_cz;       /* knots of the coszen spline = n_bands + 1 + 4                 */
    int32_t enpow;            /* flux_weights.py `enpow` (1)                                  */
    const double *d_knots_cz; /* [n_knots_cz]                                                 */
    const double *d_cardinal; /* [n_knots_cz-4][n_bands+1]: coefficients of the interpolating
                                 cubic splines through the unit vectors at the coszen knots   */
    double cz_step;           /* 0.1 (flux_weights.py:343)                                    */
} pisa_hip_flux_table;

/* `calculate_2d_flux_weights` (pisa/utils/flux_weights.py:267-349) for all four
 * primaries at once, as flux.honda_ip uses it (pisa/stages/flux/honda_ip.py:86-104):
 * d_nu_flux[n][2] = (nue, numu), d_nubar_flux[n][2] = (nuebar, numubar).
 * d_status (int32, may be NULL) is set non-zero if a coszen is outside [-1, 1]
 * (the reference raises ValueError, :318-319). */
int pisa_hip_flux_2d(const pisa_hip_flux_table *h_table, const double *d_true_energy,
                     const double *d_true_coszen, int64_t n, double *d_nu_flux,
                     double *d_nubar_flux, int32_t *d_status, void *stream);

/* `hist.apply_function` with a binned calc_mode (pisa/stages/utils/hist.py:132-160):
 *   hist = (unc*w) @ T, sumw2 = (unc*w)^2 @ T, bin_unc2 = (unc^2*w) @ T
 * d_weights[n_calc], d_unc_weights[n_calc] (NULL = 1); T = `hist_transform` (:69-84, event counts
 * per (calc bin, output bin)) given by its non-zeros grouped by output bin: d_ptr[n_out+1],
 * d_col[nnz] (calc bin), d_val[nnz] (count).  Outputs [n_out], any may be NULL. */
int pisa_hip_transform_apply(const double *d_weights, const double *d_unc_weights, const int32_t *d_ptr,
                             const int32_t *d_col, const double *d_val, int64_t n_out, double *d_hist,
                             double *d_sumw2, double *d_bin_unc2, void *stream);

/* Flux on the oscillation grid (flux stages whose calc_mode is osc.prob3's, e.g. the IceCube 3-year
 * cfgs): the reference looks up nu_flux and prob_e / prob_mu at the same node for every event
 * (container.py:981-1012) and multiplies them per event (prob3.py:621-622).  Here the products are
 * formed per node,
 *   d_out[c][node] = (f_e[c][node] * P_e[c][node], f_mu[c][node] * P_mu[c][node]),
 * h_d_flux_nodes[c] = device pointer to container c's [n_nodes][2] flux, d_pepmu = the gather tables
 * of pisa_hip_prob3_grid[_planned]; d_out[c] is then passed as pisa_hip_container.d_pepmu with the
 * static pair (w0*aeff, w0*aeff) in d_weighted_flux[_q]: a flux systematic costs this launch. */
int pisa_hip_flux_prob_tables(const double *const *h_d_flux_nodes, const int32_t *h_nubar,
                              const int32_t *h_flav, int32_t n_containers, const double *d_pepmu,
                              int64_t n_nodes, double *d_out, void *stream);

/* Refresh of the fused kernel's folded flux column after a flux stage rewrote `nu_flux`
 * (flux stages write container['nu_flux'], pisa/stages/flux/barr_simple.py:100; the reference then
 * multiplies it in every evaluation, prob3.py:621-622):
 *   out[e] = d_static_w[e] * d_flux[d_perm ? d_perm[e] : e][0..1]
 * d_flux[.][2] in the container's own event order, d_perm[n] (int64, may be NULL) the resident
 * order of the engine, d_static_w[n] = initial_weights*weighted_aeff in resident order.
 * layout 0: d_out[n][2] (pisa_hip_container.d_weighted_flux); layout 1: the quad-blocked
 * d_weighted_flux_q (padding entries are left untouched). */
int pisa_hip_fold_flux(const double *d_flux, const int64_t *d_perm, const double *d_static_w,
                       int64_t n, int32_t layout, double *d_out, void *stream);

/* pisa_hip_fold_flux for all containers in one launch; set k is folded exactly like
 * pisa_hip_fold_flux(set k). */
typedef struct {
    int64_t n;
    const double *d_flux;       /* [.][2] container order */
    const int64_t *d_perm;      /* [n] or NULL */
    const double *d_static_w;   /* [n] resident order */
    double *d_out;              /* layout 0: [n][2]; layout 1: quad-blocked */
    int32_t layout;
    int32_t reserved;
} pisa_hip_fold_set;
int pisa_hip_fold_flux_multi(const pisa_hip_fold_set *h_sets, int32_t n_sets, void *stream);

/* `apply_sys_vectorized` (pisa/stages/flux/barr_simple.py:147-233). Flux arrays [n][2]. */
int pisa_hip_barr_simple(const double *d_true_energy, const double *d_true_coszen,
                         const double *d_nu_flux_nominal, const double *d_nubar_flux_nominal,
                         int64_t nubar, double nue_numu_ratio, double nu_nubar_ratio,
                         double delta_index, double Barr_uphor_ratio, double Barr_nu_nubar_ratio,
                         int64_t n, double *d_out, void *stream);

/* The same for all containers of a pipeline in one launch (the stage's loop over containers,
 * pisa/stages/flux/barr_simple.py:83-104, with one set of parameter values): set k is evaluated exactly
 * like pisa_hip_barr_simple(set k) -- same bits.  On the oscillation grid (calc_mode = the binning of
 * osc.prob3, as in the IceCube 3-year cfgs) a container is a few 10^4 nodes and the launches, not the
 * arithmetic, are what a flux systematic costs. */
typedef struct {
    int64_t n;                          /* elements of this container */
    const double *d_true_energy;        /* [n] */
    const double *d_true_coszen;        /* [n] */
    const double *d_nu_flux_nominal;    /* [n][2] */
    const double *d_nubar_flux_nominal; /* [n][2] */
    double *d_out;                      /* [n][2] nu_flux */
    int32_t nubar;                      /* +1 / -1 */
    int32_t reserved;
} pisa_hip_barr_set;
int pisa_hip_barr_simple_multi(const pisa_hip_barr_set *h_sets, int32_t n_sets,
                               double nue_numu_ratio, double nu_nubar_ratio, double delta_index,
                               double Barr_uphor_ratio, double Barr_nu_nubar_ratio, void *stream);

/* A flux systematic moved and the flux is held per event (flux.barr_simple in an event representation,
 * pisa/stages/flux/barr_simple.py:83-104, then multiplied in every evaluation, prob3.py:621-622): ONE pass
 * over the events instead of pisa_hip_barr_simple_multi + pisa_hip_fold_flux_multi.
 * pisa_hip_barr_factors (once per event set): the parts of apply_sys_vectorized (:147-233) that depend on
 *   (E, coszen) only -- d_factors[5][n] = ModFlux(nue), ModFlux(numu), the energy and the zenith factor of
 *   the up/horizontal Gaussian, log(E / E_pivot); d_status (int32, may be NULL) is set if an energy is not
 *   positive (such a set must use the two-pass calls, which keep the reference's answers there).
 * pisa_hip_barr_fold_multi (per moved systematic): every array of a set in the SAME order and layout as
 *   d_out -- the engine's resident order, e.g. the quad-blocked d_weighted_flux_q --
 *   d_out[p] = d_static_w[p] * apply_sys(nominal fluxes[p], factors[p]; parameters): the bits of the two
 *   calls it replaces. */
typedef struct {
    int64_t n;                          /* positions (padding included) */
    const double *d_nu_flux_nominal;    /* [n][2] */
    const double *d_nubar_flux_nominal; /* [n][2] */
    const double *d_factors;            /* [5][n] from pisa_hip_barr_factors */
    const double *d_static_w;           /* [n] initial_weights*weighted_aeff (0 at padding) */
    double *d_out;                      /* [n][2] */
    int32_t nubar;                      /* +1 / -1 */
    int32_t reserved;
} pisa_hip_barr_fold_set;
int pisa_hip_barr_factors(const double *d_true_energy, const double *d_true_coszen, int64_t n,
                          double *d_factors, int32_t *d_status, void *stream);
int pisa_hip_barr_fold_multi(const pisa_hip_barr_fold_set *h_sets, int32_t n_sets, double nue_numu_ratio,
                             double nu_nubar_ratio, double delta_index, double Barr_uphor_ratio,
                             double Barr_nu_nubar_ratio, void *stream);

/* ------------------------------------------------------- raw device memory */
/* Thin wrappers so hosts without torch (a cgo/ctypes binding of the
 * reference) can own device buffers. */
int pisa_hip_malloc(void **d_ptr, int64_t bytes);
int pisa_hip_free(void *d_ptr);
int pisa_hip_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes, void *stream);
int pisa_hip_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes, void *stream);
int pisa_hip_memset(void *d_dst, int value, int64_t bytes, void *stream);
int pisa_hip_stream_synchronize(void *stream);
int pisa_hip_set_device(int device);

#ifdef __cplusplus
}
#endif
#endif /* PISA_HIP_H */
