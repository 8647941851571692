/*
 * gmmhmm.h -- C ABI of the MI355X-native GMM-HMM core (libgmmhmm.so).
 *
 * Drop-in boundary for the hot path of tjysdsg/speech-recognition's
 * `sr/recognition` package.  The reference has no FFI layer of its own (it is
 * pure Python/numpy), so each entry point below names the reference function
 * whose inner loop it replaces (paths relative to the reference repo); the
 * Python mirror in speech-recognition_amd/sr/recognition binds them with
 * ctypes and keeps the reference's names, arguments and error behaviour
 * (INTEGRATION.md shows the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no C++/torch types.
 *   - every call returns 0 (GH_OK) or a negative gh_status; the message of the
 *     last failure on the calling thread is `gh_last_error()`.
 *   - host pointers are caller-owned and only read/written during the call;
 *     `*_dev` arguments are device pointers on the context's GPU.
 *   - handles are created/destroyed by the library; one gh_ctx per GPU; a
 *     handle is not thread-safe, distinct handles are.
 *   - all costs are NEGATIVE LOG likelihoods ("cost" in the reference), arcs
 *     are `cost[to, from]`, +inf = no arc (decode.py:12-13).
 *   - dtype: GH_F32 or GH_F64 selects the arithmetic of the likelihood kernel
 *     and the type of the resident feature / likelihood matrices; the dynamic
 *     programs always accumulate in fp64 (reference arithmetic is float64).
 */
#ifndef GMMHMM_H
#define GMMHMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GH_OK = 0,
    GH_ERR_INVALID = -1,     /* bad argument (shape, NULL, range)                       */
    GH_ERR_HIP = -2,         /* a HIP runtime call failed                               */
    GH_ERR_NOMEM = -3,       /* device or host allocation failed                        */
    GH_ERR_NODEVICE = -4,    /* no usable GPU                                           */
    GH_ERR_SELF_POINTER = -5,/* a DP cell chose itself as origin: decode.py:120-121     */
    GH_ERR_UNSUPPORTED = -6, /* shape outside what the kernels are built for            */
    GH_ERR_COMM = -7         /* a collective failed or timed out; the communicator has been aborted */
} gh_status;

typedef enum { GH_F32 = 0, GH_F64 = 1 } gh_dtype;

typedef struct gh_ctx gh_ctx;            /* one GPU: device id, stream, scratch arena      */
typedef struct gh_gmm gh_gmm;            /* S diagonal-covariance mixtures, packed SoA      */
typedef struct gh_batch gh_batch;        /* ragged batch of utterances resident in HBM      */
typedef struct gh_lattices gh_lattices;  /* one or more DP graphs (rows, arcs, starts, ends)*/

const char* gh_last_error(void);
int gh_version(void);

/* ------------------------------------------------------------------ context */
int gh_ctx_create(int device, gh_ctx** out);
void gh_ctx_destroy(gh_ctx* ctx);
int gh_ctx_sync(gh_ctx* ctx);
/* hipDeviceSynchronize on the context's GPU: every stream of this library's runtime has drained (the bracket of a
 * timed region when several contexts are in flight) */
int gh_device_sync(gh_ctx* ctx);
/* Number of launches the last gh_viterbi* / gh_forward_backward call on this context was cut into: the DP scratch
 * (back-pointers, alpha columns) of one launch is bounded by a quarter of the free HBM (<= 24 GiB), or by
 * GMMHMM_SCRATCH_BUDGET=<bytes>[K|M|G]; larger batches run as several launches with identical results. */
int gh_ctx_last_chunks(const gh_ctx* ctx);
/* Compatibility switches of a context (default 1 = bit 0 set; GMMHMM_COMPAT=0 clears it at creation).
 *   bit 0  gh_loglik* return +inf for a state whose every weighted density is below 2^-1075: GMM.evaluate sums w pdf in
 *          the LINEAR domain (hmm_state.py:114-120), where such terms round to 0 and -log 0 = inf -- a decode through
 *          these frames is then unreachable exactly where it is in the reference.  Without it the kernels stay in the
 *          log domain and return the (large, finite) cost -- the numerically kinder choice for un-normalised features,
 *          but not what the reference computes.  The test is made on the term's total logarithm (one compare against a
 *          runtime threshold in the kernels' epilogues: no cost); the reference can also lose a term earlier, in
 *          exp(-q/2) before the normalisation, when w * norm > 1 (variances below ~1/2pi): models with such a component
 *          get a second pass over the few entries whose cost lies within log(w norm) of the threshold, which re-tests
 *          them per component (round 4; a no-op launch for ordinary models).
 *   bit 1  (off by default; GMMHMM_LSE=f32exp sets it at creation) the fp64 likelihood kernel takes the exponentials of its
 *          log-sum-exp in FP32: maximum, differences and the final max + log(sum) stay fp64, the terms 2^(a - max) <= 1
 *          and their sum are fp32.  |delta nll| <= ~2.4e-7 absolute; bench.py reports the kernel both ways with the
 *          measured difference and path mismatch rates.  Mixtures of >= 4 components, fp64 batches. */
int gh_ctx_set_compat(gh_ctx* ctx, int flags);
/* raw hipStream_t of the context (for torch interop) */
void* gh_ctx_stream(gh_ctx* ctx);
/* number of GPUs the library's HIP runtime sees (<= 0: none).  Callers must not dlopen a HIP runtime of their own
 * to find out: a second runtime in the process does not know this library's streams and allocations. */
int gh_device_count(void);
/* hipEvent timing on the context's stream, through the library's own runtime (measurement plumbing of bench.py; no
 * reference counterpart): record an event behind the work submitted so far, make the stream wait for an event of
 * another context, elapsed milliseconds between two completed events (waits for the second). */
int gh_event_create(gh_ctx* ctx, void** out_event);
void gh_event_destroy(void* event);
int gh_event_record(gh_ctx* ctx, void* event);
int gh_ctx_wait_event(gh_ctx* ctx, void* event);
int gh_event_elapsed_ms(void* start, void* stop, float* out_ms);

/* ----------------------------------------------------------- emission model
 * Packs S states x M components x D dims.  Replaces the per-state object graph
 * GMM -> [MultivariateNormal] (hmm_state.py:5-45,100-120).  `weight` is used as
 * given (NOT renormalised: hmm_state.py:115 multiplies whatever `w` holds).
 * var <= 0 is rejected (the reference raises LinAlgError from np.linalg.inv,
 * hmm_state.py:17,30). */
int gh_gmm_create(gh_ctx* ctx, int S, int M, int D,
                  const double* mean /*[S,M,D]*/, const double* var /*[S,M,D]*/,
                  const double* weight /*[S,M]*/, gh_gmm** out);
void gh_gmm_destroy(gh_gmm* g);

/* ---------------------------------------------------------------- utterances
 * feats: row-major [N, D] (N = utt_offsets[U]); utterance u owns frames
 * [utt_offsets[u], utt_offsets[u+1]).  Replaces the Python list of per-utterance
 * np.ndarray[T_u, D] every reference entry point takes (hmm.py:57, decode.py:80). */
int gh_batch_create(gh_ctx* ctx, gh_dtype dtype, int D, int64_t N, int64_t U,
                    const void* feats_host, const int64_t* utt_offsets, gh_batch** out);
/* same, features already in HBM (e.g. a torch tensor's data_ptr); not copied, not freed */
/* gh_batch_create with a choice of wire format: wire_dtype == dtype, or GH_F32 for a GH_F64 batch -- half the bytes over
 * the host link, widened on the device (the caller's features are then fp32-rounded BEFORE the call; the arithmetic stays
 * fp64).  pin != 0: the host buffer is page-locked for the duration of the copy (hipHostRegister) so that the copy runs at
 * the link's rate; pin == 2: it STAYS page-locked after the call (registering costs about what one faster copy saves: a
 * caller that uploads from the same buffer repeatedly pays it once) until gh_host_unpin(feats_host), which must come
 * before the buffer is freed.  For callers that hand over host buffers every time (bench.py's `pcie_inclusive` leg). */
int gh_host_unpin(const void* host_ptr);
int gh_batch_create_wire(gh_ctx* ctx, gh_dtype dtype, gh_dtype wire_dtype, int pin, int D, int64_t N, int64_t U,
                         const void* feats_host, const int64_t* utt_offsets /*[U+1]*/, gh_batch** out);
int gh_batch_wrap(gh_ctx* ctx, gh_dtype dtype, int D, int64_t N, int64_t U,
                  void* feats_dev, const int64_t* utt_offsets, gh_batch** out);
/* Rows idx[0..n) of a resident batch as a new resident batch of U utterances (utt_offsets as in gh_batch_create), copied
 * on the device: continuous_train's regrouping of the frames per state (continuous_speech.py:107-113, np.vstack of the
 * segments) without a trip through the host. */
int gh_batch_gather(gh_ctx* ctx, const gh_batch* src, const int64_t* idx /*[n]*/, int64_t n, int64_t U,
                    const int64_t* utt_offsets /*[U+1]*/, gh_batch** out);
/* Runs of consecutive rows instead of single rows: rows [start[r], start[r] + len[r]) of `src` become rows
 * [dest[r], ...) of the result (n rows in all; the runs must tile it exactly). */
int gh_batch_gather_runs(gh_ctx* ctx, const gh_batch* src, int64_t n_runs, const int64_t* start, const int64_t* len,
                         const int64_t* dest, int64_t n, int64_t U, const int64_t* utt_offsets /*[U+1]*/, gh_batch** out);
/* feats += scale * z, z ~ N(0, 1) independent per feature from a counter-based generator keyed on (seed, element): the
 * copies of a tiled batch (gh_batch_tile) become utterances of their own on the device.  Not in the reference (its data
 * come from files); for synthetic workloads at sizes whose host synthesis would cost more than the run. */
int gh_batch_jitter(gh_ctx* ctx, gh_batch* b, uint64_t seed, double scale);
/* `reps` copies of a resident batch back to back as a new resident batch of reps * U utterances (device-to-device
 * copies; measurement plumbing: a large batch from a small upload; no reference counterpart) */
int gh_batch_tile(gh_ctx* ctx, const gh_batch* src, int reps, gh_batch** out);
void gh_batch_destroy(gh_batch* b);
/* N3 front-end: cepstra [N, C] (fp64, ragged by utt_offsets) -> [cepstra | delta | delta-delta]
 * (delta_feature, sr/core.py:13-22) -> per-utterance (x - mean) / std (standardize,
 * sr/feature/feature.py:85-88), i.e. the tail of load_wav_as_mfcc (sr/core.py:41-44), computed on the
 * GPU into a resident batch of D = 3C features.  Every utterance needs >= 2 frames (the reference
 * indexes feat[i + 1]).  gh_batch_fetch_features copies a batch's feature matrix back ([N, D], dtype).
 * mode 0: stack + standardise (D = 3C); 1: stack only, raw [ceps|delta|ddelta] (D = 3C); 2: standardise the C
 * input columns as they are (D = C). */
int gh_batch_create_from_cepstra(gh_ctx* ctx, gh_dtype dtype, int mode, int C, int64_t N, int64_t U,
                                 const double* ceps_host, const int64_t* utt_offsets, gh_batch** out);
int gh_batch_fetch_features(gh_ctx* ctx, const gh_batch* b, void* out_host);
/* N3 front-end, first half: mfcc_features (sr/feature/feature.py:43-82) without the wav-file read:
 * pre-emphasis 0.97 (:45-46) -> frames of int(frame_size*rate) samples every int(frame_stride*rate)
 * (segment, :7-22; ceil(len/step) frames, zero-extended) -> centred zero padding to the next power of
 * two (zero_padding, :25-40) -> Hamming window over the PADDED frame (:52) -> |rfft(., 512)|^2 / 512
 * (:54-56) -> 40 triangular mel filters from low_freq to high_freq (<= 0: rate/2) (:58-75) -> log10, 0
 * replaced by eps (:76-78) -> DCT-II (ortho), coefficients 1..13 (:80-81).
 * samples: U utterances back to back, sample_fmt 0 = int16, 1 = float32, 2 = float64; sample_off[U+1];
 * frame_off[U+1] must hold the reference's frame counts (ceil(len/step)) -- gh_mfcc_frames computes one.
 * out_fbank [N,40] / out_mfcc [N,13] (fp64, either may be NULL) receive what mfcc_features returns.
 * gh_batch_create_from_pcm chains the cepstra on the device into gh_batch_create_from_cepstra's
 * kernels (delta, delta-delta, standardise; same `mode`): PCM in, resident 39-dim batch out. */
int64_t gh_mfcc_frames(int64_t n_samples, int sample_rate, double frame_stride);
int gh_mfcc(gh_ctx* ctx, int sample_fmt, int sample_rate, double frame_size, double frame_stride, double low_freq,
            double high_freq, int64_t U, const void* samples, const int64_t* sample_off, const int64_t* frame_off,
            double* out_fbank, double* out_mfcc);
int gh_batch_create_from_pcm(gh_ctx* ctx, gh_dtype dtype, int mode, int sample_fmt, int sample_rate, double frame_size,
                             double frame_stride, double low_freq, double high_freq, int64_t U, const void* samples,
                             const int64_t* sample_off, const int64_t* frame_off, gh_batch** out);

/* ------------------------------------------------ A3: batched GMM.evaluate
 * nll[n, s] = -log sum_m w[s,m] N(x_n; mean[s,m], diag var[s,m])  for every
 * frame of the batch and every state (hmm_state.py:114-120, log-domain).
 * The [N, S] matrix stays resident in the batch (dtype of the batch) for the
 * dynamic programs; `out_host` (may be NULL) receives a copy.
 * gh_loglik_dev_ptr returns the resident matrix (device pointer, [N,S]). */
int gh_loglik(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, void* out_host /*[N,S] or NULL*/);
/* Same, restricted: utterance u only needs the states [state_lo[u], state_hi[u]) -- forced alignment and EM with a known
 * transcript never look at the other words' states (the reference evaluates exactly the states of the lattice it was
 * given, decode.py:123).  Entries outside an utterance's range are unspecified (zero after the first call).  10 word
 * models: a tenth of the work of gh_loglik. */
int gh_loglik_subset(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int32_t* state_lo /*[U]*/,
                     const int32_t* state_hi /*[U]*/);
/* The same with SEVERAL state ranges per utterance: utterance u needs [state_lo[r], state_hi[r]) for r in
 * [range_off[u], range_off[u+1]) -- the states of the words of its transcript (continuous_speech.py:80: the lattice of an
 * utterance holds its own words' states only), which for word strings is a set, not an interval. */
int gh_loglik_sets(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int64_t* range_off /*[U+1]*/,
                   const int32_t* state_lo, const int32_t* state_hi);
void* gh_loglik_dev_ptr(gh_batch* b);
/* copy of the resident [N,S] likelihood matrix (dtype of the batch) after gh_loglik / gh_loglik_subset */
int gh_loglik_fetch(gh_ctx* ctx, const gh_batch* b, void* out_host /*[N,S]*/);
/* weighted component densities in the log domain, log(w_m pdf_m(x)) for one
 * state over arbitrary frames (hmm_state.py:114-116 with
 * return_neg_log_likelihood=False, in logs): out[n, m], fp64. */
int gh_component_loglik(gh_ctx* ctx, const gh_gmm* g, int state, int64_t N,
                        const double* x_host /*[N,D]*/, double* out_host /*[N,M]*/);
/* A2: diagonal-Gaussian negative log-likelihood / Euclidean distance matrix
 * dist[i, n] between frames x[n] and templates y[i] (hmm_state.py:48-58;
 * default dist_fun of kmeans.py:111,167).  var may be NULL (Euclidean),
 * [1,D] (shared, kmeans.py:183 uses cov[0]) or [K,D] (dtw with variance[i],
 * decode.py:38,59). */
int gh_distance_matrix(gh_ctx* ctx, int64_t N, int K, int D, const double* x_host /*[N,D]*/,
                       const double* y_host /*[K,D]*/, const double* var_host, int var_rows,
                       double* out_host /*[K,N]*/);

/* ------------------------------------------------------------ DP graphs
 * L graphs, concatenated.  Graph l owns rows [row_off[l], row_off[l+1]) etc;
 * arc endpoints and start/end rows are LOCAL row indices of their graph.
 *   row_state[r]  >= 0 : emitting row scored by that state of the gh_gmm
 *                 == -1: non-emitting row (NES, hmm_state.py:81-97; evaluate -> 0)
 *   arcs          : finite entries of the reference's dense transitions[to, from]
 *   start rows    : cells (r, 0) initialised with their emission only; the
 *                   reference has exactly one, row 0 (decode.py:99-101); stacking
 *                   W isolated word models into one graph gives W of them
 *   end rows      : candidate final rows in the LAST column, in the caller's
 *                   order (decode.py:126-134: the last of equal minima wins)
 * Replaces the dense R x R matrix built by build_state_sequences
 * (continuous_speech.py:13-53) / HMM.transitions (hmm.py:24). */
int gh_lattices_create(gh_ctx* ctx, int L,
                       const int64_t* row_off /*[L+1]*/, const int32_t* row_state,
                       const int64_t* arc_off /*[L+1]*/, const int32_t* arc_to,
                       const int32_t* arc_from, const double* arc_cost,
                       const int64_t* start_off /*[L+1]*/, const int32_t* start_rows,
                       const int64_t* end_off /*[L+1]*/, const int32_t* end_rows,
                       gh_lattices** out);
/* The forced-alignment graphs of continuous_train (continuous_speech.py:80-82: build_state_sequences(models,
 * [[l] for l in labels]) -- one word per layer) straight from the transcripts: W word models of n states each,
 * word_trans [W, n, n] with entry [i, j] = cost of j -> i and +inf = no arc (HMM.transitions, hmm.py:24), the first
 * state index of every word (NULL: w * n), and L label strings (graph l = labels[label_off[l] .. label_off[l+1])).
 * Graph l equals the one build_state_sequences makes -- row 0 non-emitting, then per word its n states and one
 * non-emitting row; zero-cost arcs into a word's first and out of its last state; end row = last state of the last
 * word -- and every call that takes a gh_lattices accepts the handle. */
int gh_lattices_create_transcripts(gh_ctx* ctx, int W, int n, const double* word_trans /*[W,n,n]*/,
                                   const int32_t* state_base /*[W] or NULL*/, int64_t L,
                                   const int64_t* label_off /*[L+1]*/, const int32_t* labels, gh_lattices** out);
void gh_lattices_destroy(gh_lattices* l);
/* Rank beam for gh_viterbi / gh_viterbi_labels on these graphs (decode_hmm_states itself has no
 * pruning -- this is the beam of dtw, decode.py:62-68, carried over to lattices).  After every column but the last, the
 * cells of the finished column are ranked in ascending (cost, row) order and every finite cell ranked >= beam reads +inf
 * when the next column takes it as an origin (and shows as +inf in out_costs); reads inside the column -- arcs touching
 * a non-emitting row -- are not affected.  beam <= 0 switches pruning off (the default; results are then bit-identical
 * to the unpruned kernels).  A pruned decode runs on the generic kernel. */
int gh_lattices_set_beam(gh_lattices* l, int beam);

/* Which special forms the graphs were recognised in when they were created (the kernels behind gh_viterbi and
 * gh_forward_backward are chosen by form; anything else runs on the row-per-lane kernels).  Bit 0: one left-to-right
 * chain (hmm.py:126-135); bit 1: K layers of the same W words (build_state_sequences, continuous_speech.py:13-53);
 * bit 2: word-loop grammar; bit 3: one word per layer, a graph per transcript (continuous_speech.py:80); bit 4:
 * one-word chains for forward-backward.  < 0: NULL argument.
 * What the forms take: words of 2 .. 8, 12 or 16 states with arcs from s, s-1, s-2; bit 1: up to 16 words per layer
 * and 8 layers (16 layers for words of <= 8 states), or 17 .. 64 words per layer with <= 8 layers of <= 8 states;
 * bit 2: up to 16 words, or 17 .. 64 words of <= 8 states; bit 3: transcripts of up to 16 words. */
int gh_lattices_forms(const gh_lattices* l);

/* --------------------------------------------------- A6: decode_hmm_states
 * Viterbi over graph utt_lattice[u] (NULL: graph 0) for every utterance of the
 * batch, using the resident likelihood matrix (gh_loglik must have run).
 * Reproduces decode.py:80-146 including same-column hops through non-emitting
 * rows, first-minimum tie-break among origins, last-minimum among end points,
 * and the path convention (end -> start, end cell excluded, stops on reaching
 * column 0).
 *   out_end_cost  [sum_u n_end(u)]  cost of every end row at the last column (may be NULL: the costs then stay on the
 *                 device and only out_best_end comes back -- 8 MB less copy-back for 100 000 utterances x 10 word ends)
 *   out_best_end  [U]               index (into the graph's end list) chosen
 *   out_path      [path_off[U], 2]  (row, col) pairs, utterance u at path_off[u];
 *                                   path_off[u+1]-path_off[u] >= gh_viterbi_path_cap
 *   out_path_len  [U]
 *   out_costs     full cost matrices, utterance u at costs_off[u], row-major
 *                 [R_u, T_u] like the reference's `costs` (tests / single-utterance API)
 * Any output pointer may be NULL. */
int gh_viterbi(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b,
               const int32_t* utt_lattice /*[U] or NULL*/,
               double* out_end_cost, int32_t* out_best_end,
               int32_t* out_path, const int64_t* path_off /*[U+1]*/, int32_t* out_path_len,
               double* out_costs, const int64_t* costs_off /*[U+1]*/);
/* A5 + A2 (and A6 + A3 with one component) in ONE sweep: HMM.evaluate for word models with a single Gaussian per state,
 * hmm.py:131-135 -- `dtw(x, self.mu, mahalanobis, self.transitions, self.sigma)` (use_gmm=False; hmm_state.py:48-58,
 * decode.py:7-77) or `decode_hmm_states(x, [GMM(mu, sigma, 1)...], self.transitions)` (hmm_state.py:114-120) -- which both
 * score a cell while they fill the cost matrix.  Same graphs, outputs and conventions as gh_viterbi (one graph for the
 * whole batch), but every lane evaluates its own state's Gaussian against the frame: no [N, S] likelihood matrix is
 * written or read (8 D + 4 bytes of HBM traffic per frame instead of 8 D + 16 S + 4).
 *   g           model with ONE component per state (its weight enters as -log w, 1.0 for a plain Gaussian)
 *   log_domain  1: the distance is mahalanobis() -- a log-domain quantity that never underflows (use_gmm=False models);
 *               0: the cell is GMM.evaluate: with compat bit 0 of the context, +inf where exp(-q/2) or the weighted
 *               density rounds to 0 in the reference's linear domain
 * The fused kernel takes left-to-right chains (any number of word models side by side), D <= 40, no beam, no
 * single-frame utterance; anything else runs as gh_loglik + gh_viterbi inside the call with identical results
 * (gh_ctx_last_fused tells which; GMMHMM_FUSED=0 forces the two-kernel form). */
int gh_viterbi_fused(gh_ctx* ctx, const gh_gmm* g, const gh_lattices* lat, gh_batch* b, int log_domain,
                     double* out_end_cost, int32_t* out_best_end,
                     int32_t* out_path, const int64_t* path_off /*[U+1]*/, int32_t* out_path_len,
                     double* out_costs, const int64_t* costs_off /*[U+1]*/);
/* 1: the last gh_viterbi_fused call on this context ran the fused kernel; 0: gh_loglik + gh_viterbi */
int gh_ctx_last_fused(const gh_ctx* ctx);
/* upper bound on the number of path cells of an utterance of T frames on graph l */
int64_t gh_viterbi_path_cap(const gh_lattices* lat, int l, int64_t T);
/* A6 + A12 in one call: the same decode, but the path stays on the device and only the DECODED LABEL
 * SEQUENCE comes back -- main.py:59-67 / split_result (main.py:39-52): reverse the path to start -> end
 * order, and for every maximal run of emitting rows between non-emitting ones report the label of its
 * first row.  row_label [sum_l R_l] (graphs concatenated like row_state): label of each row, < 0 on
 * non-emitting rows (e.g. the word index).  out_labels: utterance u at label_off[u], capacity
 * label_off[u+1]-label_off[u] (gh_viterbi_path_cap/2 + 1 always suffices); out_n_labels [U].
 * A continuous decode of 2000 x 317 frames returns ~60 KB instead of 15 MB of (row, col) pairs. */
int gh_viterbi_labels(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b,
                      const int32_t* utt_lattice /*[U] or NULL*/, const int32_t* row_label,
                      double* out_end_cost, int32_t* out_best_end,
                      int32_t* out_labels, const int64_t* label_off /*[U+1]*/, int32_t* out_n_labels);

/* A6 + the regrouping step of continuous_train in one call (continuous_speech.py:80-106): the same decode, the path
 * stays on the device and what comes back is, per FRAME of the batch, the state whose training data the frame joins:
 * walking an utterance's path start -> end, a run opens at the first cell of an emitting row seen while no run is open;
 * a cell of a different row closes the open run as the frames [start, c) -- c = that cell's column -- provided
 * start < c, and does not itself open a run (:96-106; so the frame on which a state is entered inside a word is
 * dropped, a word-boundary frame goes to the next word only, the final state's last run is never closed).
 * out_frame_state [N]: row_state of the run's row, | GH_SEGMENT_START on the first frame of a run (the number of runs
 * of a state is the segment count of :158-160); -1 for frames in no run. */
#define GH_SEGMENT_START (1 << 30)
int gh_align_segments(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b,
                      const int32_t* utt_lattice /*[U] or NULL*/,
                      double* out_end_cost, int32_t* out_best_end, int32_t* out_frame_state /*[N]*/);
/* The same alignment with the runs themselves as the result (round 4): out_runs [U, run_cap, 3] = (state, first frame
 * inside the utterance, frames) of every run of an utterance in time order, out_run_cnt [U] their number (above run_cap:
 * the table was too small; run_cap = the most rows a graph has is always enough).  ~N / 20 runs instead of N labels, and
 * a run's frames are contiguous in the batch -- gh_batch_gather_runs regroups them (continuous_speech.py:90-113). */
int gh_align_runs(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice /*[U] or NULL*/,
                  double* out_end_cost, int32_t* out_best_end, int run_cap, int32_t* out_runs /*[U,run_cap,3]*/,
                  int32_t* out_run_cnt /*[U]*/);

/* The same with a PACKED result: every utterance may produce up to `max_labels` labels (device-side slots), but only
 * the labels that exist come back: out_labels holds utterance 0's labels, then utterance 1's, ... (utterance u at
 * sum of out_n_labels[0..u-1]); out_capacity = entries out_labels can take (U * max_labels always suffices). */
int gh_viterbi_labels_packed(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b,
                             const int32_t* utt_lattice /*[U] or NULL*/, const int32_t* row_label, int max_labels,
                             double* out_end_cost, int32_t* out_best_end,
                             int32_t* out_labels, int64_t out_capacity, int32_t* out_n_labels);

/* ------------------------------------------------------------------ A5: dtw
 * Template DP of every utterance of an fp64 batch against the n template rows y
 * (decode.py:7-77): every origin of the previous column is a candidate (+inf arcs
 * included), first minimum wins, optional beam with the reference's "-1 mark"
 * semantics, path from (n-1, T-1) back to (0, 0), end cell excluded.
 *   var        NULL: Euclidean norm (default dist_fun); [n,D]: mahalanobis(x, y[i], var[i])
 *   trans      dense [n,n], +inf = no arc
 *   beam       <= 0: no pruning
 *   dist_host  optional caller-supplied distances (an arbitrary Python dist_fun evaluated
 *              by the caller), utterance u = [n, T_u] block at n*utt_offsets[u]; then y/var unused
 *   out_costs  same layout as dist_host (marked cells: -1 in the last column, +inf elsewhere)
 *   out_path   [N, 2] (row, col) pairs, utterance u at utt_offsets[u]; out_path_len [U]
 * n <= 1024 (one workgroup per utterance, thread = template row). */
int gh_dtw(gh_ctx* ctx, const gh_batch* b, int n, const double* y, const double* var,
           const double* trans, int beam, const double* dist_host,
           double* out_costs, int32_t* out_path, int32_t* out_path_len);

/* ---------------------------------------------- A14: k-means assignment step
 * clusters[i] = argmin_c dist(centroid_c, x_i) over frames [first, first+count) of an
 * fp64 batch (the N x k loop of kmeans.py:180-186).  var NULL: Euclidean; var [D]: the
 * shared variance cov[0] of kmeans.py:183 (mahalanobis).  First minimum wins. */
int gh_kmeans_assign(gh_ctx* ctx, const gh_batch* b, int64_t first, int64_t count, int k,
                     const double* centroids /*[k,D]*/, const double* var /*[D] or NULL*/,
                     int32_t* out_clusters /*[count]*/);

/* ------------------------------------------- A7: mixture-EM E-step statistics
 * For frames [first, first+count) of an fp64 batch and the FIRST k components given:
 * r_ic = w_c pdf_c(x_i) / sum_{c<k} w_c pdf_c(x_i)   (hmm_state.py:127-133), accumulated as
 *   out_stats[c, 0] = sum_i r_ic,  [c, 1+d] = sum_i r_ic (x_id - mean_cd),
 *   [c, 1+D+d] = sum_i r_ic (x_id - mean_cd)^2      (centred on the CURRENT mean given here)
 * i.e. the sufficient statistics of hmm_state.py:134-143,148:
 *   new mean = mean + S1/S0,  new var = S2/S0 - (S1/S0)^2,  new weight = S0/count.
 * This is the buffer the multi-GPU trainer all-reduces.  out_loglik (may be NULL) = sum_i log sum_{c<k} w_c pdf_c. */
int gh_em_accumulate(gh_ctx* ctx, const gh_batch* b, int64_t first, int64_t count, int k,
                     const double* mean /*[k,D]*/, const double* var /*[k,D]*/, const double* weight /*[k]*/,
                     double* out_stats /*[k, 1+2D]*/, double* out_loglik);

/* ------------------------------------- A9 / A11: every state refit in lock-step
 * The reference refits the states one after the other (hmm.py:97-124, continuous_speech.py:114-142): split k-means
 * (kmeans.py:167-193), then mixture EM (hmm_state.py:122-159), each iteration a pass over the state's frames.  Here
 * the frames of ALL states sit back to back in one fp64 batch -- state s owns frames [seg_off[s], seg_off[s+1]) --
 * and one call advances every state whose `active[s]` is non-zero (NULL: all): the sharded trainer runs all states in
 * lock-step with a converged mask and all-reduces one buffer per lock-step iteration.
 *
 * gh_kmeans_assign_multi: clusters_io[i] = argmin_c dist(centroids[s,c], x_i) under var[s] ([S,D]: the shared
 *   variance cov[0] of kmeans.py:183, mahalanobis; NULL: Euclidean) for every frame of every active state; frames of
 *   inactive states keep their entry.  out_changed[s] (may be NULL) = number of frames of s whose entry changed;
 *   out_sums (may be NULL) [S,k,D+1] = per cluster the sum of its frames and, in column D, their number (the centroid
 *   update of kmeans.py:158-164 as sufficient statistics), accumulated in FRAME ORDER -- the order in which
 *   np.mean(data[clusters == c], axis=0) adds the rows of a C-contiguous array -- so that sums / count is bitwise the
 *   reference's centroid for D >= 2 (with D = 1 numpy sums pairwise).
 *   clusters_io NULL: the assignments live in the batch, on the device, from one call to the next (initially -1);
 *   gh_kmeans_resident_clusters resets them to -1 (reset != 0) and / or copies them out (out != NULL, [N]).
 * gh_em_accumulate_multi: gh_em_accumulate for every active state in one launch: the first k components of state s are
 *   mean/var/weight[s, 0..k-1]; out_stats [S,k,1+2D] (centred on the means given), out_loglik [S] (either may be NULL);
 *   stats_dev (may be NULL): device buffer [S,k,1+2D] that receives the statistics, e.g. a tensor RCCL all-reduces. */
int gh_kmeans_resident_clusters(gh_ctx* ctx, gh_batch* b, int reset, int32_t* out /*[N] or NULL*/);
int gh_kmeans_assign_multi(gh_ctx* ctx, const gh_batch* b, int S, const int64_t* seg_off /*[S+1]*/,
                           const uint8_t* active /*[S] or NULL*/, int k, const double* centroids /*[S,k,D]*/,
                           const double* var /*[S,D] or NULL*/, int32_t* clusters_io /*[N]*/, int32_t* out_changed /*[S]*/,
                           double* out_sums /*[S,k,D+1]*/);
int gh_em_accumulate_multi(gh_ctx* ctx, const gh_batch* b, int S, const int64_t* seg_off /*[S+1]*/,
                           const uint8_t* active /*[S] or NULL*/, int k, const double* mean /*[S,k,D]*/,
                           const double* var /*[S,k,D]*/, const double* weight /*[S,k]*/, double* out_stats,
                           double* out_loglik, double* stats_dev);

/* ----------------------------------------------- A13: forward-backward
 * NOT in the reference (it trains by Viterbi alignment only); the sum-product twin of
 * gh_viterbi: same graphs, same same-column rule for arcs touching a non-emitting row,
 * start = the start rows in column 0, end = the end rows in the last column, log domain fp64.
 *   out_logp   [U]     log P(utterance)  (= -cost; -inf if no path)
 *   out_alpha / out_beta / out_gamma   optional [R_u, T_u] matrices at mat_off[u]:
 *              log alpha, log beta, and the posterior of passing through the cell
 *   want_occ   != 0: keep occ[n, s] = sum of gamma over the rows scored by state s, for every
 *              frame, resident in the batch (input of the Baum-Welch statistics);
 *              out_occ (may be NULL) receives a copy [N, S].
 *   out_self_xi (may be NULL) [S]: expected number of SELF transitions per state, sum over utterances and frames of
 *              xi_t(r -> r) over the emitting rows r scored by the state -- with the occupancies this re-estimates
 *              the transition costs the way continuous_speech.py:146-164 does from hard counts:
 *              p_stay = self transitions / frames in the state, p_jump = 1 - p_stay. */
int gh_forward_backward(gh_ctx* ctx, const gh_lattices* lat, gh_batch* b,
                        const int32_t* utt_lattice /*[U] or NULL*/, int want_occ,
                        double* out_logp, double* out_alpha, double* out_beta, double* out_gamma,
                        const int64_t* mat_off /*[U+1]*/, double* out_occ, double* out_self_xi /*[S] or NULL*/);

/* ------------------------------------ Baum-Welch (soft) sufficient statistics
 * After gh_forward_backward(want_occ = 1): for every frame n and state s,
 *   r_nsm = occ[n,s] * w_sm pdf_sm(x_n) / sum_m' w_sm' pdf_sm'(x_n)
 * accumulated over all frames of the batch as
 *   out_stats[s, m, 0] = sum r,  [s, m, 1+d] = sum r (x_d - mean_smd),  [s, m, 1+D+d] = sum r (x_d - mean_smd)^2
 * (the same centred layout as gh_em_accumulate; [S, M, 1+2D], 253 KB for 50 x 8 x 39).
 * States whose occupancy is <= occ_floor on a whole 128-frame tile are skipped.  Deterministic
 * (per-workgroup slabs + ordered reduction).  stats_dev (may be NULL): device pointer that
 * receives the result instead of scratch, e.g. a torch tensor about to be all-reduced by RCCL.
 * Mixtures of more than 8 components on one-word chain graphs (BASELINE configs[3]: 16 states x 32 mixtures): the
 * matrix-core statistics kernel takes the denominators sum_m' w pdf from the batch's likelihood matrix -- the one the
 * forward-backward consumed -- when that matrix was computed with THIS model handle in its present state (the library
 * keeps a stamp per model and per matrix); otherwise the generic kernel recomputes the densities.  Same results. */
int gh_bw_accumulate(gh_ctx* ctx, const gh_gmm* g, const gh_batch* b, double occ_floor,
                     double* out_stats /*[S,M,1+2D] or NULL*/, double* stats_dev /*or NULL*/);

/* --------------------------------------------- multi-GPU: the exchange step (RCCL over xGMI)
 * No reference counterpart (the reference is one process).  One process per GPU;
 * utterances are sharded, models replicated, and the ONLY data-path collective is the sum of the EM sufficient
 * statistics (the sums of hmm_state.py:134-148) over the ranks -- enqueued on the context's stream, so the kernels
 * that produce the statistics and the M-step that consumes them order themselves around it without a host sync.
 * librccl is opened on the first call (from the directory of the HIP runtime this library is bound to), never at
 * load time.
 *   gh_comm_unique_id  rank 0 makes the 128-byte id; the host side hands it to the other ranks (a TCP
 *                      socket on MASTER_ADDR, like the launcher's rendezvous)
 *   gh_comm_create     ncclCommInitRank on the context's GPU; collective over all ranks
 *   gh_stats_allreduce in-place fp64 sum of stats_dev[0..n) over the ranks, asynchronous on ctx's stream
 *   gh_comm_allreduce_host  the same for a small host buffer (sum, or max when op_max != 0), synchronous
 *   gh_comm_barrier    a one-element all-reduce + stream sync
 *   gh_comm_count      ncclCommCount: the number of ranks RCCL itself reports */
#define GH_COMM_ID_BYTES 128
typedef struct gh_comm gh_comm;
int gh_comm_unique_id(char* out_id /*[GH_COMM_ID_BYTES]*/);
int gh_comm_create(gh_ctx* ctx, int rank, int world, const char* unique_id /*[GH_COMM_ID_BYTES]*/, gh_comm** out);
void gh_comm_destroy(gh_comm* comm);
/* Failure path: ncclCommAbort -- the collectives in flight on this rank leave, the stream drains, every later call on the
 * handle returns GH_ERR_COMM (gh_comm_destroy still frees it).  A rank that fails calls it before it exits, so that its
 * peers' connections close; the peers themselves never hang either: every wait behind a collective (gh_comm_barrier,
 * gh_comm_allreduce_host, the read-back of gh_em_iteration, the polls of gh_fit_kmeans / gh_fit_em) watches RCCL's
 * asynchronous errors and a deadline of GMMHMM_COMM_TIMEOUT seconds (default 300), aborts and returns GH_ERR_COMM. */
int gh_comm_abort(gh_comm* comm);
int gh_comm_count(const gh_comm* comm);
int gh_comm_rank(const gh_comm* comm);
int gh_comm_version(void);            /* ncclGetVersion, 0 when librccl cannot be opened */
const char* gh_comm_library(void);    /* path librccl was opened from ("" when it cannot be) */
int gh_stats_allreduce(gh_ctx* ctx, gh_comm* comm, double* stats_dev, int64_t n);
int gh_comm_allreduce_host(gh_ctx* ctx, gh_comm* comm, double* host_io, int64_t n, int op_max);
int gh_comm_barrier(gh_ctx* ctx, gh_comm* comm);

/* In-place re-estimation of a packed model: the same packing as gh_gmm_create, done by kernels on the context's
 * stream into the arrays the handle already owns (a trainer updates its model every iteration; hmm_state.py:150-154
 * update_models).  Shapes are those of the handle; var == 0 is rejected as in gh_gmm_create. */
int gh_gmm_update(gh_ctx* ctx, gh_gmm* g, const double* mean /*[S,M,D]*/, const double* var /*[S,M,D]*/,
                  const double* weight /*[S,M]*/);

/* ------------------------------- soft-EM session: one DEVICE-RESIDENT, STREAM-ORDERED iteration per call
 * The Baum-Welch loop of the trainer (no reference counterpart as a whole: the reference trains by Viterbi alignment;
 * the statistics are GMM.em's, hmm_state.py:122-159, the transition update and stop rule continuous_train's in soft
 * form, continuous_speech.py:144-179) for utterances with ONE-WORD transcripts (isolated-word training, BASELINE
 * configs[2]).  Everything that does not change between iterations is built once at gh_em_create (block table of the
 * own-state likelihoods, forward-backward launch order and scratch, work lists of the statistics kernel); the model,
 * the transition costs and the convergence test live on the device.  gh_em_iteration enqueues
 *     likelihoods -> forward-backward -> statistics -> [all-reduce over `comm`] -> M-step -> model re-pack
 * on the context's stream without a host synchronisation in between:
 *   - word models: W words x n states (n <= 16, arcs from s, s-1, s-2 only), mixtures [W*n, M, D], M <= 64, D <= 40,
 *     fp64 batch; anything else: GH_ERR_UNSUPPORTED (callers keep the call-by-call path)
 *   - utt_word[u]: the word utterance u is an example of
 *   - var_floor: lower bound of re-estimated variances; occ_floor: occupancies <= it are dropped from the statistics;
 *     min_occupancy: a component keeps its parameters unless its summed responsibility exceeds it
 *   - update_transitions != 0: cost(s -> s) = -log p_stay, cost(s -> s+1) = -log(1 - p_stay),
 *     p_stay = expected self transitions / expected frames of the state (states nobody visited keep theirs)
 *   - comm (may be NULL): the packed buffer [statistics | self transitions | log P | utterances] is summed over its
 *     ranks between the statistics and the M-step, so every rank ends the iteration with the same model
 *   - out_tail (may be NULL) [4]: total log P BEFORE the update, utterances, converged (every parameter allclose to
 *     the one it replaced, numpy's default tolerances), error bits -- passing it costs one 32-byte copy + stream
 *     sync; with NULL the call returns as soon as the work is enqueued and gh_em_history reads the rows later. */
typedef struct gh_em gh_em;
int gh_em_create(gh_ctx* ctx, gh_batch* b, int W, int n, int M, const double* mean /*[W*n,M,D]*/,
                 const double* var /*[W*n,M,D]*/, const double* weight /*[W*n,M]*/, const double* word_trans /*[W,n,n]*/,
                 const int32_t* utt_word /*[U]*/, double var_floor, double occ_floor, double min_occupancy,
                 int update_transitions, gh_em** out);
/* The same session over WORD STRINGS (continuous_train's transcripts, continuous_speech.py:80-82: one word per layer of
 * the forced-alignment graph): L distinct transcripts as in gh_lattices_create_transcripts (label_off [L+1], labels),
 * utt_graph [U] = the transcript of every utterance.  Words of 2 <= n <= 8 states, transcripts of <= 16 words, at least
 * one of them longer than one word (else GH_ERR_UNSUPPORTED: gh_em_create is the form for isolated words).  The
 * forward-backward in the middle is the sequence-form kernel, the occupancies stay a [frames, states] matrix in HBM, the
 * M-step writes the new transition costs into the word templates of the graphs; everything else -- the packed buffer
 * that crosses the ranks included -- is as above, and ranks of either kind of session may share a communicator. */
int gh_em_create_transcripts(gh_ctx* ctx, gh_batch* b, int W, int n, int M, const double* mean /*[W*n,M,D]*/,
                             const double* var /*[W*n,M,D]*/, const double* weight /*[W*n,M]*/,
                             const double* word_trans /*[W,n,n]*/, int64_t L, const int64_t* label_off /*[L+1]*/,
                             const int32_t* labels, const int32_t* utt_graph /*[U]*/, double var_floor, double occ_floor,
                             double min_occupancy, int update_transitions, gh_em** out);
void gh_em_destroy(gh_em* em);
int gh_em_iteration(gh_ctx* ctx, gh_em* em, gh_comm* comm /*or NULL*/, double* out_tail /*[4] or NULL*/);
int gh_em_iterations_done(const gh_em* em);
/* Measurement aid: gh_em_profile(on) makes every following iteration record HIP events between its phases, on the stream
 * its kernels run on; gh_em_phase_ms waits for the last iteration and returns four spans in milliseconds: own-state
 * likelihoods | chain forward-backward | statistics kernel + its reduction | tail + collective + M-step + model re-pack. */
int gh_em_profile(gh_ctx* ctx, gh_em* em, int on);
int gh_em_phase_ms(gh_ctx* ctx, gh_em* em, double* out /*[4]*/);
/* rows [first, first + count) of the iteration history; the session keeps the last 4096 rows (a ring: older ones are gone) */
int gh_em_history(gh_ctx* ctx, gh_em* em, int first, int count, double* out /*[count,4]*/);
int gh_em_get_model(gh_ctx* ctx, gh_em* em, double* mean, double* var, double* weight, double* word_trans /*any may be NULL*/);
int gh_em_packed(gh_ctx* ctx, gh_em* em, double* out /*[n] or NULL*/, int64_t* out_n);

/* ----------------------- device-resident refit of ALL states: binary-split k-means + mixture EM in lock-step
 * What hmm.py:97-124 (inside HMM.fit) and continuous_speech.py:114-142 (inside continuous_train) do one state after the
 * other.  The frames of every state sit back to back in `b` (fp64; state s owns [seg_off[s], seg_off[s+1]), e.g. from
 * gh_batch_gather); tile lists, centroids, variances, mixture parameters, the allclose test's "old" parameters, the
 * active mask and the iteration counters live on the device, an iteration is a few kernel launches on the context's
 * stream, and the host reads ONE counter (states still active) every `check_every` iterations.  Per state the
 * arithmetic and the stopping rule are the sequential algorithm's.  D in [2, 64], k <= kmax <= 32.
 *   gh_fit_segment_means  out [S, D+1]: per state the sum of its frames in frame order (np.mean's order) | frame count
 *   gh_fit_kmeans         kmeans(data_s, k, centroids_s) of kmeans.py:167-193 for every state at once:
 *       part [N] (uint8): the random partition np.random.randint(0, k, N_s) of every state's frames, concatenated (the
 *         caller draws it from numpy's generator in the reference's order); partition variances = two passes per
 *         (state, cluster), ddof 1 (np.cov(...).diagonal() without the D x D matrix);
 *       assignment by mahalanobis distance under the variance of partition cluster 0 (kmeans.py:183), centroids = mean
 *         of the assigned frames in frame order, a state stops when its centroids repeat (np.array_equal);
 *       out_centroids / out_cov [S,k,D], out_counts [S,k] (cluster sizes, as doubles), out_iters [S]
 *   gh_fit_clusters       the final assignments [N] (cluster id per frame)
 *   gh_fit_em             GMM.em(data_s, k) of hmm_state.py:122-159 for every state at once: E-step statistics, then
 *       GMM.em_update per state on the device (M-step, parameters installed, allclose against mu_old / sigma_old /
 *       w_old, which are updated unless the state converged); all six arrays are in/out [S,k,D] / [S,k];
 *       n_frames [S]: frames of the state over all ranks; out_converged_at [S]: iteration of the `break`, -1: none.
 *       A zero variance is GH_ERR_INVALID ("singular").
 *   comm (may be NULL): every rank holds part of each state's frames; cluster sums / changed counts / cluster sizes /
 *       partition sums / EM statistics are summed over the ranks on the device buffers (one collective per lock-step
 *       iteration); with a communicator the partition variance is cluster 0's, from global sums, for every cluster. */
typedef struct gh_fit gh_fit;
int gh_fit_create(gh_ctx* ctx, const gh_batch* b, int S, const int64_t* seg_off /*[S+1]*/, int kmax, gh_fit** out);
void gh_fit_destroy(gh_fit* fit);
int gh_fit_segment_means(gh_ctx* ctx, gh_fit* fit, double* out /*[S,D+1]*/);
int gh_fit_kmeans(gh_ctx* ctx, gh_fit* fit, gh_comm* comm, int k, const double* centroids_in /*[S,k,D]*/,
                  const uint8_t* part /*[N]*/, int max_iteration, int check_every, double* out_centroids, double* out_cov,
                  double* out_counts, int32_t* out_iters);
int gh_fit_clusters(gh_ctx* ctx, gh_fit* fit, int32_t* out /*[N]*/);
/* Segmental k-means of MANY word models in lock-step (skmeans, kmeans.py:111-155, for every word at once; the reference
 * trains word after word, sr/core.py:57-60).  The session's "states" are then the WORDS (a word's templates back to back
 * in the batch) and the "clusters" the n segments of a word:
 *   gh_fit_set_ids      the segment of every frame [N] (the initial uniform segmentation, kmeans.py:122-127)
 *   gh_fit_dtw          dtw (decode.py:7-77, Euclidean distance, no beam) of every template against the segment means
 *                       y [W,n,D] of ITS word (utt_model [U]) under that word's transition costs trans [W,n,n], one launch;
 *                       the path stays on the device as the segment of every frame (what get_segments_from_path,
 *                       kmeans.py:98-108, counts); words whose `active` entry is 0 are skipped
 *   gh_fit_group_stats  combine_templates (kmeans.py:15-30) from the resident segment ids: per (word, segment) the mean of
 *                       its frames summed in template / frame order, the variance (two passes, ddof 1) and the frame
 *                       count; words whose `active` entry is 0 keep their previous rows */
int gh_fit_set_ids(gh_ctx* ctx, gh_fit* fit, const int32_t* ids /*[N]*/);
int gh_fit_dtw(gh_ctx* ctx, gh_fit* fit, int n, const double* y /*[S,n,D]*/, const double* trans /*[S,n,n]*/,
               const int32_t* utt_model /*[U]*/, const uint8_t* active /*[S] or NULL*/);
int gh_fit_group_stats(gh_ctx* ctx, gh_fit* fit, int k, const uint8_t* active /*[S] or NULL*/, double* out_mean /*[S,k,D]*/,
                       double* out_var /*[S,k,D]*/, double* out_count /*[S,k]*/);
int gh_fit_em(gh_ctx* ctx, gh_fit* fit, gh_comm* comm, int k, double* mean_io, double* var_io, double* weight_io,
              double* mu_old_io, double* sigma_old_io, double* w_old_io, const double* n_frames /*[S]*/, int max_iteration,
              int check_every, int32_t* out_converged_at);

#ifdef __cplusplus
}
#endif
#endif /* GMMHMM_H */
