// C ABI: a soft-EM (Baum-Welch) training session whose ITERATION IS DEVICE-RESIDENT AND STREAM-ORDERED (gh_em_*).
//
// NOT in the reference as such (it trains by Viterbi alignment); the statistics are those of GMM.em
// (hmm_state.py:122-159), the transition update and the stop rule those of continuous_train in soft form
// (continuous_speech.py:144-179).  Round 2 ran an iteration as three synchronous C-ABI calls with host work between
// them: a new packed model per E-step, graphs rebuilt from the re-estimated transition costs, the work lists of the
// fused statistics kernel rebuilt and uploaded per call, M-step and convergence test in numpy -- 2.56 ms of wall time
// around 1.31 ms of kernels.  Here everything an iteration needs that does NOT change between iterations is built once
// (block table of the own-state likelihoods, launch order / scratch offsets of the forward-backward, work lists of the
// statistics kernel, utterances grouped by word), and everything that does change lives in HBM and is rewritten by
// kernels on the context's stream:
//
//     loglik (own states) -> fb_chain (gamma compact, self transitions, log P) -> bw_fused (+ sum, re-centre)
//       -> tail (expected self transitions per state, total log P)   [packed buffer: stats | xi | log P | utterances]
//       -> ncclAllReduce(packed)                                      (gh_comm, same stream; skipped without a comm)
//       -> M-step (means / variances / weights, variance floor, transition costs -> chain costs, allclose test)
//       -> model re-pack (gh_gmm_update_dev: plain arrays + MFMA operand fragments in place)
//       -> history[it] = (log P, utterances, converged, error flags)
//
// No hipStreamSynchronize inside; the caller may fetch history[it] (one 32-byte D2H) per iteration or after many.
// Scope: one-word transcripts (isolated-word EM, BASELINE configs[2]; the configs[3] shape -- 16 states x 32 mixtures --
// too) on an fp64 batch, word models of n <= 16 states with arcs from s, s-1, s-2 only, M <= 64 mixtures, D <= 40 --
// what fb_chain_kernel / bw_fused_kernel cover; anything else returns GH_ERR_UNSUPPORTED and the caller keeps the
// call-by-call path.
//
// WORD STRINGS (gh_em_create_transcripts; continuous_train's transcripts, continuous_speech.py:80-82, in soft form): the
// same iteration with the sequence-form forward-backward in the middle --
//
//     loglik (states of the utterance's words) -> fb_seq (occupancy matrix [N, S], self transitions, log P, frame range of
//       every layer) -> ranges per (utterance, word) (bw_seq_ranges_kernel) -> bw_fused (columns by state)
//       -> tail (self transitions summed, total log P) -> all-reduce -> M-step (transition costs -> word templates) -> ...
//
// with words of n <= 8 states and transcripts of <= 16 words (what fb_seq_kernel covers).  The call-by-call form of the
// same iteration paid a graph rebuild, three synchronous calls, a host pass over 10^5 segments and a numpy M-step per
// iteration: 5.6 ms of wall time around 2.8 ms of kernels on the configs[2] shard with 7-word strings.
#include "gh_internal.h"
#include "gh_host.h"
#include "gh_fb.h"

struct gh_em {
    gh_ctx* ctx;
    gh_batch* b;
    int W, n, M, D, S;
    int64_t U, N;
    double var_floor, occ_floor, min_occ;
    int update_trans;
    gh_gmm* gmm;
    void* d_arena;
    double *d_mean, *d_var, *d_weight, *d_trans;
    gh_fbchain* d_chains;
    int32_t *d_utt_word, *d_word_utts, *d_word_off;
    int64_t* d_coff;
    double *d_alpha, *d_logp, *d_xi_utt, *d_gam;
    int32_t* d_rng;        // [U, GH_FBCHAIN_MAX, 2] frames of every chain row with gamma above the floor (fb_chain_kernel -> bw_fused_kernel)
    double* d_packed;
    int64_t n_stats, n_packed;
    int* d_flags;          // [0] entries not allclose to the previous iteration, [1] error bits (16: zero variance)
    double* d_hist;        // [hist_cap][4]: log P, utterances, converged, error bits
    int hist_cap, it;
    // Optionally (GMMHMM_EM_TWO_STREAMS=1) the utterances are split in TWO halves (alternating in length order) that run
    // on two streams, so that one half's forward-backward -- 1.5 waves per SIMD, latency bound -- could hide under the
    // other half's likelihood / statistics kernels.  Half 0 on the context's stream, half 1 on `s2`; they join before the
    // tail kernel.  Same results; measured no faster (see gh_em_create), so one half is the default.
    int n_half;
    bool norm_nll;         // mixture normalisers from the likelihood matrix (always for M > 8; GMMHMM_BWF_NORM=0: own log-sum-exp for M <= 8)
    bool use_rng;          // block lists from occupancy ranges (default) or from a pass over gamma (GMMHMM_BWF_RANGES=0)
    int lanes;             // lanes per utterance of the chain forward-backward = columns of d_gam (8, or 16 when n > 8)
    gh_loglik_plan ll_plan[2];
    bool ll_subset;
    gh_bwf_plan bw_plan[2];
    int64_t Uh[2];
    int64_t *d_perm_h[2], *d_coff_h[2];
    double* d_stats1;      // half 1's statistics [n_stats] (half 0 writes into the packed buffer)
    hipStream_t s2;
    hipEvent_t ev_start, ev_half;
    double* h_tail;        // pinned [4]
    // gh_em_profile: HIP events between the phases of an iteration (likelihoods | forward-backward | statistics | tail +
    // collective + M-step + re-pack), on the stream the kernels run on; one half only
    // word strings (gh_em_create_transcripts): the session's own transcripts handle and the buffers of the sequence form
    bool seq = false;
    gh_lattices* lat = nullptr;
    int32_t* d_utt_graph = nullptr;
    int64_t *d_soff = nullptr, *d_slot_off = nullptr;
    double *d_scratch = nullptr, *d_occ = nullptr, *d_xiparts = nullptr;
    int32_t *d_seglo = nullptr, *d_seghi = nullptr, *d_rowlo = nullptr, *d_rowhi = nullptr;
    bool occ_lds = false;
    int max_cells = 0;
    gh_comm* last_comm = nullptr;   // the communicator of the last iteration: every later wait on the stream is behind its collective
    bool prof = false;
    hipEvent_t pe[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace {

// one block per word (expected self transitions of its rows, summed over its utterances in a fixed order) + one block
// for the total log-likelihood; also clears the allclose counter of the M-step that follows
__global__ __launch_bounds__(256) void em_tail_kernel(const double* __restrict__ xi_utt, const double* __restrict__ logp,
                                                      const int32_t* __restrict__ word_utts, const int32_t* __restrict__ word_off,
                                                      int W, int n, int64_t U, double* __restrict__ tail, int* __restrict__ flags) {
    // (xi_utt rows are GH_FBCHAIN_MAX wide; only the first n entries are written by the forward-backward)
    __shared__ double red[256][GH_FBCHAIN_MAX + 1];
    const int tid = threadIdx.x;
    const int w = blockIdx.x;
    double acc[GH_FBCHAIN_MAX];
#pragma unroll
    for (int j = 0; j < GH_FBCHAIN_MAX; ++j) acc[j] = 0.0;
    if (w < W) {
        for (int i = word_off[w] + tid; i < word_off[w + 1]; i += 256) {
            const double* x = xi_utt + (int64_t)word_utts[i] * GH_FBCHAIN_MAX;
#pragma unroll
            for (int j = 0; j < GH_FBCHAIN_MAX; ++j) if (j < n) acc[j] += x[j];
        }
    } else {
        for (int64_t u = tid; u < U; u += 256) {
            const double l = logp[u];
            if (l - l == 0.0) acc[0] += l;      // finite: utterances without a path do not count (train.py e_step)
        }
        if (tid == 0) flags[0] = 0;
    }
#pragma unroll
    for (int j = 0; j < GH_FBCHAIN_MAX; ++j) red[tid][j] = acc[j];
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h)
#pragma unroll
            for (int j = 0; j < GH_FBCHAIN_MAX; ++j) red[tid][j] += red[tid + h][j];
        __syncthreads();
    }
    if (w < W) {
        if (tid < n) tail[w * n + tid] = red[0][tid];
    } else if (tid == 0) {
        tail[W * n] = red[0][0];
        tail[W * n + 1] = (double)U;
    }
}

// word strings: the self transitions arrive as GH_FBSEQ_XI_PARTS partial rows (summed here in a fixed order); block 1 is
// the total log-likelihood as above
__global__ __launch_bounds__(256) void em_tail_seq_kernel(const double* __restrict__ xi_parts, const double* __restrict__ logp,
                                                          int S, int64_t U, double* __restrict__ tail, int* __restrict__ flags) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int s = tid; s < S; s += 256) {
            double a = 0.0;
            for (int p = 0; p < GH_FBSEQ_XI_PARTS; ++p) a += xi_parts[(size_t)p * S + s];
            tail[s] = a;
        }
        return;
    }
    double acc = 0.0;
    for (int64_t u = tid; u < U; u += 256) {
        const double l = logp[u];
        if (l - l == 0.0) acc += l;
    }
    if (tid == 0) flags[0] = 0;
    red[tid] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    if (tid == 0) { tail[S] = red[0]; tail[S + 1] = (double)U; }
}

__device__ __forceinline__ bool em_close(double a, double b) {   // np.isclose(a, b) with numpy's default tolerances
    if (a == b) return true;
    if (!(a - a == 0.0) || !(b - b == 0.0)) return false;
    return fabs(a - b) <= 1e-8 + 1e-5 * fabs(b);
}

// M-step of one state per block from the (all-reduced) packed buffer: parallel.m_step + the floor / occupancy rules of
// train.BaumWelchTrainer.iteration + _update_transitions, and the allclose test against the parameters it replaces.
__global__ __launch_bounds__(256) void em_mstep_kernel(const double* __restrict__ packed, int64_t n_stats, int n, int M, int D,
                                                       double var_floor, double min_occ, int update_trans,
                                                       double* __restrict__ mean, double* __restrict__ var, double* __restrict__ weight,
                                                       double* __restrict__ trans, gh_fbchain* __restrict__ chains,
                                                       gh_seqword* __restrict__ seqwords, int* __restrict__ flags) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const int W1 = 1 + 2 * D;
    const double* st = packed + (int64_t)s * M * W1;
    double counts = 0.0;
    for (int m = 0; m < M; ++m) counts += st[m * W1];
    if (!(counts > 0)) return;                 // a state nobody visited keeps everything
    int bad = 0;
    for (int idx = tid; idx < M * D; idx += blockDim.x) {
        const int m = idx / D, d = idx - m * D;
        const double s0 = st[m * W1], S1 = st[m * W1 + 1 + d], S2 = st[m * W1 + 1 + D + d];
        const int64_t at = ((int64_t)s * M + m) * D + d;
        const double mu0 = mean[at], v0 = var[at];
        const double occ = (s0 == 0) ? 1e-5 : s0;
        const double mu = (mu0 * s0 + S1) / occ;
        const double dl = mu - mu0;
        double sg = (S2 - dl * (2.0 * S1 - dl * s0)) / occ;
        if (sg == sg && sg < var_floor) sg = var_floor;      // np.maximum(sigma, var_floor): NaN stays NaN
        const bool ok = s0 > min_occ;
        const double mu1 = ok ? mu : mu0, v1 = ok ? sg : v0;
        bad += !em_close(mu1, mu0) + !em_close(v1, v0);
        mean[at] = mu1;
        var[at] = v1;
    }
    if (tid < M) {
        const double s0 = st[tid * W1];
        const double w0 = weight[(int64_t)s * M + tid];
        const double w1 = (s0 > min_occ) ? s0 / counts : w0;
        bad += !em_close(w1, w0);
        weight[(int64_t)s * M + tid] = w1;
    }
    if (tid == 0 && update_trans) {
        // continuous_speech.py:146-164 with expected counts: p_stay = self transitions / frames of the state
        const int wi = s / n, si = s - wi * n;
        double p = packed[n_stats + s] / counts;
        if (p == p) p = fmin(fmax(p, 0.0), 1.0);
        double* t = trans + (int64_t)wi * n * n;
        gh_fbchain* ch = chains + wi;
        if (si < n - 1) {
            const double c = -log(1.0 - p);
            t[(si + 1) * n + si] = c;
            ch->next_c[si + 1] = c;
            if (seqwords) seqwords[wi].c1[si + 1] = c;
        }
        const double c = -log(p);
        t[si * n + si] = c;
        ch->self_c[si] = c;
        if (seqwords) seqwords[wi].c0[si] = c;
    }
    if (bad) atomicAdd(&flags[0], bad);
}

__global__ void em_add_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

__global__ void em_finish_kernel(const double* __restrict__ tail /* log P, utterances */, const int* __restrict__ flags,
                                 double* __restrict__ hist_row) {
    hist_row[0] = tail[0];
    hist_row[1] = tail[1];
    hist_row[2] = flags[0] == 0 ? 1.0 : 0.0;
    hist_row[3] = (double)flags[1];
}

}  // namespace

extern "C" void gh_em_destroy(gh_em* e) {
    if (!e) return;
    hipSetDevice(e->ctx->device);
    hipStreamSynchronize(e->ctx->stream);
    if (e->s2) hipStreamSynchronize(e->s2);
    for (auto& ev : e->pe) if (ev) { hipEventDestroy(ev); ev = nullptr; }
    if (e->gmm) gh_gmm_destroy(e->gmm);
    if (e->lat) gh_lattices_destroy(e->lat);
    for (int h = 0; h < 2; ++h) { gh_loglik_plan_free(&e->ll_plan[h]); gh_bwf_plan_free(&e->bw_plan[h]); }
    if (e->ev_start) hipEventDestroy(e->ev_start);
    if (e->ev_half) hipEventDestroy(e->ev_half);
    if (e->s2) hipStreamDestroy(e->s2);
    if (e->d_arena) hipFree(e->d_arena);
    if (e->h_tail) hipHostFree(e->h_tail);
    delete e;
}

extern "C" int gh_em_create(gh_ctx* ctx, gh_batch* b, int W, int n, int M, const double* mean, const double* var,
                            const double* weight, const double* word_trans, const int32_t* utt_word, double var_floor,
                            double occ_floor, double min_occupancy, int update_transitions, gh_em** out) {
    GH_REQUIRE(ctx && b && mean && var && weight && word_trans && out && (utt_word || b->U == 0), "gh_em_create: NULL argument");
    GH_REQUIRE(W > 0 && n > 0 && M > 0, "gh_em_create: W=%d n=%d M=%d", W, n, M);
    *out = nullptr;
    const int D = b->D, S = W * n;
    const int64_t U = b->U;
    // ---- what the device-resident iteration covers ----
    // (D <= 40: the fused statistics kernel is built for the operand depths of up to 40 features -- gh_bwf_plan_build)
    if (b->dtype != GH_F64 || n > GH_FBCHAIN_MAX || M > 64 || D > 40 || U > 0x7fffffff) {
        gh_set_error("gh_em_create: shape outside the device-resident path (fp64 batch, n <= 16, M <= 64, D <= 40)");
        return GH_ERR_UNSUPPORTED;
    }
    std::vector<gh_fbchain> chains(W);
    for (int w = 0; w < W; ++w) {
        gh_fbchain& fc = chains[w];
        memset(&fc, 0, sizeof fc);
        fc.n = n;
        fc.c0 = 0.0;                                    // non-emitting start row -> first state: cost 0 (continuous_speech.py:31)
        for (int i = 0; i < GH_FBCHAIN_MAX; ++i) fc.self_c[i] = fc.next_c[i] = fc.skip_c[i] = INFINITY;
        for (int i = 0; i < n; ++i) {
            fc.state[i] = w * n + i;
            for (int j = 0; j < n; ++j) {
                const double c = word_trans[((size_t)w * n + i) * n + j];
                if (std::isinf(c) && c > 0) continue;
                const int d = i - j;
                if (c != c || d < 0 || d > 2) {
                    gh_set_error("gh_em_create: word %d has an arc %d -> %d (only s, s-1, s-2 -> s are covered)", w, j, i);
                    return GH_ERR_UNSUPPORTED;
                }
                (d == 0 ? fc.self_c[i] : d == 1 ? fc.next_c[i] : fc.skip_c[i]) = c;
                if (d == 2) fc.pad = 1;
            }
        }
    }
    for (int64_t u = 0; u < U; ++u) GH_REQUIRE(utt_word[u] >= 0 && utt_word[u] < W, "gh_em_create: utt_word[%lld]=%d", (long long)u, utt_word[u]);
    GH_HIP(hipSetDevice(ctx->device));
    gh_em* e = new gh_em();
    memset((void*)e, 0, sizeof *e);
    e->ctx = ctx; e->b = b; e->W = W; e->n = n; e->M = M; e->D = D; e->S = S; e->U = U; e->N = b->N;
    e->lanes = n > 8 ? 16 : 8;
    e->use_rng = !(getenv("GMMHMM_BWF_RANGES") && !atoi(getenv("GMMHMM_BWF_RANGES")));
    e->norm_nll = M > 8 || !(getenv("GMMHMM_BWF_NORM") && !atoi(getenv("GMMHMM_BWF_NORM")));
    e->var_floor = var_floor; e->occ_floor = occ_floor; e->min_occ = min_occupancy; e->update_trans = update_transitions ? 1 : 0;
    int rc = gh_gmm_create(ctx, S, M, D, mean, var, weight, &e->gmm);
    if (rc) { gh_em_destroy(e); return rc; }
    // ---- the two halves: alternating in launch (length) order ----
    // (measured on the configs[2] shard: 1.115 ms per iteration with two halves on two streams against 1.085 ms with one --
    //  the forward-backward does not slide under the other half's kernels, and the half-sized launches are each a
    //  little less efficient; off unless GMMHMM_EM_TWO_STREAMS=1)
    e->n_half = (U >= 2048 && getenv("GMMHMM_EM_TWO_STREAMS") && atoi(getenv("GMMHMM_EM_TWO_STREAMS"))) ? 2 : 1;
    std::vector<uint8_t> half_of(std::max<int64_t>(U, 1), 0);
    std::vector<int64_t> perm_h[2];
    for (int64_t k = 0; k < U; ++k) {
        const int h = e->n_half == 2 ? (int)(k & 1) : 0;
        half_of[b->perm[k]] = (uint8_t)h;
        perm_h[h].push_back(b->perm[k]);
    }
    e->Uh[0] = (int64_t)perm_h[0].size(); e->Uh[1] = (int64_t)perm_h[1].size();
    // ---- likelihoods of every utterance's own states: persistent block tables (or the full matrix) ----
    {
        std::vector<int32_t> lo(U), hi(U);
        for (int64_t u = 0; u < U; ++u) { lo[u] = utt_word[u] * n; hi[u] = lo[u] + n; }
        e->ll_subset = true;
        for (int h = 0; h < e->n_half && U > 0; ++h) {
            std::vector<uint8_t> inc(U);
            for (int64_t u = 0; u < U; ++u) inc[u] = half_of[u] == h;
            rc = gh_loglik_plan_build(ctx, e->gmm, b, lo.data(), hi.data(), nullptr, &e->ll_plan[h], inc.data());
            if (rc < 0) { gh_em_destroy(e); return rc; }
            if (rc == 1) { e->ll_subset = false; break; }
        }
        if (!e->ll_subset) {
            e->n_half = 1;                 // (the full-matrix kernel has no notion of halves)
            e->Uh[0] = U; e->Uh[1] = 0;
            perm_h[0].assign(b->perm.begin(), b->perm.end()); perm_h[1].clear();
            std::fill(half_of.begin(), half_of.end(), 0);
            const int KS = e->gmm->KP / 2;
            if (KS != 2 && KS != 4 && KS != 8 && KS != 12 && KS != 20) {
                gh_em_destroy(e);
                gh_set_error("gh_em_create: D=%d is not a matrix-core likelihood shape", D);
                return GH_ERR_UNSUPPORTED;
            }
        }
    }
    // ---- statistics kernel: work lists built once (utterances by word, longest first), per half ----
    std::vector<std::vector<int32_t>> by_word(W);
    {
        std::vector<int64_t> seg_first(U);
        std::vector<int32_t> seg_len(U);
        for (int64_t u = 0; u < U; ++u) { seg_first[u] = b->offsets[u]; seg_len[u] = (int32_t)(b->offsets[u + 1] - b->offsets[u]); }
        for (int64_t u = 0; u < U; ++u) by_word[utt_word[u]].push_back((int32_t)u);
        for (int h = 0; h < e->n_half; ++h) {
            std::vector<std::vector<int32_t>> by_graph(W);
            for (int64_t u : perm_h[h])
                if (seg_len[u] > 0) by_graph[utt_word[u]].push_back((int32_t)u);
            rc = gh_bwf_plan_build(ctx, S, M, D, e->gmm->KP, chains, seg_first, seg_len, by_graph, /*persistent=*/true, &e->bw_plan[h]);
            if (rc) {
                gh_em_destroy(e);
                if (rc == 1) { gh_set_error("gh_em_create: shape outside the fused statistics kernel"); return GH_ERR_UNSUPPORTED; }
                return rc;
            }
        }
    }
    // ---- forward-backward: scratch offsets in launch order (one arena, disjoint pieces per slot of either half) ----
    std::vector<int64_t> coff_h[2];
    size_t cacc = 0;
    for (int h = 0; h < 2; ++h)
        for (int64_t u : perm_h[h]) {
            coff_h[h].push_back((int64_t)cacc);
            const size_t cells = (size_t)(b->offsets[u + 1] - b->offsets[u]) * n;
            cacc += gh_fbchain_scratch(cells, true) ;   // (sized for the two-way form: alpha and beta columns)
        }
    std::vector<int32_t> word_utts, word_off(W + 1, 0);
    for (int w = 0; w < W; ++w) {
        word_utts.insert(word_utts.end(), by_word[w].begin(), by_word[w].end());
        word_off[w + 1] = (int32_t)word_utts.size();
    }
    std::vector<double> trans(word_trans, word_trans + (size_t)W * n * n);
    e->n_stats = (int64_t)S * M * (1 + 2 * D);
    e->n_packed = e->n_stats + S + 2;
    e->hist_cap = 4096;
    const size_t nd = (size_t)S * M * D;
    UploadLayout lay;
    lay.add((void**)&e->d_mean, nd * 8, mean, nd * 8);
    lay.add((void**)&e->d_var, nd * 8, var, nd * 8);
    lay.add((void**)&e->d_weight, (size_t)S * M * 8, weight, (size_t)S * M * 8);
    lay.add((void**)&e->d_trans, trans.size() * 8, trans.data(), trans.size() * 8);
    lay.add((void**)&e->d_chains, (size_t)W * sizeof(gh_fbchain), chains.data(), (size_t)W * sizeof(gh_fbchain));
    lay.add((void**)&e->d_utt_word, std::max<size_t>(1, U) * 4, utt_word, (size_t)U * 4);
    lay.add((void**)&e->d_word_utts, std::max<size_t>(1, word_utts.size()) * 4, word_utts.data(), word_utts.size() * 4);
    lay.add((void**)&e->d_word_off, word_off.size() * 4, word_off.data(), word_off.size() * 4);
    for (int h = 0; h < 2; ++h) {
        lay.add((void**)&e->d_perm_h[h], std::max<size_t>(1, perm_h[h].size()) * 8, perm_h[h].data(), perm_h[h].size() * 8);
        lay.add((void**)&e->d_coff_h[h], std::max<size_t>(1, coff_h[h].size()) * 8, coff_h[h].data(), coff_h[h].size() * 8);
    }
    lay.add((void**)&e->d_stats1, (size_t)e->n_stats * 8, nullptr);
    lay.add((void**)&e->d_alpha, std::max<size_t>(1, cacc) * 8, nullptr);
    lay.add((void**)&e->d_logp, std::max<size_t>(1, U) * 8, nullptr);
    lay.add((void**)&e->d_xi_utt, std::max<size_t>(1, U) * GH_FBCHAIN_MAX * 8, nullptr);
    lay.add((void**)&e->d_gam, std::max<size_t>(1, (size_t)b->N) * e->lanes * 8, nullptr);
    lay.add((void**)&e->d_rng, std::max<size_t>(1, U) * GH_FBCHAIN_MAX * 2 * 4, nullptr);
    lay.add((void**)&e->d_packed, (size_t)e->n_packed * 8, nullptr);
    lay.add((void**)&e->d_flags, 64, nullptr);
    lay.add((void**)&e->d_hist, (size_t)e->hist_cap * 4 * 8, nullptr);
    hipError_t he = hipMalloc(&e->d_arena, lay.total);
    if (he == hipSuccess) he = hipHostMalloc((void**)&e->h_tail, 64, hipHostMallocDefault);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->s2, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_start, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_half, hipEventDisableTiming);
    if (he != hipSuccess) {
        gh_set_error("gh_em_create: %s", hipGetErrorString(he));
        gh_em_destroy(e);
        return he == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
    }
    rc = lay.commit(e->d_arena, ctx->stream, true);
    if (!rc && hipMemsetAsync(e->d_flags, 0, 64, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (!rc && hipMemsetAsync(e->d_packed, 0, (size_t)e->n_packed * 8, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (!rc) rc = gh_batch_ensure_nll(ctx, b, S, /*zero=*/true);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (rc) { gh_em_destroy(e); return rc; }
    *out = e;
    return GH_OK;
}

// The session over WORD STRINGS: L distinct transcripts (label_off / labels as in gh_lattices_create_transcripts),
// utt_graph [U] the transcript of every utterance of the batch.  Everything else as gh_em_create; the handle goes through
// the same gh_em_iteration / gh_em_history / gh_em_get_model / gh_em_packed / gh_em_destroy.
extern "C" int gh_em_create_transcripts(gh_ctx* ctx, gh_batch* b, int W, int n, int M, const double* mean, const double* var,
                                        const double* weight, const double* word_trans, int64_t L, const int64_t* label_off,
                                        const int32_t* labels, const int32_t* utt_graph, double var_floor, double occ_floor,
                                        double min_occupancy, int update_transitions, gh_em** out) {
    GH_REQUIRE(ctx && b && mean && var && weight && word_trans && out, "gh_em_create_transcripts: NULL argument");
    GH_REQUIRE(W > 0 && n > 0 && M > 0 && L >= 0, "gh_em_create_transcripts: W=%d n=%d M=%d L=%lld", W, n, M, (long long)L);
    GH_REQUIRE(b->U == 0 || (L > 0 && label_off && labels && utt_graph), "gh_em_create_transcripts: utterances without transcripts");
    *out = nullptr;
    const int D = b->D, S = W * n;
    const int64_t U = b->U;
    if (b->dtype != GH_F64 || n < 2 || n > GH_LAYERFORM_MAXN || M > 64 || D > 40 || U > 0x7fffffff) {
        gh_set_error("gh_em_create_transcripts: shape outside the device-resident path (fp64 batch, 2 <= n <= %d, M <= 64, D <= 40)",
                     GH_LAYERFORM_MAXN);
        return GH_ERR_UNSUPPORTED;
    }
    for (int64_t u = 0; u < U; ++u)
        GH_REQUIRE(utt_graph[u] >= 0 && utt_graph[u] < L, "gh_em_create_transcripts: utt_graph[%lld]=%d", (long long)u, utt_graph[u]);
    GH_HIP(hipSetDevice(ctx->device));
    gh_em* e = new gh_em();
    memset((void*)e, 0, sizeof *e);
    e->ctx = ctx; e->b = b; e->W = W; e->n = n; e->M = M; e->D = D; e->S = S; e->U = U; e->N = b->N;
    e->seq = true;
    e->lanes = 8; e->use_rng = true; e->norm_nll = true; e->n_half = 1; e->ll_subset = false;
    e->Uh[0] = U; e->Uh[1] = 0;
    e->var_floor = var_floor; e->occ_floor = occ_floor; e->min_occ = min_occupancy; e->update_trans = update_transitions ? 1 : 0;
    e->occ_lds = S <= 256;
    int rc = gh_gmm_create(ctx, S, M, D, mean, var, weight, &e->gmm);
    if (rc) { gh_em_destroy(e); return rc; }
    // ---- the graphs: the sequence form written down from the label strings (a rank without utterances has none) ----
    if (U > 0) {
        rc = gh_lattices_create_transcripts(ctx, W, n, word_trans, nullptr, L, label_off, labels, &e->lat);
        if (rc) { gh_em_destroy(e); return rc; }
        if (!e->lat->seq_ok || !e->lat->deferred_src) {
            gh_em_destroy(e);
            gh_set_error("gh_em_create_transcripts: transcripts outside the sequence form (one-word transcripts: gh_em_create; "
                         "<= %d words, arcs from s, s-1, s-2 only)", GH_SEQ_MAXK);
            return GH_ERR_UNSUPPORTED;
        }
    }
    // word -> states of the statistics kernel (the M-step keeps its transition costs in step, unused here)
    std::vector<gh_fbchain> chains(W);
    for (int w = 0; w < W; ++w) {
        gh_fbchain& fc = chains[w];
        memset(&fc, 0, sizeof fc);
        fc.n = n;
        for (int i = 0; i < GH_FBCHAIN_MAX; ++i) fc.self_c[i] = fc.next_c[i] = fc.skip_c[i] = INFINITY;
        for (int i = 0; i < n; ++i) fc.state[i] = w * n + i;
    }
    // ---- likelihoods: the states of every utterance's words (persistent block table), else the whole matrix ----
    if (U > 0) {
        std::vector<int64_t> rng_off(U + 1, 0);
        std::vector<int32_t> lo, hi;
        for (int64_t u = 0; u < U; ++u) {
            const gh_seqgraph& sg = e->lat->h_seqgraphs[utt_graph[u]];
            uint64_t seen[4] = {0, 0, 0, 0};
            for (int k = 0; k < sg.K; ++k) {
                const int w = sg.word[k];
                if (w < 256 && (seen[w >> 6] >> (w & 63) & 1)) continue;
                if (w < 256) seen[w >> 6] |= 1ull << (w & 63);
                lo.push_back(w * n); hi.push_back(w * n + n);
            }
            rng_off[u + 1] = (int64_t)lo.size();
        }
        // the table-driven kernel pays ~1.3x per tile (short runs of tiles per 32-frame block, the table itself): it wins
        // while the transcripts' words cover less than ~0.75 of the model (10 % for isolated words; 7 of 10 digits cover
        // 52 %: 1.24 ms against 1.75 ms for the whole matrix on the configs[2] shard).  GMMHMM_EM_LL=subset|full decides.
        double covered = 0.0;
        for (int64_t u = 0; u < U; ++u) covered += (double)(b->offsets[u + 1] - b->offsets[u]) * (double)(rng_off[u + 1] - rng_off[u]) * n;
        const char* ev = getenv("GMMHMM_EM_LL");
        const int KSf = e->gmm->KP / 2;
        const bool full_ok = KSf == 2 || KSf == 4 || KSf == 8 || KSf == 12 || KSf == 20;
        const bool want_full = ev ? !strcmp(ev, "full") : covered > 0.75 * (double)b->N * S;
        rc = 1;
        if (!(want_full && full_ok)) rc = gh_loglik_plan_build(ctx, e->gmm, b, lo.data(), hi.data(), rng_off.data(), &e->ll_plan[0]);
        if (rc < 0) { gh_em_destroy(e); return rc; }
        e->ll_subset = rc == 0;
        if (!e->ll_subset) {
            const int KS = e->gmm->KP / 2;
            if (KS != 2 && KS != 4 && KS != 8 && KS != 12 && KS != 20) {
                gh_em_destroy(e);
                gh_set_error("gh_em_create_transcripts: D=%d is not a matrix-core likelihood shape", D);
                return GH_ERR_UNSUPPORTED;
            }
        }
    }
    // ---- statistics kernel: one segment slot per (utterance, layer), grouped by the layer's word, longest first ----
    std::vector<int64_t> slot_off(U + 1, 0), soff(U, 0);
    size_t sacc = 0;
    {
        for (int64_t u = 0; u < U; ++u) {
            const int K = e->lat->h_seqgraphs[utt_graph[u]].K;
            slot_off[u + 1] = slot_off[u] + K;
            e->max_cells = std::max(e->max_cells, K * n);
        }
        const int64_t n_slots = slot_off[U];
        std::vector<int64_t> seg_first(n_slots);
        std::vector<int32_t> seg_len(n_slots);
        std::vector<std::vector<int32_t>> by_graph(W);
        if (n_slots > 0x7fffffff) { gh_em_destroy(e); gh_set_error("gh_em_create_transcripts: too many layers"); return GH_ERR_UNSUPPORTED; }
        for (int64_t kk = 0; kk < U; ++kk) {
            const int64_t u = b->perm[kk];
            const gh_seqgraph& sg = e->lat->h_seqgraphs[utt_graph[u]];
            const int64_t T = b->offsets[u + 1] - b->offsets[u];
            for (int k = 0; k < sg.K; ++k) {
                const int64_t sl = slot_off[u] + k;
                seg_first[sl] = b->offsets[u]; seg_len[sl] = (int32_t)T;
                if (T > 0) by_graph[sg.word[k]].push_back((int32_t)sl);
            }
            // forward-backward scratch in launch order: [T, K, n] mantissas followed by as many int32 exponents
            soff[kk] = (int64_t)sacc;
            const size_t cells = (size_t)T * sg.K * n;
            sacc += cells + (cells + 1) / 2;
        }
        rc = gh_bwf_plan_build(ctx, S, M, D, e->gmm->KP, chains, seg_first, seg_len, by_graph, /*persistent=*/true, &e->bw_plan[0]);
        if (rc) {
            gh_em_destroy(e);
            if (rc == 1) { gh_set_error("gh_em_create_transcripts: shape outside the fused statistics kernel"); return GH_ERR_UNSUPPORTED; }
            return rc;
        }
    }
    std::vector<double> trans(word_trans, word_trans + (size_t)W * n * n);
    e->n_stats = (int64_t)S * M * (1 + 2 * D);
    e->n_packed = e->n_stats + S + 2;
    e->hist_cap = 4096;
    const size_t nd = (size_t)S * M * D;
    const size_t n_slots = (size_t)slot_off[U];
    UploadLayout lay;
    lay.add((void**)&e->d_mean, nd * 8, mean, nd * 8);
    lay.add((void**)&e->d_var, nd * 8, var, nd * 8);
    lay.add((void**)&e->d_weight, (size_t)S * M * 8, weight, (size_t)S * M * 8);
    lay.add((void**)&e->d_trans, trans.size() * 8, trans.data(), trans.size() * 8);
    lay.add((void**)&e->d_chains, (size_t)W * sizeof(gh_fbchain), chains.data(), (size_t)W * sizeof(gh_fbchain));
    lay.add((void**)&e->d_utt_graph, std::max<size_t>(1, U) * 4, utt_graph, (size_t)U * 4);
    lay.add((void**)&e->d_soff, std::max<size_t>(1, U) * 8, soff.data(), (size_t)U * 8);
    lay.add((void**)&e->d_slot_off, slot_off.size() * 8, slot_off.data(), slot_off.size() * 8);
    lay.add((void**)&e->d_scratch, std::max<size_t>(1, sacc) * 8, nullptr);
    lay.add((void**)&e->d_logp, std::max<size_t>(1, U) * 8, nullptr);
    lay.add((void**)&e->d_xiparts, (size_t)GH_FBSEQ_XI_PARTS * S * 8, nullptr);
    lay.add((void**)&e->d_occ, std::max<size_t>(1, (size_t)b->N) * S * 8, nullptr);
    lay.add((void**)&e->d_seglo, std::max<size_t>(1, U) * GH_SEQ_MAXK * 4, nullptr);
    lay.add((void**)&e->d_seghi, std::max<size_t>(1, U) * GH_SEQ_MAXK * 4, nullptr);
    lay.add((void**)&e->d_rowlo, std::max<size_t>(1, U) * GH_SEQ_MAXK * GH_LAYERS_MAXN * 4, nullptr);
    lay.add((void**)&e->d_rowhi, std::max<size_t>(1, U) * GH_SEQ_MAXK * GH_LAYERS_MAXN * 4, nullptr);
    lay.add((void**)&e->d_rng, std::max<size_t>(1, n_slots) * GH_FBCHAIN_MAX * 2 * 4, nullptr);
    lay.add((void**)&e->d_packed, (size_t)e->n_packed * 8, nullptr);
    lay.add((void**)&e->d_flags, 64, nullptr);
    lay.add((void**)&e->d_hist, (size_t)e->hist_cap * 4 * 8, nullptr);
    hipError_t he = hipMalloc(&e->d_arena, lay.total);
    if (he == hipSuccess) he = hipHostMalloc((void**)&e->h_tail, 64, hipHostMallocDefault);
    if (he != hipSuccess) {
        gh_set_error("gh_em_create_transcripts: %s (%zu bytes)", hipGetErrorString(he), lay.total);
        gh_em_destroy(e);
        return he == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
    }
    rc = lay.commit(e->d_arena, ctx->stream, true);
    if (!rc && hipMemsetAsync(e->d_flags, 0, 64, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (!rc && hipMemsetAsync(e->d_packed, 0, (size_t)e->n_packed * 8, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (!rc && hipMemsetAsync(e->d_xiparts, 0, (size_t)GH_FBSEQ_XI_PARTS * S * 8, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;   // (a rank without utterances never runs the forward-backward)
    if (!rc) rc = gh_batch_ensure_nll(ctx, b, S, /*zero=*/true);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (rc) { gh_em_destroy(e); return rc; }
    *out = e;
    return GH_OK;
}

extern "C" int gh_em_iteration(gh_ctx* ctx, gh_em* e, gh_comm* comm, double* out_tail) {
    GH_REQUIRE(ctx && e && e->ctx == ctx, "gh_em_iteration: NULL argument / foreign context");
    GH_REQUIRE(!comm || gh_comm_context(comm) == ctx, "gh_em_iteration: the communicator belongs to another context (its "
               "collective must sit on the stream of the kernels around it)");
    GH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    gh_batch* b = e->b;
    GH_REQUIRE(b->N == e->N && b->U == e->U, "gh_em_iteration: the batch changed under the session");
    int rc = gh_batch_ensure_nll(ctx, b, e->S, true);
    if (rc) return rc;
    // half h: own-state likelihoods -> forward-backward -> statistics, all on stream `on`
    auto run_half = [&](int h, hipStream_t on, double* stats_out) -> int {
        hipStream_t keep = ctx->stream;
        ctx->stream = on;                      // (the launch helpers enqueue on the context's stream)
        int r = GH_OK;
        const bool pf = e->prof && e->n_half == 1;
        if (pf) hipEventRecord(e->pe[0], on);
        if (e->N > 0 && e->Uh[h] > 0) {
            r = gh_launch_loglik_mfma(ctx, e->gmm, b, nullptr, nullptr, nullptr, e->ll_subset ? &e->ll_plan[h] : nullptr);
            if (pf) hipEventRecord(e->pe[1], on);
            if (r == 1) { gh_set_error("gh_em_iteration: likelihood shape not covered"); r = GH_ERR_UNSUPPORTED; }
            if (!r && e->seq) {
                // word strings: four utterances per wave, lane = layer (gh_seq.hip); every frame's row of the occupancy
                // matrix is written whole when the S columns fit in LDS, else it is cleared and added to
                if (hipMemsetAsync(e->d_xiparts, 0, (size_t)GH_FBSEQ_XI_PARTS * e->S * 8, on) != hipSuccess) r = GH_ERR_HIP;
                if (!r && !e->occ_lds && hipMemsetAsync(e->d_occ, 0, (size_t)e->N * e->S * 8, on) != hipSuccess) r = GH_ERR_HIP;
                gh_fbseq_args q;
                memset(&q, 0, sizeof q);
                q.occ_in_lds = e->occ_lds;
                q.graphs = e->lat->d_seqgraphs; q.words = e->lat->d_seqwords; q.end_slot = e->lat->d_seq_end_slot;
                q.nll = b->nll; q.S = e->S; q.utt_off = b->d_offsets; q.utt_lat = e->d_utt_graph; q.perm = b->d_perm;
                q.alpha_scratch = e->d_scratch; q.scratch_off = e->d_soff; q.logp = e->d_logp; q.occ = e->d_occ;
                q.self_xi_parts = e->d_xiparts; q.seg_lo = e->d_seglo; q.seg_hi = e->d_seghi; q.occ_floor = e->occ_floor;
                q.max_cells = e->max_cells; q.row_lo = e->d_rowlo; q.row_hi = e->d_rowhi;
                const bool rows = gh_fb_seq_by_cell(q, e->lat->seq_N) != 0;     // (the lane = cell kernel knows every chain row's range)
                if (!r) r = gh_launch_fb_seq(ctx, q, e->lat->seq_N, e->lat->seq_skip, 0, e->U, true);
                if (!r) r = gh_bwf_seq_ranges_launch(ctx, e->lat->d_seqgraphs, e->d_utt_graph, e->d_slot_off, e->d_seglo, e->d_seghi,
                                                     rows ? e->d_rowlo : nullptr, rows ? e->d_rowhi : nullptr, e->U, e->n, e->d_rng);
            } else if (!r) {
                gh_fbchain_args ca;
                memset(&ca, 0, sizeof ca);
                ca.chains = e->d_chains; ca.nll = b->nll; ca.S = e->S; ca.utt_off = b->d_offsets; ca.utt_lat = e->d_utt_word;
                ca.perm = e->d_perm_h[h]; ca.U = e->Uh[h]; ca.alpha_scratch = e->d_alpha; ca.scratch_off = e->d_coff_h[h];
                ca.logp = e->d_logp; ca.gam = e->d_gam; ca.self_xi_utt = e->d_xi_utt; ca.lanes = e->lanes;
                ca.occ_rng = e->use_rng ? e->d_rng : nullptr; ca.rng_floor = e->occ_floor;
                r = gh_launch_fb_chain(ctx, ca, true);
            }
        } else if (pf) hipEventRecord(e->pe[1], on);
        if (pf) hipEventRecord(e->pe[2], on);
        // (the statistics kernel normalises with the likelihoods written a few lines up -- same model, same stream)
        if (!r && e->seq)
            r = gh_bwf_launch(ctx, e->bw_plan[h], e->gmm, (const double*)b->feats, e->d_occ, e->S, 1, e->occ_floor, e->d_chains,
                              stats_out, e->norm_nll ? (const double*)b->nll : nullptr, e->S, e->d_rng);
        else if (!r)
            r = gh_bwf_launch(ctx, e->bw_plan[h], e->gmm, (const double*)b->feats, e->d_gam, e->lanes, 0, e->occ_floor,
                              e->d_chains, stats_out, e->norm_nll ? (const double*)b->nll : nullptr, e->S,
                              e->use_rng ? e->d_rng : nullptr);
        if (pf) hipEventRecord(e->pe[3], on);
        ctx->stream = keep;
        return r;
    };
    if (e->n_half == 2) {
        GH_HIP(hipEventRecord(e->ev_start, st));             // behind the previous iteration's model re-pack
        GH_HIP(hipStreamWaitEvent(e->s2, e->ev_start, 0));
        rc = run_half(1, e->s2, e->d_stats1);
        if (rc) return rc;
        GH_HIP(hipEventRecord(e->ev_half, e->s2));
    }
    rc = run_half(0, st, e->d_packed);
    if (rc) return rc;
    if (e->n_half == 2) {
        GH_HIP(hipStreamWaitEvent(st, e->ev_half, 0));
        hipLaunchKernelGGL(em_add_kernel, dim3((unsigned)((e->n_stats + 255) / 256)), dim3(256), 0, st, e->d_packed, (const double*)e->d_stats1, e->n_stats);
        GH_HIP(hipGetLastError());
    }
    double* tail = e->d_packed + e->n_stats;
    if (e->seq)
        hipLaunchKernelGGL(em_tail_seq_kernel, dim3(2), dim3(256), 0, st, e->d_xiparts, e->d_logp, e->S, e->U, tail, e->d_flags);
    else
        hipLaunchKernelGGL(em_tail_kernel, dim3(e->W + 1), dim3(256), 0, st, e->d_xi_utt, e->d_logp, e->d_word_utts, e->d_word_off,
                           e->W, e->n, e->U, tail, e->d_flags);
    GH_HIP(hipGetLastError());
    e->last_comm = comm;
    if (comm) {
        rc = gh_comm_allreduce_enqueue(comm, e->d_packed, e->n_packed);   // the ONE collective of the iteration
        if (rc) return rc;
    }
    hipLaunchKernelGGL(em_mstep_kernel, dim3(e->S), dim3(256), 0, st, e->d_packed, e->n_stats, e->n, e->M, e->D, e->var_floor,
                       e->min_occ, e->update_trans, e->d_mean, e->d_var, e->d_weight, e->d_trans, e->d_chains,
                       (e->seq && e->lat) ? e->lat->d_seqwords : nullptr, e->d_flags);
    GH_HIP(hipGetLastError());
    rc = gh_gmm_update_dev(ctx, e->gmm, e->d_mean, e->d_var, e->d_weight, e->d_flags + 1);
    if (rc) return rc;
    double* row = e->d_hist + (size_t)(e->it % e->hist_cap) * 4;   // a ring: the last hist_cap iterations stay readable
    hipLaunchKernelGGL(em_finish_kernel, dim3(1), dim3(1), 0, st, tail + e->S, e->d_flags, row);
    GH_HIP(hipGetLastError());
    if (e->prof && e->n_half == 1) hipEventRecord(e->pe[4], st);
    e->it += 1;
    if (out_tail) {
        GH_HIP(hipMemcpyAsync(e->h_tail, row, 32, hipMemcpyDeviceToHost, st));
        rc = gh_stream_wait(ctx, comm, "gh_em_iteration");     // (behind the all-reduce: a lost peer is an error, not a hang)
        if (rc) return rc;
        memcpy(out_tail, e->h_tail, 32);
        if ((int)e->h_tail[3] & 16) {
            gh_set_error("gh_em_iteration: a re-estimated variance is 0 (singular covariance); raise var_floor");
            return GH_ERR_INVALID;
        }
    }
    return GH_OK;
}

extern "C" int gh_em_iterations_done(const gh_em* e) { return e ? e->it : 0; }

// Measurement aid: with on != 0 every following iteration records HIP events between its phases (on the stream the
// kernels are launched on); gh_em_phase_ms waits for the last iteration and returns the four spans in milliseconds:
// own-state likelihoods | chain forward-backward | statistics (+ its reduction) | tail + collective + M-step + re-pack.
extern "C" int gh_em_profile(gh_ctx* ctx, gh_em* e, int on) {
    GH_REQUIRE(ctx && e && e->ctx == ctx, "gh_em_profile: NULL argument / foreign context");
    GH_HIP(hipSetDevice(ctx->device));
    if (on && !e->pe[0])
        for (auto& ev : e->pe) GH_HIP(hipEventCreate(&ev));
    e->prof = on != 0;
    return GH_OK;
}

extern "C" int gh_em_phase_ms(gh_ctx* ctx, gh_em* e, double* out /*[4]*/) {
    GH_REQUIRE(ctx && e && out && e->ctx == ctx, "gh_em_phase_ms: NULL argument / foreign context");
    GH_REQUIRE(e->prof && e->pe[0] && e->it > 0 && e->n_half == 1, "gh_em_phase_ms: no profiled iteration (gh_em_profile, one stream)");
    GH_HIP(hipSetDevice(ctx->device));
    GH_HIP(hipEventSynchronize(e->pe[4]));
    for (int k = 0; k < 4; ++k) {
        float ms = 0.f;
        GH_HIP(hipEventElapsedTime(&ms, e->pe[k], e->pe[k + 1]));
        out[k] = ms;
    }
    return GH_OK;
}

extern "C" int gh_em_history(gh_ctx* ctx, gh_em* e, int first, int count, double* out) {
    GH_REQUIRE(ctx && e && out && first >= 0 && count >= 0 && first + count <= e->it, "gh_em_history: range [%d, %d) of %d",
               first, first + count, e ? e->it : 0);
    GH_REQUIRE(first >= e->it - e->hist_cap, "gh_em_history: iteration %d has left the history (the last %d of %d are kept)", first,
               e->hist_cap, e->it);
    if (count == 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    // iterations enqueued without a read-back wait HERE for their collectives -- BEFORE the copies: a device-to-host copy
    // into pageable memory blocks inside hipMemcpyAsync until the stream gets there, deadline or not
    const int rcw = gh_stream_wait(ctx, e->last_comm, "gh_em_history");
    if (rcw) return rcw;
    const int r0 = first % e->hist_cap, n0 = std::min(count, e->hist_cap - r0);      // (the range may wrap around the ring)
    GH_HIP(hipMemcpyAsync(out, e->d_hist + (size_t)r0 * 4, (size_t)n0 * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (count > n0) GH_HIP(hipMemcpyAsync(out + (size_t)n0 * 4, e->d_hist, (size_t)(count - n0) * 32, hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}

extern "C" int gh_em_get_model(gh_ctx* ctx, gh_em* e, double* mean, double* var, double* weight, double* word_trans) {
    GH_REQUIRE(ctx && e, "gh_em_get_model: NULL argument");
    GH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t nd = (size_t)e->S * e->M * e->D;
    const int rcw = gh_stream_wait(ctx, e->last_comm, "gh_em_get_model");     // (before the pageable copies: see gh_em_history)
    if (rcw) return rcw;
    if (mean) GH_HIP(hipMemcpyAsync(mean, e->d_mean, nd * 8, hipMemcpyDeviceToHost, st));
    if (var) GH_HIP(hipMemcpyAsync(var, e->d_var, nd * 8, hipMemcpyDeviceToHost, st));
    if (weight) GH_HIP(hipMemcpyAsync(weight, e->d_weight, (size_t)e->S * e->M * 8, hipMemcpyDeviceToHost, st));
    if (word_trans) GH_HIP(hipMemcpyAsync(word_trans, e->d_trans, (size_t)e->W * e->n * e->n * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

// device pointer of the packed buffer [statistics S*M*(1+2D) | expected self transitions S | log P | utterances] and its
// length: what crosses the ranks (tests compare it with the call-by-call path)
extern "C" int gh_em_packed(gh_ctx* ctx, gh_em* e, double* out /*[n] or NULL*/, int64_t* out_n) {
    GH_REQUIRE(ctx && e, "gh_em_packed: NULL argument");
    if (out_n) *out_n = e->n_packed;
    if (out) {
        GH_HIP(hipSetDevice(ctx->device));
        const int rcw = gh_stream_wait(ctx, e->last_comm, "gh_em_packed");
        if (rcw) return rcw;
        GH_HIP(hipMemcpyAsync(out, e->d_packed, (size_t)e->n_packed * 8, hipMemcpyDeviceToHost, ctx->stream));
        GH_HIP(hipStreamSynchronize(ctx->stream));
    }
    return GH_OK;
}
