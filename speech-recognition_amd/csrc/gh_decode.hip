// C ABI: launch orchestration of the dynamic programs (gh_viterbi, gh_dtw, gh_forward_backward).
#include "gh_internal.h"
#include <chrono>
#include <string>
#include "gh_host.h"
#include "gh_viterbi.h"
#include "gh_dtw.h"
#include "gh_fb.h"

// ------------------------------------------------------------------ viterbi
namespace {

// A12 on the device: one wave per utterance walks its path (stored end -> start) in start -> end order, 64
// cells at a time; a cell opens a run when its row is emitting and the cell before it is not (or it is the
// first); the run's label is written at the position given by a ballot prefix count.
__global__ __launch_bounds__(64) void path_labels_kernel(const int32_t* __restrict__ path, const int64_t* __restrict__ path_off,
                                                         const int32_t* __restrict__ path_len, const int32_t* __restrict__ utt_lat,
                                                         const gh_lattices::desc* __restrict__ descs,
                                                         const int32_t* __restrict__ row_label,
                                                         const int64_t* __restrict__ label_off, int32_t* __restrict__ labels,
                                                         int32_t* __restrict__ n_labels, int* __restrict__ flag) {
    const int64_t u = blockIdx.x;
    const int lane = threadIdx.x;
    const int len = path_len[u];
    const int32_t* p = path + 2 * path_off[u];
    const int32_t* lab = row_label + descs[utt_lat ? utt_lat[u] : 0].row_base;
    const int64_t cap = label_off[u + 1] - label_off[u];
    int32_t* out = labels + label_off[u];
    int count = 0, carry = -1;  // label of the cell before this chunk (-1: start / non-emitting)
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        const int l = (i < len) ? lab[p[2 * (int64_t)(len - 1 - i)]] : -1;
        int prev = __shfl_up(l, 1);
        if (lane == 0) prev = carry;
        const bool open = l >= 0 && prev < 0;
        const unsigned long long m = __ballot(open);
        const int pos = count + __popcll(m & ((1ull << lane) - 1));
        if (open) {
            if (pos < cap) out[pos] = l;
            else atomicOr(flag, 8);
        }
        count += __popcll(m);
        carry = __shfl(l, 63);
    }
    if (lane == 0) n_labels[u] = count;
}

// The regrouping step of continuous_train (continuous_speech.py:90-106) on the device-resident path, one lane per
// utterance: walking the path start -> end, a run opens at the first cell of an emitting row seen while no run is open;
// a cell of a DIFFERENT row closes the open run as the frames [start, c), c being that cell's column, provided
// start < c, and does not itself open a run (so the frame on which a state is entered inside a word is dropped, the
// frame on a word boundary goes to the next word only, and the final state's last run is never closed).
// frame_state[f] = state of the run frame f belongs to, | GH_SEGMENT_START on the first frame of a run; -1 otherwise.
__global__ void cut_segments_kernel(const int32_t* __restrict__ path, const int64_t* __restrict__ path_off,
                                    const int32_t* __restrict__ path_len, const int32_t* __restrict__ utt_lat,
                                    const gh_lattices::desc* __restrict__ descs, const int32_t* __restrict__ row_state,
                                    const int64_t* __restrict__ utt_off, int64_t U, int32_t* __restrict__ frame_state,
                                    int32_t* __restrict__ run_buf, int32_t* __restrict__ run_cnt, int run_cap) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int2* p = reinterpret_cast<const int2*>(path) + path_off[u];
    const int32_t* rs = row_state + descs[utt_lat ? utt_lat[u] : 0].row_base;
    int32_t* out = frame_state ? frame_state + utt_off[u] : nullptr;
    int32_t* rb = run_buf ? run_buf + u * (int64_t)run_cap * 3 : nullptr;     // (state, first frame in the utterance, frames), in time order
    int open_row = -1, open_at = -1, open_sid = -1, nr = 0;
    // the cells and their rows' states come EIGHT at a time (two round trips per eight cells instead of two per cell: the walk
    // itself is a handful of instructions, 0.51 ms of continuous_train's outer iteration were its exposed loads)
    constexpr int CB = 8;
    for (int i0 = path_len[u] - 1; i0 >= 0; i0 -= CB) {
        int2 cells[CB];
        int st[CB];
#pragma unroll
        for (int q = 0; q < CB; ++q) cells[q] = p[i0 - q >= 0 ? i0 - q : 0];
#pragma unroll
        for (int q = 0; q < CB; ++q) st[q] = rs[cells[q].x];
#pragma unroll
        for (int q = 0; q < CB; ++q) {
            if (i0 - q < 0) break;
            const int2 cell = cells[q];
            if (open_at < 0 && st[q] >= 0) { open_row = cell.x; open_at = cell.y; open_sid = st[q]; }
            if (cell.x != open_row && open_at >= 0 && open_at < cell.y) {
                const int sid = open_sid;
                if (out) {
                    out[open_at] = sid | GH_SEGMENT_START;
                    for (int f = open_at + 1; f < cell.y; ++f) out[f] = sid;
                }
                if (rb && nr < run_cap) { rb[3 * nr] = sid; rb[3 * nr + 1] = open_at; rb[3 * nr + 2] = cell.y - open_at; }
                ++nr;
                open_row = -1;
                open_at = -1;
            }
        }
    }
    if (run_cnt) run_cnt[u] = nr;
}

// label mode, packed result: utterance u's labels move from its slot (label_off[u], capacity) to packed[pack_off[u] ...]
__global__ void pack_labels_kernel(const int32_t* __restrict__ labels, const int64_t* __restrict__ label_off,
                                   const int32_t* __restrict__ n_labels, const int64_t* __restrict__ pack_off,
                                   int32_t* __restrict__ packed, int64_t U) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int32_t* src = labels + label_off[u];
    int32_t* dst = packed + pack_off[u];
    // (an utterance whose labels overflowed its slot reports count > capacity together with flag 8: copy what the slot holds)
    const int64_t cap = label_off[u + 1] - label_off[u];
    const int n = (int)(n_labels[u] < cap ? n_labels[u] : cap);
    for (int i = 0; i < n; ++i) dst[i] = src[i];
}

}  // namespace

// GMMHMM_VITERBI=lean / generic (read at every call: the parity tests switch it) keeps the form-specific kernels out:
// 1 = the row-per-lane lean kernel where it applies, 2 = the generic kernel
static int forced_kernel() {
    const char* e = getenv("GMMHMM_VITERBI");
    return !e ? 0 : !strcmp(e, "lean") ? 1 : !strcmp(e, "generic") ? 2 : 0;
}

static int viterbi_impl(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                        double* out_end_cost, int32_t* out_best_end, int32_t* out_path,
                        const int64_t* path_off, int32_t* out_path_len, double* out_costs,
                        const int64_t* costs_off, const int32_t* row_label, int32_t* out_labels,
                        const int64_t* label_off, int32_t* out_n_labels, int64_t packed_cap = -1,
                        int32_t* out_frame_state = nullptr, const gh_gmm* fused = nullptr, int fused_log_domain = 0,
                        int run_cap = 0, int32_t* out_runs = nullptr, int32_t* out_run_cnt = nullptr) {
    GH_REQUIRE(ctx && lat && b, "gh_viterbi: NULL argument");
    GH_REQUIRE(fused || b->nll || b->N == 0, "gh_viterbi: gh_loglik has not been run on this batch");
    GH_REQUIRE(!out_path || (path_off && out_path_len), "gh_viterbi: out_path needs path_off and out_path_len");
    GH_REQUIRE(!out_costs || costs_off, "gh_viterbi: out_costs needs costs_off");
    GH_HIP(hipSetDevice(ctx->device));
    // GMMHMM_HOST_TRACE=1: wall-clock phases of the call's host side on stderr (diagnostic)
    struct HostTrace {
        bool on; std::chrono::steady_clock::time_point t0, last; std::string line;
        HostTrace() : on(getenv("GMMHMM_HOST_TRACE") && atoi(getenv("GMMHMM_HOST_TRACE"))) { t0 = last = std::chrono::steady_clock::now(); }
        void mark(const char* what) {
            if (!on) return;
            const auto now = std::chrono::steady_clock::now();
            char buf[96];
            snprintf(buf, sizeof buf, " %s %.0f us |", what, std::chrono::duration<double, std::micro>(now - last).count());
            line += buf; last = now;
        }
        ~HostTrace() {
            if (on) fprintf(stderr, "[gh_viterbi host]%s total %.0f us\n", line.c_str(),
                            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    } trace;
    const int64_t U = b->U;
    if (U == 0) return GH_OK;
    const int S = fused ? fused->S : b->nll_S;
    if (lat->deferred_src) {   // a transcripts handle: everything but the sequence-form kernels runs on its expanded twin
        bool seq_fit = forced_kernel() == 0 && lat->beam <= 0 && !out_costs;
        if (b->any_T1) seq_fit = false;
        if (!seq_fit) {
            const int rc0 = gh_lattices_full(lat, &lat);
            if (rc0) return rc0;
        }
    }
    const bool want_labels = out_labels != nullptr;
    const bool want_runs = out_runs != nullptr && out_run_cnt != nullptr && run_cap > 0;
    const bool want_segments = out_frame_state != nullptr || want_runs;
    const bool uniform = utt_lattice == nullptr;
    // chain kernel: one left-to-right graph for the whole batch, no single-frame utterance (T == 1 has
    // the reference's wrap-around semantics, which only the lean / generic kernels implement)
    bool use_chain = lat->chain_ok && uniform && lat->beam <= 0;   // (a beam needs the generic kernel)
    {
        const bool no_chain = forced_kernel() != 0;
        if (no_chain) use_chain = false;
        if (b->any_T1) use_chain = false;
    }
    // fused single-Gaussian decode (gh_viterbi_fused.hip): the chain form without the [N, S] matrix
    const int fused_dv = (fused && use_chain) ? gh_fused_dv(fused->D) : 0;
    const bool use_fused = fused_dv > 0;
    // layer-form kernels: one K-layer word lattice / word-loop grammar for the whole batch (GMMHMM_VITERBI=lean /
    // generic force the others)
    bool use_layers = !use_chain && lat->layers_ok && uniform && !out_costs && lat->beam <= 0;
    {
        const bool no_layers = forced_kernel() != 0;
        if (no_layers) use_layers = false;
        if (b->any_T1) use_layers = false;   // T == 1: the reference's column wrap (lean kernel)
    }
    // sequence form (forced-alignment graphs, one per distinct transcript; gh_seq.hip)
    bool use_seq = !use_chain && !use_layers && lat->seq_ok && !out_costs && lat->beam <= 0;
    {
        const bool no_seq = forced_kernel() != 0;
        if (no_seq) use_seq = false;
        if (b->any_T1) use_seq = false;
    }
    GH_REQUIRE(!lat->deferred_src || use_seq, "gh_viterbi: internal: a transcripts handle left the sequence-form path unexpanded");
    // layer-form kernels in label mode: the back-trace writes the label sequences itself, no path is materialised
    const bool labels_direct = use_layers && want_labels && !out_path;
    std::vector<int64_t> own_path_off;  // label mode: the path lives on the device only, capacities are ours
    if ((want_labels || want_segments) && !out_path && !labels_direct) {
        own_path_off.assign(U + 1, 0);
        for (int64_t u = 0; u < U; ++u) {
            const int l = utt_lattice ? utt_lattice[u] : 0;
            GH_REQUIRE(l >= 0 && l < lat->L, "gh_viterbi: utt_lattice[%lld]=%d out of range", (long long)u, l);
            const int64_t T = b->offsets[u + 1] - b->offsets[u];
            own_path_off[u + 1] = own_path_off[u] + (T > 1 ? T * lat->lat[l].nlev : 0);
        }
        path_off = own_path_off.data();
    }
    const bool want_path = out_path != nullptr || want_labels || want_segments;
    for (int l = 0; l < lat->L; ++l)
        GH_REQUIRE(lat->lat[l].max_state < S, "gh_viterbi: graph %d uses state %d but the model has %d", l,
                   lat->lat[l].max_state, S);
    // per-utterance bookkeeping (host); with one graph for all utterances everything is implicit
    const std::vector<int64_t>& perm = b->perm;
    std::vector<int64_t> end_off;
    int64_t n_end_total = 0;
    if (uniform) {
        n_end_total = U * lat->lat[0].n_end;
    } else {
        end_off.assign(U + 1, 0);
        for (int64_t u = 0; u < U; ++u) {
            const int l = utt_lattice[u];
            GH_REQUIRE(l >= 0 && l < lat->L, "gh_viterbi: utt_lattice[%lld]=%d out of range", (long long)u, l);
            end_off[u + 1] = end_off[u] + lat->lat[l].n_end;
        }
        n_end_total = end_off[U];
    }
    if (want_path && !labels_direct)
        for (int64_t u = 0; u < U; ++u) {
            const int l = uniform ? 0 : utt_lattice[u];
            const int64_t T = b->offsets[u + 1] - b->offsets[u];
            GH_REQUIRE(path_off[u + 1] - path_off[u] >= (T > 1 ? T * lat->lat[l].nlev : 0),
                       "gh_viterbi: path capacity of utterance %lld too small", (long long)u);
        }
    // back-pointer scratch is chunked: at most gh_scratch_budget() bytes per launch (a quarter of the free HBM, <= 24 GiB
    // -- C5's 125 000 utterances need 5.1 GB of decision words: one launch on a 288 GB part; GMMHMM_SCRATCH_BUDGET
    // overrides it, the tests force several chunks with it), halved and re-planned when the allocation fails
    std::vector<int64_t> bp_off(U, 0);
    std::vector<int64_t> chunk_begin;
    size_t bp_max = 0;
    const bool want_bp = want_path || (use_chain && out_costs);
    auto plan_chunks = [&](size_t BP_BUDGET) {
        chunk_begin.assign(1, 0);
        bp_max = 0;
        if (want_bp) {
            size_t acc = 0;
            for (int64_t k = 0; k < U; ++k) {
                const int64_t u = perm[k];
                const int l = utt_lattice ? utt_lattice[u] : 0;
                // (blocks padded to 8 entries = 16 bytes: the lean kernel flushes back-pointers with 16-byte stores)
                const size_t need = use_layers ? gh_layers_bp_entries(lat->h_layers, b->offsets[u + 1] - b->offsets[u])
                                    : use_seq ? gh_seq_bp_entries(lat->seq_N, lat->seq_skip, b->offsets[u + 1] - b->offsets[u])
                                               : ((size_t)(b->offsets[u + 1] - b->offsets[u]) * lat->lat[l].R + 7) & ~size_t(7);
                if (acc && (acc + need) * 2 > BP_BUDGET) {
                    chunk_begin.push_back(k);
                    bp_max = std::max(bp_max, acc);
                    acc = 0;
                }
                bp_off[k] = (int64_t)acc;
                acc += need;
            }
            bp_max = std::max(bp_max, acc);
        }
        chunk_begin.push_back(U);
    };
    trace.mark("checks");
    const int64_t n_path = (want_path && !labels_direct) ? path_off[U] : 0;
    const int64_t n_costs = out_costs ? costs_off[U] : 0;

    gh_vit_args a;
    memset(&a, 0, sizeof a);
    int64_t *d_bpoff = nullptr, *d_endoff = nullptr, *d_pathoff = nullptr, *d_costsoff = nullptr;
    int32_t *d_uttlat = nullptr, *d_bestend, *d_path = nullptr, *d_pathlen = nullptr;
    double *d_endcost, *d_costs = nullptr;
    uint16_t* d_bp = nullptr;
    int* d_flag2;  // [flag | best_end | end_cost] are carved back to back: ONE D2H copy into pinned memory
    size_t small_bytes = 0;
    int32_t *d_framestate = nullptr, *d_runs = nullptr, *d_runcnt = nullptr;
    double* d_fpar = nullptr;
    const int fused_rp = (lat->max_R + 63) & ~63;
    int32_t *d_rowlabel = nullptr, *d_labels = nullptr, *d_nlabels = nullptr;
    int64_t *d_labeloff = nullptr, *d_poff = nullptr;
    int32_t* d_packed = nullptr;
    int64_t n_rows_total = 0;
    for (auto& lh : lat->lat) n_rows_total = std::max<int64_t>(n_rows_total, lh.row_base + lh.R);
    auto carve = [&]() -> int {
        Carver cv;
        cv.add(&d_flag2, 64); cv.add(&d_bestend, U);
        if (use_fused) cv.add(&d_fpar, (size_t)(2 * fused_dv + 2) * fused_rp);
        if (!out_end_cost) small_bytes = cv.total;       // (nobody wants the end costs: they stay on the device -- 8 MB of
        cv.add(&d_endcost, n_end_total);                 //  copy-back for 100 000 utterances x 10 word ends otherwise)
        if (out_end_cost) small_bytes = cv.total;
        if (want_bp) cv.add(&d_bpoff, U);
        if (!uniform) cv.add(&d_endoff, U + 1);
        if (utt_lattice) cv.add(&d_uttlat, U);
        if (want_path && !labels_direct) { cv.add(&d_pathoff, U + 1); cv.add(&d_path, 2 * n_path); cv.add(&d_pathlen, U); }
        if (want_bp) cv.add(&d_bp, bp_max);
        if (out_costs) { cv.add(&d_costsoff, U + 1); cv.add(&d_costs, n_costs); }
        if (out_frame_state) cv.add(&d_framestate, b->N);
        if (want_runs) { cv.add(&d_runs, (size_t)U * run_cap * 3); cv.add(&d_runcnt, U); }
        if (want_labels) {
            cv.add(&d_rowlabel, n_rows_total); cv.add(&d_labeloff, U + 1);
            cv.add(&d_nlabels, U); cv.add(&d_labels, label_off[U]);   // [n_labels | labels] back to back: one copy
            if (packed_cap >= 0) { cv.add(&d_poff, U + 1); cv.add(&d_packed, label_off[U]); }
        }
        return cv.commit(ctx);
    };
    int rc;
    for (size_t budget = gh_scratch_budget(ctx);;) {
        plan_chunks(budget);
        rc = carve();
        const size_t floor_b = (size_t)64 << 20;
        if (rc == GH_ERR_NOMEM && want_bp && bp_max * 2 > floor_b && budget > floor_b) {   // smaller chunks, same result
            budget = std::min(std::min(budget, gh_scratch_budget(ctx, /*fresh=*/true)), bp_max * 2) / 2;
            continue;
        }
        break;
    }
    trace.mark("plan+carve");
    if (rc) return rc;
    ctx->last_chunks = (int)chunk_begin.size() - 1;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemsetAsync(d_flag2, 0, sizeof(int), st));
    if (want_bp) GH_HIP(hipMemcpyAsync(d_bpoff, bp_off.data(), U * 8, hipMemcpyHostToDevice, st));
    if (!uniform) GH_HIP(hipMemcpyAsync(d_endoff, end_off.data(), (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (utt_lattice) GH_HIP(hipMemcpyAsync(d_uttlat, utt_lattice, U * 4, hipMemcpyHostToDevice, st));
    if (want_path && !labels_direct) GH_HIP(hipMemcpyAsync(d_pathoff, path_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (out_costs) GH_HIP(hipMemcpyAsync(d_costsoff, costs_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (want_labels) {
        GH_HIP(hipMemcpyAsync(d_rowlabel, row_label, n_rows_total * 4, hipMemcpyHostToDevice, st));
        GH_HIP(hipMemcpyAsync(d_labeloff, label_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    }

    trace.mark("uploads");
    a.descs = lat->d_desc; a.row_state = lat->d_row_state; a.row_start = lat->d_row_start;
    a.pred_ptr = lat->d_pred_ptr; a.pred_row = lat->d_pred_row; a.pred_cost = lat->d_pred_cost;
    a.order = lat->d_order; a.level_ptr = lat->d_level_ptr; a.end_rows = lat->d_end_rows;
    a.level_narrow = lat->d_level_narrow;
    a.nll = b->nll; a.S = S; a.r_pad = (lat->max_R + 1) & ~1;
    a.utt_off = b->d_offsets; a.utt_lat = d_uttlat; a.perm = b->d_perm;
    a.bp = d_bp; a.bp_off = d_bpoff; a.end_cost = d_endcost; a.end_off = d_endoff; a.best_end = d_bestend;
    a.path = d_path; a.path_off = d_pathoff; a.path_len = d_pathlen; a.costs = d_costs; a.costs_off = d_costsoff;
    a.flag = d_flag2;

    if (use_chain) {
        gh_fused_args fa;
        memset(&fa, 0, sizeof fa);
        gh_chain_args& c = fa.c;
        c.cost0 = lat->d_ch_cost0; c.cost1 = lat->d_ch_cost1; c.cost2 = lat->d_ch_cost2; c.row_info = lat->d_ch_info;
        c.row_state = lat->d_row_state; c.end_slot = lat->d_ch_end_slot; c.end_rows = lat->d_end_rows;
        c.group_row0 = lat->d_ch_group_row0; c.n_groups = lat->chain_groups; c.R = lat->lat[0].R; c.S = S;
        c.n_end = lat->lat[0].n_end; c.nll = b->nll; c.utt_off = b->d_offsets; c.perm = b->d_perm;
        c.bp = reinterpret_cast<uint8_t*>(d_bp); c.bp_off = d_bpoff; c.end_cost = d_endcost; c.best_end = d_bestend;
        c.path = d_path; c.path_off = d_pathoff; c.path_len = d_pathlen; c.costs = d_costs; c.costs_off = d_costsoff;
        c.flag = d_flag2;
        if (use_fused) {
            // ln 2^-1075: below it np.exp() and the weighted density round to +0 in the reference's linear domain
            const double thr = (!fused_log_domain && (ctx->compat & 1)) ? 745.1332191019412 : INFINITY;
            rc = gh_launch_fused_params(ctx, fused, lat->d_row_state, c.R, fused_rp, fused_dv, thr, d_fpar);
            if (rc) return rc;
            fa.feats = b->feats; fa.par = d_fpar; fa.D = fused->D; fa.Rp = fused_rp; fa.skip = lat->chain_skip ? 1 : 0; fa.DVp = fused_dv; fa.lin = std::isfinite(thr) ? 1 : 0;
            fa.select_end = (c.n_groups == 1 && !want_path && !out_costs) ? 1 : 0;
            fa.pw = c.n_groups == 1 ? gh_fused_window(c.R, lat->chain_unit, &fa.period) : 0;
        }
        ctx->last_fused = use_fused ? 1 : 0;
        for (size_t k = 0; k + 1 < chunk_begin.size(); ++k) {
            const int64_t u0 = chunk_begin[k], nu = chunk_begin[k + 1] - u0;
            rc = use_fused ? gh_launch_viterbi_fused(ctx, fa, u0, nu, b->dtype == GH_F64, want_bp, out_costs != nullptr)
                           : gh_launch_viterbi_chain(ctx, c, u0, nu, b->dtype == GH_F64, want_bp, out_costs != nullptr, lat->chain_skip);
            if (!rc && !(use_fused && fa.select_end)) rc = gh_launch_chain_backtrace(ctx, c, u0, nu);  // end selection (+ path when requested)
            if (rc) return rc;
        }
    }
    if (use_layers) {
        gh_layers_args c;
        memset(&c, 0, sizeof c);
        c.lf = lat->d_layers; c.end_slot = lat->d_lf_end_slot; c.end_rows = lat->d_end_rows; c.n_end = lat->lat[0].n_end; c.S = S;
        c.nll = b->nll; c.utt_off = b->d_offsets; c.perm = b->d_perm; c.bp = d_bp; c.bp_off = d_bpoff;
        c.end_cost = d_endcost; c.best_end = d_bestend; c.path = d_path; c.path_off = d_pathoff; c.path_len = d_pathlen;
        c.flag = d_flag2;
        if (labels_direct) { c.row_label = d_rowlabel; c.labels = d_labels; c.label_off = d_labeloff; c.n_labels = d_nlabels; }
        for (size_t k = 0; k + 1 < chunk_begin.size(); ++k) {
            rc = gh_launch_viterbi_layers(ctx, c, lat->h_layers, chunk_begin[k], chunk_begin[k + 1] - chunk_begin[k],
                                          b->dtype == GH_F64, want_path);
            if (!rc && want_path) rc = gh_launch_lattice_backtrace(ctx, c, lat->h_layers, chunk_begin[k], chunk_begin[k + 1] - chunk_begin[k]);
            if (rc) return rc;
        }
    }
    if (use_seq) {
        gh_layers_args c;
        memset(&c, 0, sizeof c);
        c.seqgraphs = lat->d_seqgraphs; c.seqwords = lat->d_seqwords; c.utt_lat = d_uttlat; c.end_off = d_endoff;
        c.end_slot = lat->d_seq_end_slot; c.end_rows = lat->d_end_rows; c.n_end = lat->lat[0].n_end; c.S = S;
        c.nll = b->nll; c.utt_off = b->d_offsets; c.perm = b->d_perm; c.bp = d_bp; c.bp_off = d_bpoff;
        c.end_cost = d_endcost; c.best_end = d_bestend; c.path = d_path; c.path_off = d_pathoff; c.path_len = d_pathlen;
        c.flag = d_flag2;
        for (size_t k = 0; k + 1 < chunk_begin.size(); ++k) {
            const int64_t u0 = chunk_begin[k], nu = chunk_begin[k + 1] - u0;
            rc = gh_launch_viterbi_seq(ctx, c, lat->seq_N, lat->seq_skip, u0, nu, b->dtype == GH_F64, want_path);
            if (!rc && want_path) rc = gh_launch_seq_backtrace(ctx, c, lat->seq_N, lat->seq_skip, u0, nu);
            if (rc) return rc;
        }
        use_layers = true;   // (from here on: "a lattice kernel has run", the row-per-lane kernels are skipped)
    }
    int max_level_rows = 1;
    for (auto& d : lat->h_desc) max_level_rows = std::max(max_level_rows, d.pad);
    int block = std::min(512, std::max(64, (max_level_rows + 63) & ~63));
    // lean kernel: <= 3 levels (the same number in every graph), one row per lane per level, arc lists
    // that fit LDS, no NaN arc cost, no same-column self arc (GMMHMM_VITERBI=generic forces the generic one)
    int lean_levels = 0, max_arcs = 0;
    for (auto& lh : lat->lat) max_arcs = std::max(max_arcs, lh.A);
    {
        const bool no_lean = forced_kernel() == 2;
        int lean_lanes = 1;
        for (auto& d : lat->h_desc) lean_lanes = std::max(lean_lanes, d.lean_lanes);
        const int lb = std::max(64, (lean_lanes + 63) & ~63);
        bool same_nlev = true;
        for (auto& d : lat->h_desc) same_nlev = same_nlev && d.nlev == lat->max_nlev;
        if (!no_lean && lat->beam <= 0 && !(out_costs && !want_path) && !lat->has_nan_arc && !lat->has_self_arc && same_nlev && lat->max_nlev <= 3 && lb <= 1024 &&
            S <= 8 * lb && max_arcs <= 4096) {
            lean_levels = lat->max_nlev;
            block = lb;
        }
    }
    size_t lds = 0;
    auto lean_lds = [&](int ch) {
        size_t v = (size_t)2 * a.r_pad * 8 + (size_t)2 * ch * (S + 1) * 8 + (size_t)(((max_arcs + 1) & ~1)) * 12 + 16;
        if (want_path) {  // 8 columns of back-pointers are collected in LDS and flushed with 16-byte stores
            v = (v + 15) & ~size_t(15);
            v += (size_t)8 * lat->max_R * 2 + 16;
        }
        // back-trace: 2 KB path buffer + as many back-pointer columns as fit (at least 8)
        if (want_path) v = std::max(v, (size_t)2048 + 16 + 32 + (size_t)8 * lat->max_R * 2);
        return (v + 15) & ~size_t(15);
    };
    if (lean_levels && !use_chain && !use_layers) {
        a.em_chunk = std::max(1, std::min(8, 8 * block / std::max(S, 1)));
        if (const char* e = getenv("GMMHMM_EMCHUNK")) a.em_chunk = std::max(1, std::min(a.em_chunk, atoi(e)));  // tuning knob
        while (a.em_chunk > 1 && lean_lds(a.em_chunk) > 96 * 1024) a.em_chunk >>= 1;   // keep >= 1 workgroup pair per CU
        if (lean_lds(a.em_chunk) > 160 * 1024) lean_levels = 0;                         // does not fit: generic kernel
    }
    if (lean_levels && !use_chain && !use_layers) {
        a.arc_cap = (max_arcs + 1) & ~1;
        lds = (size_t)2 * a.r_pad * 8 + (size_t)2 * a.em_chunk * (S + 1) * 8 + (size_t)a.arc_cap * 12 + 16;
        if (want_path) {
            lds = (lds + 15) & ~size_t(15);
            a.bpc_off = (int)lds;
        }
        lds = lean_lds(a.em_chunk);
        a.lds_bytes = (int)lds;
    } else if (!use_chain && !use_layers) {
        lean_levels = 0;
        int max_level_rows2 = 1;
        for (auto& d : lat->h_desc) max_level_rows2 = std::max(max_level_rows2, d.pad);
        block = std::min(512, std::max(64, (max_level_rows2 + 63) & ~63));
        a.em_chunk = 1;
        a.beam = lat->beam;
        lds = ((size_t)2 * a.r_pad + S) * sizeof(double) + (lat->beam > 0 ? (size_t)a.r_pad * 4 : 0);
    }
    if (!use_chain && !use_layers && lds > 160 * 1024) {
        gh_set_error("gh_viterbi: %d rows + %d states need %zu B of LDS (> 160 KiB)", lat->max_R, S, lds);
        return GH_ERR_UNSUPPORTED;
    }
    for (size_t c = 0; !use_chain && !use_layers && c + 1 < chunk_begin.size(); ++c) {
        a.u_begin = chunk_begin[c];
        const int64_t nu = chunk_begin[c + 1] - chunk_begin[c];
        rc = lean_levels ? gh_launch_viterbi_lean(ctx, a, nu, block, lds, b->dtype == GH_F64, want_path, lean_levels)
                         : gh_launch_viterbi(ctx, a, nu, block, lds, b->dtype == GH_F64, want_path);
        if (rc) return rc;
    }
    trace.mark("launches");
    char* pin;
    rc = gh_pinned(ctx, small_bytes, (void**)&pin);
    if (rc) return rc;
    if (want_labels) {
        if (!labels_direct) {
            hipLaunchKernelGGL(path_labels_kernel, dim3((unsigned)U), dim3(64), 0, st, d_path, d_pathoff, d_pathlen, d_uttlat,
                               lat->d_desc, d_rowlabel, d_labeloff, d_labels, d_nlabels, d_flag2);
            GH_HIP(hipGetLastError());
        }
        GH_HIP(hipMemcpyAsync(out_n_labels, d_nlabels, U * 4, hipMemcpyDeviceToHost, st));
        if (packed_cap >= 0) {
            // packed result: only the labels that exist cross the bus (a slot per utterance is sized for the longest
            // word string the grammar allows -- 40 MB of slots against 3.5 MB of labels for 125 000 loop-grammar decodes)
            GH_HIP(hipStreamSynchronize(st));
            std::vector<int64_t> poff(U + 1, 0);
            for (int64_t u = 0; u < U; ++u)
                poff[u + 1] = poff[u] + std::min<int64_t>(std::max(0, out_n_labels[u]), label_off[u + 1] - label_off[u]);
            GH_REQUIRE(poff[U] <= packed_cap, "gh_viterbi_labels_packed: %lld labels, capacity %lld", (long long)poff[U], (long long)packed_cap);
            GH_HIP(hipMemcpyAsync(d_poff, poff.data(), (size_t)(U + 1) * 8, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(pack_labels_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, d_labels, d_labeloff, d_nlabels,
                               d_poff, d_packed, U);
            GH_HIP(hipGetLastError());
            if (poff[U] > 0) GH_HIP(hipMemcpyAsync(out_labels, d_packed, (size_t)poff[U] * 4, hipMemcpyDeviceToHost, st));
            GH_HIP(hipStreamSynchronize(st));   // (poff is a host vector of this scope)
        } else if (label_off[U] > 0) GH_HIP(hipMemcpyAsync(out_labels, d_labels, label_off[U] * 4, hipMemcpyDeviceToHost, st));
    }
    if (want_segments && b->N > 0) {
        if (d_framestate) GH_HIP(hipMemsetAsync(d_framestate, 0xFF, (size_t)b->N * 4, st));
        hipLaunchKernelGGL(cut_segments_kernel, dim3((unsigned)((U + 63) / 64)), dim3(64), 0, st, d_path, d_pathoff, d_pathlen, d_uttlat,
                           lat->d_desc, lat->d_row_state, b->d_offsets, U, d_framestate, d_runs, d_runcnt, run_cap);
        GH_HIP(hipGetLastError());
        if (out_frame_state) GH_HIP(hipMemcpyAsync(out_frame_state, d_framestate, (size_t)b->N * 4, hipMemcpyDeviceToHost, st));
        if (want_runs) {
            GH_HIP(hipMemcpyAsync(out_runs, d_runs, (size_t)U * run_cap * 3 * 4, hipMemcpyDeviceToHost, st));
            GH_HIP(hipMemcpyAsync(out_run_cnt, d_runcnt, (size_t)U * 4, hipMemcpyDeviceToHost, st));
        }
    } else if (want_runs) {
        memset(out_run_cnt, 0, (size_t)U * 4);
    }
    trace.mark("labels");
    GH_HIP(hipMemcpyAsync(pin, d_flag2, small_bytes, hipMemcpyDeviceToHost, st));
    if (out_path) {
        GH_HIP(hipMemcpyAsync(out_path, d_path, 2 * n_path * 4, hipMemcpyDeviceToHost, st));
        GH_HIP(hipMemcpyAsync(out_path_len, d_pathlen, U * 4, hipMemcpyDeviceToHost, st));
    }
    if (out_costs) GH_HIP(hipMemcpyAsync(out_costs, d_costs, n_costs * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    trace.mark("final sync");
    const int flag = *reinterpret_cast<int*>(pin);
    if (out_best_end) memcpy(out_best_end, pin + ((char*)d_bestend - (char*)d_flag2), U * 4);
    if (out_end_cost) memcpy(out_end_cost, pin + ((char*)d_endcost - (char*)d_flag2), n_end_total * 8);
    if (flag & 1) {
        gh_set_error("gh_viterbi: a DP cell chose itself as its origin (decode.py:120-121)");
        return GH_ERR_SELF_POINTER;
    }
    if (flag & 4) {
        gh_set_error("gh_viterbi: back-trace does not terminate (cycle of same-column arcs between unreachable cells)");
        return GH_ERR_INVALID;
    }
    if (flag & 2) {
        gh_set_error("gh_viterbi: back-trace reached a cell without predecessor");
        return GH_ERR_INVALID;
    }
    if (flag & 8) {
        gh_set_error("gh_viterbi_labels: label capacity of an utterance too small");
        return GH_ERR_INVALID;
    }
    return GH_OK;
}

extern "C" int gh_viterbi(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                          double* out_end_cost, int32_t* out_best_end, int32_t* out_path,
                          const int64_t* path_off, int32_t* out_path_len, double* out_costs,
                          const int64_t* costs_off) {
    return viterbi_impl(ctx, lat, b, utt_lattice, out_end_cost, out_best_end, out_path, path_off, out_path_len, out_costs,
                        costs_off, nullptr, nullptr, nullptr, nullptr);
}

// A5 + A2 / A6 + A3 in one sweep for models with ONE Gaussian per state (HMM.evaluate, hmm.py:131-135).  Shapes the fused
// kernel does not take (several components, graphs that are not left-to-right chains, single-frame utterances, a beam,
// D > 40) run as gh_loglik followed by gh_viterbi: same results, two kernels.
extern "C" int gh_viterbi_fused(gh_ctx* ctx, const gh_gmm* g, const gh_lattices* lat, gh_batch* b, int log_domain,
                                double* out_end_cost, int32_t* out_best_end, int32_t* out_path, const int64_t* path_off,
                                int32_t* out_path_len, double* out_costs, const int64_t* costs_off) {
    GH_REQUIRE(ctx && g && lat && b, "gh_viterbi_fused: NULL argument");
    GH_REQUIRE(g->D == b->D || b->N == 0, "gh_viterbi_fused: model has %d dimensions, batch %d", g->D, b->D);
    ctx->last_fused = 0;
    bool can = g->M == 1 && lat->chain_ok && lat->beam <= 0 && forced_kernel() == 0 && gh_fused_dv(g->D) > 0 && !lat->deferred_src;
    if (const char* e = getenv("GMMHMM_FUSED")) can = can && atoi(e) != 0;
    if (b->any_T1) can = false;   // T == 1: the reference's column wrap (other kernels)
    if (!can) {
        const int compat = ctx->compat;
        if (log_domain) ctx->compat &= ~1;     // mahalanobis() is a log-domain distance: it never underflows
        const int rc = gh_loglik(ctx, g, b, nullptr);
        ctx->compat = compat;
        if (rc) return rc;
    }
    return viterbi_impl(ctx, lat, b, nullptr, out_end_cost, out_best_end, out_path, path_off, out_path_len, out_costs, costs_off,
                        nullptr, nullptr, nullptr, nullptr, -1, nullptr, can ? g : nullptr, log_domain);
}

extern "C" int gh_ctx_last_fused(const gh_ctx* ctx) { return ctx ? ctx->last_fused : -1; }

extern "C" int gh_viterbi_labels(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                                 const int32_t* row_label, double* out_end_cost, int32_t* out_best_end,
                                 int32_t* out_labels, const int64_t* label_off, int32_t* out_n_labels) {
    GH_REQUIRE(row_label && out_labels && label_off && out_n_labels, "gh_viterbi_labels: NULL argument");
    return viterbi_impl(ctx, lat, b, utt_lattice, out_end_cost, out_best_end, nullptr, nullptr, nullptr, nullptr, nullptr,
                        row_label, out_labels, label_off, out_n_labels);
}

// ---------------------------------------------------------------------- dtw
extern "C" int gh_dtw(gh_ctx* ctx, const gh_batch* b, int n, const double* y, const double* var,
                      const double* trans, int beam, const double* dist_host, double* out_costs,
                      int32_t* out_path, int32_t* out_path_len) {
    GH_REQUIRE(ctx && b && trans, "gh_dtw: NULL argument");
    GH_REQUIRE(dist_host || y, "gh_dtw: need template rows or a distance matrix");
    GH_REQUIRE(n > 1 && n <= 1024, "gh_dtw: n=%d (2..1024 supported; decode.py:22 asserts n > 1)", n);
    GH_REQUIRE(b->dtype == GH_F64 || dist_host, "gh_dtw: built-in distances need an fp64 batch");
    GH_REQUIRE(!out_path || out_path_len, "gh_dtw: out_path needs out_path_len");
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t U = b->U, N = b->N;
    const int D = b->D;
    if (U == 0) return GH_OK;
    for (int64_t u = 0; u < U; ++u)
        GH_REQUIRE(b->offsets[u + 1] - b->offsets[u] > 1, "gh_dtw: utterance %lld has fewer than 2 frames (decode.py:22)",
                   (long long)u);
    std::vector<double> logdet(n, 0.0);
    if (var && !dist_host)
        for (int i = 0; i < n; ++i) {
            double prod = 1.0;
            for (int d = 0; d < D; ++d) prod *= var[(size_t)i * D + d];
            logdet[i] = 0.5 * std::log(std::pow(2.0 * M_PI, D) * prod);  // hmm_state.py:58
        }
    std::vector<int64_t> moff(U + 1);  // n * frame offset: [n,T] blocks
    for (int64_t u = 0; u <= U; ++u) moff[u] = (int64_t)n * b->offsets[u];
    double *d_y = nullptr, *d_var = nullptr, *d_ld = nullptr, *d_tr, *d_E = nullptr, *d_costs = nullptr;
    int64_t* d_moff;
    uint8_t* d_bp;
    int32_t *d_path = nullptr, *d_plen = nullptr;
    Carver cv;
    cv.add(&d_tr, (size_t)n * n); cv.add(&d_moff, U + 1); cv.add(&d_bp, (size_t)n * N * (n <= 255 ? 1 : 2));
    if (!dist_host) { cv.add(&d_y, (size_t)n * D); if (var) { cv.add(&d_var, (size_t)n * D); cv.add(&d_ld, n); } }
    if (dist_host) cv.add(&d_E, (size_t)n * N);
    if (out_costs) cv.add(&d_costs, (size_t)n * N);
    if (out_path) { cv.add(&d_path, 2 * (size_t)N); cv.add(&d_plen, U); }
    int rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), st));
    GH_HIP(hipMemcpyAsync(d_tr, trans, (size_t)n * n * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_moff, moff.data(), (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (d_y) GH_HIP(hipMemcpyAsync(d_y, y, (size_t)n * D * 8, hipMemcpyHostToDevice, st));
    if (d_var) {
        GH_HIP(hipMemcpyAsync(d_var, var, (size_t)n * D * 8, hipMemcpyHostToDevice, st));
        GH_HIP(hipMemcpyAsync(d_ld, logdet.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    if (d_E) GH_HIP(hipMemcpyAsync(d_E, dist_host, (size_t)n * N * 8, hipMemcpyHostToDevice, st));
    gh_dtw_args a;
    memset(&a, 0, sizeof a);
    a.x = dist_host ? nullptr : (const double*)b->feats;
    a.E = d_E; a.e_off = d_moff; a.utt_off = b->d_offsets; a.n = n; a.D = D; a.beam = beam;
    a.y = d_y; a.var = d_var; a.logdet = d_ld; a.trans = d_tr; a.bp = d_bp; a.bp_off = d_moff;
    a.costs = d_costs; a.costs_off = d_moff; a.path = d_path; a.path_off = b->d_offsets; a.path_len = d_plen;
    a.flag = ctx->d_flag;
    rc = gh_launch_dtw(ctx, a, U);
    if (rc) return rc;
    int flag = 0;
    GH_HIP(hipMemcpyAsync(&flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    if (out_costs) GH_HIP(hipMemcpyAsync(out_costs, d_costs, (size_t)n * N * 8, hipMemcpyDeviceToHost, st));
    if (out_path) {
        GH_HIP(hipMemcpyAsync(out_path, d_path, 2 * (size_t)N * 4, hipMemcpyDeviceToHost, st));
        GH_HIP(hipMemcpyAsync(out_path_len, d_plen, U * 4, hipMemcpyDeviceToHost, st));
    }
    GH_HIP(hipStreamSynchronize(st));
    if (flag & 8) {
        gh_set_error("gh_dtw: the beam pruned every origin of a column (np.argmin of an empty list)");
        return GH_ERR_INVALID;
    }
    if (flag & 2) {
        gh_set_error("gh_dtw: back-trace left the matrix: cell (0,0) is not reachable from the end cell");
        return GH_ERR_INVALID;
    }
    return GH_OK;
}

// --------------------------------------------------------- forward-backward
extern "C" int gh_forward_backward(gh_ctx* ctx, const gh_lattices* lat, gh_batch* b, const int32_t* utt_lattice,
                                   int want_occ, double* out_logp, double* out_alpha, double* out_beta,
                                   double* out_gamma, const int64_t* mat_off, double* out_occ, double* out_self_xi) {
    GH_REQUIRE(ctx && lat && b, "gh_forward_backward: NULL argument");
    GH_REQUIRE(b->nll || b->N == 0, "gh_forward_backward: gh_loglik has not been run on this batch");
    GH_REQUIRE(!(out_alpha || out_beta || out_gamma) || mat_off, "gh_forward_backward: matrices need mat_off");
    GH_REQUIRE(!out_occ || want_occ, "gh_forward_backward: out_occ needs want_occ");
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t U = b->U;
    if (U == 0) return GH_OK;
    const int S = b->nll_S;
    if (want_occ) b->seq_seg_valid = false;
    if (lat->deferred_src) {   // a transcripts handle: everything but the sequence-form kernel runs on its expanded twin
        const char* e = getenv("GMMHMM_FB");
        if (out_alpha || out_beta || out_gamma || (e && !strcmp(e, "generic")) || lat->seq_N > GH_LAYERFORM_MAXN) {   // (12 / 16 states per word: Viterbi only)
            const int rc0 = gh_lattices_full(lat, &lat);
            if (rc0) return rc0;
        }
    }
    for (int l = 0; l < lat->L; ++l)
        GH_REQUIRE(lat->lat[l].max_state < S, "gh_forward_backward: graph %d uses state %d but the model has %d", l,
                   lat->lat[l].max_state, S);
    if (utt_lattice)
        for (int64_t u = 0; u < U; ++u)
            GH_REQUIRE(utt_lattice[u] >= 0 && utt_lattice[u] < lat->L, "gh_forward_backward: utt_lattice[%lld] out of range",
                       (long long)u);
    // chain graphs and nobody wants the [N, S] matrix on the host: gamma stays compact ([N, 8] or [N, 16]) for the fused statistics
    // kernel (GMMHMM_BW=generic keeps the full occupancy matrix and the generic statistics kernel)
    bool compact_gam = false;
    {
        const char* e = getenv("GMMHMM_FB");
        const char* e2 = getenv("GMMHMM_BW");
        const bool mats_ = out_alpha || out_beta || out_gamma;
        compact_gam = want_occ && !out_occ && lat->fbchain_ok && !mats_ && !(e && !strcmp(e, "generic")) && !(e2 && !strcmp(e2, "generic")) &&
                      b->dtype == GH_F64;
    }
    const int chain_lanes = lat->fbchain_ok ? gh_fbchain_lanes(lat->h_fbchain) : 8;
    if (compact_gam) {
        if (b->gam && b->gam_lanes != chain_lanes) { GH_HIP(hipStreamSynchronize(ctx->stream)); GH_HIP(hipFree(b->gam)); b->gam = nullptr; }
        if (!b->gam && b->N > 0) { GH_HIP(hipMalloc((void**)&b->gam, (size_t)b->N * chain_lanes * 8)); b->gam_lanes = chain_lanes; }
        b->gam_chains = lat->h_fbchain;
        b->gam_utt_graph.assign(utt_lattice ? utt_lattice : nullptr, utt_lattice ? utt_lattice + U : nullptr);
        b->occ_valid = false;
    } else if (want_occ) {
        b->gam_chains.clear();
    }
    if (want_occ && !compact_gam && b->N > 0 && (!b->occ || b->occ_S != S)) {   // sized for the model in use, not for the first one seen
        if (b->occ) { GH_HIP(hipStreamSynchronize(ctx->stream)); GH_HIP(hipFree(b->occ)); b->occ = nullptr; }
        if (b->d_occ_states) { GH_HIP(hipFree(b->d_occ_states)); b->d_occ_states = nullptr; }
        GH_HIP(hipMalloc((void**)&b->occ, (size_t)b->N * S * 8));
        b->occ_S = S;
    }
    // alpha scratch, chunked (<= gh_scratch_budget() bytes per launch), launch order = longest first
    const size_t BUDGET = gh_scratch_budget(ctx);
    const bool use_fbseq = [&] { const char* e = getenv("GMMHMM_FB"); return lat->seq_ok && lat->seq_N <= GH_LAYERFORM_MAXN && !(out_alpha || out_beta || out_gamma) && !(e && !strcmp(e, "generic")); }();
    GH_REQUIRE(!lat->deferred_src || use_fbseq, "gh_forward_backward: internal: a transcripts handle left the sequence-form path unexpanded");
    std::vector<int64_t> soff(U, 0), chunk_begin{0};
    size_t acc = 0, smax = 0;
    for (int64_t k = 0; k < U; ++k) {
        const int64_t u = b->perm[k];
        const int l = utt_lattice ? utt_lattice[u] : 0;
        size_t need = (size_t)(b->offsets[u + 1] - b->offsets[u]) * lat->lat[l].R;
        if (use_fbseq) {   // [T, K, N] mantissas (double) followed by as many exponents (int32)
            const size_t cells = (size_t)(b->offsets[u + 1] - b->offsets[u]) * lat->h_seqgraphs[l].K * lat->seq_N;
            need = cells + (cells + 1) / 2;
        }
        if (acc && (acc + need) * 8 > BUDGET) { chunk_begin.push_back(k); smax = std::max(smax, acc); acc = 0; }
        soff[k] = (int64_t)acc;
        acc += need;
    }
    smax = std::max(smax, acc);
    chunk_begin.push_back(U);
    ctx->last_chunks = (int)chunk_begin.size() - 1;
    const bool mats = out_alpha || out_beta || out_gamma;
    const int64_t n_mat = mats ? mat_off[U] : 0;
    // one-word chain graphs and no matrices asked for: one lane per utterance (GMMHMM_FB=generic forces the other)
    {
        const char* e = getenv("GMMHMM_FB");
        if (lat->fbchain_ok && !mats && !(e && !strcmp(e, "generic"))) {
            ctx->last_chunks = 1;     // (one lane group per utterance, 12 bytes of alpha per cell: never chunked)
            std::vector<int64_t> coff(U, 0);
            size_t cacc = 0;
            for (int64_t k = 0; k < U; ++k) {
                const int64_t u = b->perm[k];
                coff[k] = (int64_t)cacc;
                const size_t cells = (size_t)(b->offsets[u + 1] - b->offsets[u]) * lat->h_fbchain[utt_lattice ? utt_lattice[u] : 0].n;
                cacc += gh_fbchain_scratch(cells, gh_fbchain_two_way(ctx, compact_gam, want_occ && !compact_gam, U, chain_lanes));
            }
            int64_t* d_coff;
            int32_t* d_ul = nullptr;
            double *d_al, *d_lp;
            Carver cc;
            double* d_xi = nullptr;
            cc.add(&d_coff, U); cc.add(&d_lp, U); cc.add(&d_al, cacc);
            if (out_self_xi) cc.add(&d_xi, (size_t)U * GH_FBCHAIN_MAX);
            if (utt_lattice) cc.add(&d_ul, U);
            int rc2 = cc.commit(ctx);
            if (rc2) return rc2;
            hipStream_t s2 = ctx->stream;
            GH_HIP(hipMemcpyAsync(d_coff, coff.data(), U * 8, hipMemcpyHostToDevice, s2));
            if (utt_lattice) GH_HIP(hipMemcpyAsync(d_ul, utt_lattice, U * 4, hipMemcpyHostToDevice, s2));
            if (want_occ && !compact_gam && b->N > 0) GH_HIP(hipMemsetAsync(b->occ, 0, (size_t)b->N * S * 8, s2));
            if (want_occ && !compact_gam) {   // the only states that can carry occupancy: each utterance's chain
                std::vector<int32_t> st((size_t)U * GH_FBCHAIN_MAX, -1);
                for (int64_t u = 0; u < U; ++u) {
                    const gh_fbchain& fc = lat->h_fbchain[utt_lattice ? utt_lattice[u] : 0];
                    for (int j = 0; j < fc.n; ++j) st[(size_t)u * GH_FBCHAIN_MAX + j] = fc.state[j];
                }
                if (!b->d_occ_states) GH_HIP(hipMalloc((void**)&b->d_occ_states, st.size() * 4));
                GH_HIP(hipMemcpy(b->d_occ_states, st.data(), st.size() * 4, hipMemcpyHostToDevice));
            }
            gh_fbchain_args ca;
            memset(&ca, 0, sizeof ca);
            ca.chains = lat->d_fbchain; ca.nll = b->nll; ca.S = S; ca.utt_off = b->d_offsets; ca.utt_lat = d_ul;
            ca.perm = b->d_perm; ca.U = U; ca.alpha_scratch = d_al; ca.scratch_off = d_coff; ca.logp = d_lp;
            ca.occ = (want_occ && !compact_gam) ? b->occ : nullptr;
            ca.gam = compact_gam ? b->gam : nullptr;
            if (want_occ && !compact_gam) b->occ_valid = true;
            ca.self_xi_utt = d_xi;
            ca.lanes = chain_lanes;
            rc2 = gh_launch_fb_chain(ctx, ca, b->dtype == GH_F64);
            if (rc2) return rc2;
            if (out_logp) GH_HIP(hipMemcpyAsync(out_logp, d_lp, U * 8, hipMemcpyDeviceToHost, s2));
            if (out_occ) GH_HIP(hipMemcpyAsync(out_occ, b->occ, (size_t)b->N * S * 8, hipMemcpyDeviceToHost, s2));
            std::vector<double> xi_utt;
            if (out_self_xi) {
                xi_utt.resize((size_t)U * GH_FBCHAIN_MAX);
                GH_HIP(hipMemcpyAsync(xi_utt.data(), d_xi, xi_utt.size() * 8, hipMemcpyDeviceToHost, s2));
            }
            GH_HIP(hipStreamSynchronize(s2));
            if (out_self_xi) {   // per state, summed in utterance order (deterministic)
                for (int s = 0; s < S; ++s) out_self_xi[s] = 0.0;
                for (int64_t u = 0; u < U; ++u) {
                    const gh_fbchain& fc = lat->h_fbchain[utt_lattice ? utt_lattice[u] : 0];
                    for (int j = 0; j < fc.n; ++j) out_self_xi[fc.state[j]] += xi_utt[(size_t)u * GH_FBCHAIN_MAX + j];
                }
            }
            return GH_OK;
        }
    }
    int64_t *d_soff, *d_matoff = nullptr;
    int32_t* d_uttlat = nullptr;
    double *d_scratch, *d_logp, *d_alpha = nullptr, *d_beta = nullptr, *d_gamma = nullptr;
    Carver cv;
    double* d_selfxi = nullptr;
    cv.add(&d_soff, U); cv.add(&d_logp, U); cv.add(&d_scratch, smax);
    double* d_xiparts = nullptr;
    int32_t *d_seglo = nullptr, *d_seghi = nullptr;
    if (use_fbseq && want_occ) { cv.add(&d_seglo, (size_t)U * GH_SEQ_MAXK); cv.add(&d_seghi, (size_t)U * GH_SEQ_MAXK); }
    if (out_self_xi) cv.add(&d_selfxi, S);
    if (out_self_xi && use_fbseq) cv.add(&d_xiparts, (size_t)GH_FBSEQ_XI_PARTS * S);
    if (utt_lattice) cv.add(&d_uttlat, U);
    if (mats) cv.add(&d_matoff, U + 1);
    if (out_alpha) cv.add(&d_alpha, n_mat);
    if (out_beta) cv.add(&d_beta, n_mat);
    if (out_gamma) cv.add(&d_gamma, n_mat);
    int rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(d_soff, soff.data(), U * 8, hipMemcpyHostToDevice, st));
    if (utt_lattice) GH_HIP(hipMemcpyAsync(d_uttlat, utt_lattice, U * 4, hipMemcpyHostToDevice, st));
    if (mats) GH_HIP(hipMemcpyAsync(d_matoff, mat_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    gh_fb_args a;
    memset(&a, 0, sizeof a);
    a.descs = lat->d_desc; a.row_state = lat->d_row_state; a.row_flag = lat->d_row_start;
    a.pred_ptr = lat->d_pred_ptr; a.pred_row = lat->d_pred_row; a.pred_cost = lat->d_pred_cost;
    a.succ_ptr = lat->d_succ_ptr; a.succ_row = lat->d_succ_row; a.succ_cost = lat->d_succ_cost;
    a.order = lat->d_order; a.level_ptr = lat->d_level_ptr; a.end_rows = lat->d_end_rows;
    a.nll = b->nll; a.S = S; a.r_pad = (lat->max_R + 1) & ~1;
    a.utt_off = b->d_offsets; a.utt_lat = d_uttlat; a.perm = b->d_perm;
    a.alpha_scratch = d_scratch; a.scratch_off = d_soff; a.logp = d_logp;
    a.out_alpha = d_alpha; a.out_beta = d_beta; a.out_gamma = d_gamma; a.mat_off = d_matoff;
    a.occ = want_occ ? b->occ : nullptr;
    if (want_occ) b->occ_valid = true;
    a.self_xi = d_selfxi;
    if (d_selfxi) GH_HIP(hipMemsetAsync(d_selfxi, 0, (size_t)S * 8, st));
    if (want_occ && b->d_occ_states) { GH_HIP(hipFree(b->d_occ_states)); b->d_occ_states = nullptr; }   // any state may be occupied
    // forced-alignment graphs (one word per layer): four utterances per wave, lane = layer (GMMHMM_FB=generic forces the other)
    {
        if (use_fbseq) {
            if (out_self_xi) GH_HIP(hipMemsetAsync(d_xiparts, 0, (size_t)GH_FBSEQ_XI_PARTS * S * 8, st));
            const bool occ_lds = S <= 256;   // (a word may stand in several layers of a transcript: their occupancies add up)
            if (want_occ && !occ_lds && b->N > 0) GH_HIP(hipMemsetAsync(b->occ, 0, (size_t)b->N * S * 8, st));
            gh_fbseq_args q;
            memset(&q, 0, sizeof q);
            q.occ_in_lds = occ_lds;
            q.graphs = lat->d_seqgraphs; q.words = lat->d_seqwords; q.end_slot = lat->d_seq_end_slot; q.nll = b->nll; q.S = S;
            q.utt_off = b->d_offsets; q.utt_lat = d_uttlat; q.perm = b->d_perm; q.alpha_scratch = d_scratch; q.scratch_off = d_soff;
            q.logp = d_logp; q.occ = want_occ ? b->occ : nullptr; q.self_xi_parts = d_xiparts;
            q.seg_lo = d_seglo; q.seg_hi = d_seghi; q.occ_floor = 0.0;
            for (const gh_seqgraph& sg : lat->h_seqgraphs) q.max_cells = std::max(q.max_cells, sg.K * lat->seq_N);
            for (size_t c = 0; c + 1 < chunk_begin.size(); ++c) {
                rc = gh_launch_fb_seq(ctx, q, lat->seq_N, lat->seq_skip, chunk_begin[c], chunk_begin[c + 1] - chunk_begin[c], b->dtype == GH_F64);
                if (rc) return rc;
            }
            if (out_logp) GH_HIP(hipMemcpyAsync(out_logp, d_logp, U * 8, hipMemcpyDeviceToHost, st));
            if (out_occ) GH_HIP(hipMemcpyAsync(out_occ, b->occ, (size_t)b->N * S * 8, hipMemcpyDeviceToHost, st));
            if (want_occ) {   // what the fused statistics kernel needs to walk (utterance, layer) segments grouped by word
                b->seq_seg_lo.resize((size_t)U * GH_SEQ_MAXK); b->seq_seg_hi.resize((size_t)U * GH_SEQ_MAXK);
                GH_HIP(hipMemcpyAsync(b->seq_seg_lo.data(), d_seglo, (size_t)U * GH_SEQ_MAXK * 4, hipMemcpyDeviceToHost, st));
                GH_HIP(hipMemcpyAsync(b->seq_seg_hi.data(), d_seghi, (size_t)U * GH_SEQ_MAXK * 4, hipMemcpyDeviceToHost, st));
                b->seq_utt_K.resize(U); b->seq_utt_word.assign((size_t)U * GH_SEQ_MAXK, -1);
                for (int64_t u = 0; u < U; ++u) {
                    const gh_seqgraph& sg = lat->h_seqgraphs[utt_lattice ? utt_lattice[u] : 0];
                    b->seq_utt_K[u] = sg.K;
                    for (int k = 0; k < sg.K; ++k) b->seq_utt_word[(size_t)u * GH_SEQ_MAXK + k] = sg.word[k];
                }
                b->seq_word_chains.resize(lat->h_seqwords.size());
                for (size_t w = 0; w < lat->h_seqwords.size(); ++w) {
                    gh_fbchain& fc = b->seq_word_chains[w];
                    memset(&fc, 0, sizeof fc);
                    fc.n = lat->seq_N;
                    for (int j = 0; j < lat->seq_N; ++j) fc.state[j] = lat->h_seqwords[w].state[j];
                }
                b->seq_seg_valid = true;
            }
            std::vector<double> parts;
            if (out_self_xi) {
                parts.resize((size_t)GH_FBSEQ_XI_PARTS * S);
                GH_HIP(hipMemcpyAsync(parts.data(), d_xiparts, parts.size() * 8, hipMemcpyDeviceToHost, st));
            }
            GH_HIP(hipStreamSynchronize(st));
            if (out_self_xi)
                for (int s = 0; s < S; ++s) {
                    double acc2 = 0.0;
                    for (int p = 0; p < GH_FBSEQ_XI_PARTS; ++p) acc2 += parts[(size_t)p * S + s];
                    out_self_xi[s] = acc2;
                }
            return GH_OK;
        }
    }
    int max_level_rows = 1;
    for (auto& d : lat->h_desc) max_level_rows = std::max(max_level_rows, d.pad);
    const int block = std::min(512, std::max(64, (max_level_rows + 63) & ~63));
    // LDS: two cost columns, three emission / occupancy vectors, and the whole graph (both CSRs, row tables)
    int fb_max_arcs = 0;
    for (auto& lh : lat->lat) fb_max_arcs = std::max(fb_max_arcs, lh.A);
    a.arc_cap = (fb_max_arcs + 2) & ~1;
    a.lev_cap = lat->max_nlev + 1;
    const size_t lds = ((size_t)2 * a.r_pad + 5 * (size_t)S + 2 * (size_t)a.arc_cap) * sizeof(double) +
                       ((size_t)2 * a.arc_cap + 2 * (size_t)(a.r_pad + 2) + 3 * (size_t)a.r_pad + a.lev_cap + 4) * sizeof(int32_t);
    if (lds > 150 * 1024) {
        gh_set_error("gh_forward_backward: %d rows + %d states need %zu B of LDS", lat->max_R, S, lds);
        return GH_ERR_UNSUPPORTED;
    }
    for (size_t c = 0; c + 1 < chunk_begin.size(); ++c) {
        a.u_begin = chunk_begin[c];
        rc = gh_launch_fb(ctx, a, chunk_begin[c + 1] - chunk_begin[c], block, lds, b->dtype == GH_F64);
        if (rc) return rc;
    }
    if (out_logp) GH_HIP(hipMemcpyAsync(out_logp, d_logp, U * 8, hipMemcpyDeviceToHost, st));
    if (out_alpha) GH_HIP(hipMemcpyAsync(out_alpha, d_alpha, n_mat * 8, hipMemcpyDeviceToHost, st));
    if (out_beta) GH_HIP(hipMemcpyAsync(out_beta, d_beta, n_mat * 8, hipMemcpyDeviceToHost, st));
    if (out_gamma) GH_HIP(hipMemcpyAsync(out_gamma, d_gamma, n_mat * 8, hipMemcpyDeviceToHost, st));
    if (out_occ) GH_HIP(hipMemcpyAsync(out_occ, b->occ, (size_t)b->N * S * 8, hipMemcpyDeviceToHost, st));
    if (out_self_xi) GH_HIP(hipMemcpyAsync(out_self_xi, d_selfxi, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}


extern "C" int gh_align_segments(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                                 double* out_end_cost, int32_t* out_best_end, int32_t* out_frame_state) {
    GH_REQUIRE(ctx && lat && b && out_frame_state, "gh_align_segments: NULL argument");
    return viterbi_impl(ctx, lat, b, utt_lattice, out_end_cost, out_best_end, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, -1, out_frame_state);
}

// The alignment of gh_align_segments with the result as RUNS instead of one label per frame: out_runs [U, run_cap, 3] =
// (state, first frame inside the utterance, frames) of every run of an utterance's best path in time order, out_run_cnt [U]
// their number (a count above run_cap: the table was too small -- run_cap = the most emitting rows a graph has is
// always enough).  ~N / 20 runs instead of N labels: what continuous_train's regrouping (continuous_speech.py:90-113)
// needs, at 1 MB instead of 5.6 MB per 1.4 M frames, and the frames of a run are contiguous in the batch
// (gh_batch_gather_runs).
extern "C" int gh_align_runs(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                             double* out_end_cost, int32_t* out_best_end, int run_cap, int32_t* out_runs, int32_t* out_run_cnt) {
    GH_REQUIRE(ctx && lat && b && out_runs && out_run_cnt && run_cap > 0, "gh_align_runs: NULL argument / run_cap=%d", run_cap);
    return viterbi_impl(ctx, lat, b, utt_lattice, out_end_cost, out_best_end, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, -1, nullptr, nullptr, 0, run_cap, out_runs, out_run_cnt);
}

extern "C" int gh_viterbi_labels_packed(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                                        const int32_t* row_label, int max_labels, double* out_end_cost, int32_t* out_best_end,
                                        int32_t* out_labels, int64_t out_capacity, int32_t* out_n_labels) {
    GH_REQUIRE(ctx && lat && b && row_label && out_labels && out_n_labels, "gh_viterbi_labels_packed: NULL argument");
    GH_REQUIRE(max_labels > 0 && out_capacity >= 0, "gh_viterbi_labels_packed: max_labels=%d", max_labels);
    std::vector<int64_t> off((size_t)b->U + 1);
    for (int64_t u = 0; u <= b->U; ++u) off[u] = u * (int64_t)max_labels;
    return viterbi_impl(ctx, lat, b, utt_lattice, out_end_cost, out_best_end, nullptr, nullptr, nullptr, nullptr, nullptr,
                        row_label, out_labels, off.data(), out_n_labels, out_capacity);
}
