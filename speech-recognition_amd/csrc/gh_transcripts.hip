// C ABI: forced-alignment graphs straight from transcripts (gh_lattices_create_transcripts).
//
// continuous_train builds, for every utterance, build_state_sequences(models, [[l] for l in labels])
// (continuous_speech.py:80-82): one word per layer.  Handing those graphs over as arc lists costs a Python loop per
// distinct transcript and, on this side, a sort / level / form-detection pass per graph (80 ms for 12 500 transcripts)
// -- every training iteration, because the transition costs change.  Here the graphs arrive as what they are: W word
// models and label strings.  The sequence form (gh_seqgraph, gh_internal.h) is written down directly, O(words); the
// row-per-lane representation -- needed only by the fallbacks: a beam, single-frame utterances, alpha / beta
// matrices, GMMHMM_VITERBI / GMMHMM_FB overrides -- is expanded lazily into a second, ordinary handle.
#include "gh_internal.h"
#include "gh_host.h"
#include <mutex>

struct gh_transcripts_src {
    int W, n;
    std::vector<double> wt;        // [W, n, n], entry [i, j] = cost j -> i, +inf = no arc
    std::vector<int32_t> base;     // [W] first state of every word
    std::vector<int64_t> label_off;
    std::vector<int32_t> labels;
};

void gh_transcripts_src_free(gh_transcripts_src* s) { delete s; }

// the graphs as packed_lattice(word_transitions, n, [[l] for l in labels]) lays them out (continuous_speech.py:13-53)
static int create_expanded(gh_ctx* ctx, const gh_transcripts_src& s, gh_lattices** out) {
    const int n = s.n;
    const int64_t L = (int64_t)s.label_off.size() - 1;
    std::vector<int64_t> row_off(L + 1, 0), arc_off(L + 1, 0), start_off(L + 1, 0), end_off(L + 1, 0);
    std::vector<int32_t> row_state, arc_to, arc_from, start_rows, end_rows;
    std::vector<double> arc_cost;
    for (int64_t l = 0; l < L; ++l) {
        const int K = (int)(s.label_off[l + 1] - s.label_off[l]);
        const int32_t* lab = s.labels.data() + s.label_off[l];
        row_state.push_back(-1);
        for (int k = 0; k < K; ++k) {
            const int w = lab[k], r0 = k * (n + 1) + 1;
            for (int i = 0; i < n; ++i) row_state.push_back(s.base[w] + i);
            row_state.push_back(-1);
            const double* wt = s.wt.data() + (size_t)w * n * n;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (!std::isinf(wt[i * n + j])) { arc_to.push_back(r0 + i); arc_from.push_back(r0 + j); arc_cost.push_back(wt[i * n + j]); }
            arc_to.push_back(r0); arc_from.push_back(r0 - 1); arc_cost.push_back(0.0);
            arc_to.push_back(r0 + n); arc_from.push_back(r0 + n - 1); arc_cost.push_back(0.0);
        }
        start_rows.push_back(0);
        if (K > 0) end_rows.push_back((K - 1) * (n + 1) + n);
        row_off[l + 1] = (int64_t)row_state.size();
        arc_off[l + 1] = (int64_t)arc_to.size();
        start_off[l + 1] = (int64_t)start_rows.size();
        end_off[l + 1] = (int64_t)end_rows.size();
    }
    return gh_lattices_create(ctx, (int)L, row_off.data(), row_state.data(), arc_off.data(), arc_to.data(), arc_from.data(),
                              arc_cost.data(), start_off.data(), start_rows.data(), end_off.data(), end_rows.data(), out);
}

int gh_lattices_full(const gh_lattices* l, const gh_lattices** out) {
    *out = l;
    if (!l->deferred_src) return GH_OK;
    gh_lattices* self = const_cast<gh_lattices*>(l);
    // the expansion happens inside calls the ABI presents as read-only (gh_viterbi / gh_forward_backward on a const
    // handle): two host threads sharing the handle must not both build the twin
    static std::mutex expand_lock;
    std::lock_guard<std::mutex> guard(expand_lock);
    if (!self->full) {
        const int rc = create_expanded(l->ctx, *l->deferred_src, &self->full);
        if (rc) return rc;
    }
    self->full->beam = l->beam;
    *out = self->full;
    return GH_OK;
}

extern "C" int gh_lattices_create_transcripts(gh_ctx* ctx, int W, int n, const double* word_trans, const int32_t* state_base,
                                              int64_t L, const int64_t* label_off, const int32_t* labels, gh_lattices** out) {
    GH_REQUIRE(ctx && out && word_trans && label_off && (labels || label_off[L] == 0), "gh_lattices_create_transcripts: NULL argument");
    GH_REQUIRE(W > 0 && n > 0 && L > 0 && label_off[0] == 0, "gh_lattices_create_transcripts: W=%d n=%d L=%lld", W, n, (long long)L);
    *out = nullptr;
    gh_transcripts_src* src = new gh_transcripts_src();
    src->W = W; src->n = n;
    src->wt.assign(word_trans, word_trans + (size_t)W * n * n);
    src->base.resize(W);
    for (int w = 0; w < W; ++w) src->base[w] = state_base ? state_base[w] : w * n;
    src->label_off.assign(label_off, label_off + L + 1);
    src->labels.assign(labels, labels + label_off[L]);
    bool direct = gh_seq_n_ok(n) && L <= 0x7fffffff;
    bool all_single = true;
    for (int64_t l = 0; l < L; ++l) {
        const int64_t K = label_off[l + 1] - label_off[l];
        if (K <= 0) { delete src; gh_set_error("gh_lattices_create_transcripts: transcript %lld is empty (no end row)", (long long)l); return GH_ERR_INVALID; }
        if (K > GH_SEQ_MAXK) direct = false;
        if (K != 1) all_single = false;
        for (int64_t k = label_off[l]; k < label_off[l + 1]; ++k)
            if (labels[k] < 0 || labels[k] >= W) { delete src; gh_set_error("gh_lattices_create_transcripts: label %d out of range", labels[k]); return GH_ERR_INVALID; }
    }
    // word templates; anything but arcs from s, s-1, s-2 (or a NaN cost) needs the row-per-lane kernels
    std::vector<gh_seqword> words(W);
    bool skip = false;
    for (int w = 0; direct && w < W; ++w) {
        gh_seqword& wd = words[w];
        memset(&wd, 0, sizeof wd);
        wd.cin = wd.cout = 0.0;
        wd.arcs[0] = 8;
        for (int sx = 0; sx < GH_LAYERS_MAXN; ++sx) { wd.c0[sx] = wd.c1[sx] = wd.c2[sx] = INFINITY; wd.state[sx] = -1; }
        for (int i = 0; i < n; ++i) {
            wd.state[i] = src->base[w] + i;
            for (int j = 0; j < n; ++j) {
                const double c = word_trans[((size_t)w * n + i) * n + j];
                if (std::isinf(c)) continue;
                const int d = i - j;
                if (c != c || d < 0 || d > 2) { direct = false; break; }
                (d == 0 ? wd.c0[i] : d == 1 ? wd.c1[i] : wd.c2[i]) = c;
                wd.arcs[i] |= (uint8_t)(1 << d);
                if (d == 2) skip = true;
            }
        }
    }
    // isolated words only: the one-word chain forms of the ordinary handle (fused statistics kernel) are the better fit
    if (!direct || all_single) {
        const int rc = create_expanded(ctx, *src, out);
        delete src;
        return rc;
    }
    GH_HIP(hipSetDevice(ctx->device));
    gh_lattices* lt = new gh_lattices();
    lt->ctx = ctx; lt->d_arena = nullptr; lt->beam = 0; lt->L = (int)L; lt->max_R = 0; lt->max_nlev = 0;
    lt->has_nan_arc = false; lt->has_self_arc = false;
    lt->d_row_state = nullptr; lt->d_row_start = nullptr; lt->d_pred_ptr = nullptr; lt->d_pred_row = nullptr;
    lt->d_pred_cost = nullptr; lt->d_order = nullptr; lt->d_succ_ptr = nullptr; lt->d_succ_row = nullptr;
    lt->d_succ_cost = nullptr; lt->d_level_ptr = nullptr; lt->d_end_rows = nullptr; lt->d_level_narrow = nullptr;
    lt->chain_ok = false; lt->chain_skip = false; lt->chain_groups = 0;
    lt->d_ch_cost0 = lt->d_ch_cost1 = lt->d_ch_cost2 = nullptr; lt->d_ch_info = nullptr;
    lt->d_ch_end_slot = nullptr; lt->d_ch_group_row0 = nullptr; lt->d_desc = nullptr;
    lt->d_fbchain = nullptr; lt->fbchain_ok = false;
    lt->layers_ok = false; lt->d_layers = nullptr; lt->d_lf_end_slot = nullptr;
    lt->seq_ok = true; lt->seq_N = n; lt->seq_skip = skip ? 1 : 0;
    lt->deferred_src = src; lt->full = nullptr;
    std::vector<int32_t> row_state, end_rows(L), end_slot;
    int n_arcs_word_max = 0;
    std::vector<int> word_arcs(W, 0);
    for (int w = 0; w < W; ++w) {
        for (int i = 0; i < n; ++i) word_arcs[w] += __builtin_popcount(words[w].arcs[i] & 7);
        n_arcs_word_max = std::max(n_arcs_word_max, word_arcs[w]);
    }
    int64_t r_base = 0, a_base = 0;
    lt->lat.resize(L); lt->h_desc.resize(L); lt->h_seqgraphs.resize(L);
    int64_t Rtot = 0;
    for (int64_t l = 0; l < L; ++l) Rtot += (label_off[l + 1] - label_off[l]) * (n + 1) + 1;
    row_state.reserve(Rtot);
    end_slot.assign(Rtot, -1);
    for (int64_t l = 0; l < L; ++l) {
        const int K = (int)(label_off[l + 1] - label_off[l]);
        const int32_t* lab = labels + label_off[l];
        const int R = K * (n + 1) + 1;
        gh_seqgraph& sg = lt->h_seqgraphs[l];
        memset(&sg, 0, sizeof sg);
        sg.K = K; sg.n_end = 1; sg.row_base = r_base; sg.end_base = l;
        int A = 0, max_state = -1;
        row_state.push_back(-1);
        for (int k = 0; k < K; ++k) {
            sg.word[k] = lab[k];
            for (int i = 0; i < n; ++i) row_state.push_back(src->base[lab[k]] + i);
            row_state.push_back(-1);
            A += word_arcs[lab[k]] + 2;
            max_state = std::max(max_state, src->base[lab[k]] + n - 1);
        }
        end_rows[l] = (K - 1) * (n + 1) + n;
        end_slot[r_base + end_rows[l]] = 0;
        gh_lattice_host& lh = lt->lat[l];
        lh.R = R; lh.A = A; lh.nlev = K >= 2 ? 3 : 2; lh.n_start = 1; lh.n_end = 1;
        lh.row_base = r_base; lh.arc_base = a_base; lh.end_base = l; lh.max_state = max_state;
        gh_lattices::desc& d = lt->h_desc[l];
        memset(&d, 0, sizeof d);
        d.R = R; d.nlev = lh.nlev; d.n_end = 1; d.pad = K * (n - 1) + 1; d.lean_lanes = d.pad;
        d.row_base = r_base; d.arc_base = a_base; d.end_base = l;
        lt->max_R = std::max(lt->max_R, R);
        lt->max_nlev = std::max(lt->max_nlev, lh.nlev);
        r_base += R; a_base += A;
    }
    lt->h_seqwords = words;
    UploadArena ar;
    ar.add(&lt->d_row_state, row_state); ar.add(&lt->d_end_rows, end_rows); ar.add(&lt->d_desc, lt->h_desc);
    ar.add(&lt->d_seqgraphs, lt->h_seqgraphs); ar.add(&lt->d_seqwords, lt->h_seqwords); ar.add(&lt->d_seq_end_slot, end_slot);
    const int rc = ar.commit(&lt->d_arena);
    if (rc) { gh_lattices_destroy(lt); return rc; }
    *out = lt;
    return GH_OK;
}
