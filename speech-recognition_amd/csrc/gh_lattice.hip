// C ABI: DP graphs (gh_lattices): CSR by destination in ascending origin order, same-column /
// dead arc flags, levels, narrow/wide row ordering, transposed CSR, chain form detection.
#include "gh_internal.h"
#include "gh_host.h"

// ----------------------------------------------------------------- lattices
extern "C" int gh_lattices_create(gh_ctx* ctx, int L, const int64_t* row_off, const int32_t* row_state,
                                  const int64_t* arc_off, const int32_t* arc_to, const int32_t* arc_from,
                                  const double* arc_cost, const int64_t* start_off, const int32_t* start_rows,
                                  const int64_t* end_off, const int32_t* end_rows, gh_lattices** out) {
    GH_REQUIRE(ctx && out && row_off && row_state && arc_off && start_off && end_off && end_rows,
               "gh_lattices_create: NULL argument");
    GH_REQUIRE(L > 0, "gh_lattices_create: L=%d", L);
    *out = nullptr;
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t Rtot = row_off[L], Atot = arc_off[L], Etot = end_off[L];
    GH_REQUIRE(row_off[0] == 0 && arc_off[0] == 0 && start_off[0] == 0 && end_off[0] == 0,
               "gh_lattices_create: offsets must start at 0");
    std::vector<int32_t> h_state(row_state, row_state + Rtot), h_ptr, h_order(Rtot), h_lev, h_narrow, h_end;
    std::vector<uint8_t> h_start(Rtot, 0);
    std::vector<uint32_t> h_prow(Atot), h_srow(Atot);
    std::vector<double> h_pcost(Atot), h_scost(Atot);
    std::vector<int32_t> h_sptr;
    gh_lattices* lt = new gh_lattices();
    lt->ctx = ctx; lt->d_arena = nullptr; lt->beam = 0; lt->L = L; lt->max_R = 0; lt->max_nlev = 0; lt->has_nan_arc = false; lt->has_self_arc = false;
    for (int64_t k = 0; k < Atot; ++k) lt->has_nan_arc = lt->has_nan_arc || std::isnan(arc_cost[k]);
    lt->d_row_state = nullptr; lt->d_row_start = nullptr; lt->d_pred_ptr = nullptr; lt->d_pred_row = nullptr;
    lt->d_pred_cost = nullptr; lt->d_order = nullptr; lt->d_succ_ptr = nullptr; lt->d_succ_row = nullptr;
    lt->d_succ_cost = nullptr; lt->d_level_ptr = nullptr; lt->d_end_rows = nullptr; lt->d_level_narrow = nullptr;
    lt->chain_ok = false; lt->chain_skip = false; lt->chain_groups = 0;
    lt->d_ch_cost0 = lt->d_ch_cost1 = lt->d_ch_cost2 = nullptr; lt->d_ch_info = nullptr;
    lt->d_ch_end_slot = nullptr; lt->d_ch_group_row0 = nullptr;
    lt->d_desc = nullptr;
    lt->d_fbchain = nullptr; lt->fbchain_ok = true;
    lt->layers_ok = false; lt->d_layers = nullptr; lt->d_lf_end_slot = nullptr;
    lt->seq_ok = true; lt->seq_N = 0; lt->seq_skip = 0; lt->d_seqgraphs = nullptr; lt->d_seqwords = nullptr; lt->d_seq_end_slot = nullptr;
    std::vector<int32_t> seq_slot(Rtot, -1);
    std::vector<std::string> seq_keys;   // content keys of the word templates collected so far
    h_end.assign(end_rows, end_rows + Etot);
    for (int l = 0; l < L; ++l) {
        const int64_t r0 = row_off[l], a0 = arc_off[l];
        const int R = (int)(row_off[l + 1] - r0), A = (int)(arc_off[l + 1] - a0);
        const int ns = (int)(start_off[l + 1] - start_off[l]), ne = (int)(end_off[l + 1] - end_off[l]);
#define GH_LFAIL(...) do { gh_set_error(__VA_ARGS__); delete lt; return GH_ERR_INVALID; } while (0)
        if (R <= 0 || R > 0x7FFE) GH_LFAIL("gh_lattices_create: graph %d has %d rows (1..32766 supported)", l, R);
        if (ne <= 0) GH_LFAIL("gh_lattices_create: graph %d has no end row", l);
        for (int k = 0; k < ns; ++k) {
            const int r = start_rows[start_off[l] + k];
            if (r < 0 || r >= R) GH_LFAIL("gh_lattices_create: start row %d out of range", r);
            h_start[r0 + r] |= 1;
        }
        for (int k = 0; k < ne; ++k) {
            const int r = end_rows[end_off[l] + k];
            if (r < 0 || r >= R) GH_LFAIL("gh_lattices_create: end row %d out of range", r);
            h_start[r0 + r] |= 2;  // bit1 = end row (forward-backward)
        }
        // CSR by destination, ascending origin (tie-break contract, decode.py:105-118)
        std::vector<int> idx(A);
        std::iota(idx.begin(), idx.end(), 0);
        for (int k = 0; k < A; ++k) {
            const int to = arc_to[a0 + k], fr = arc_from[a0 + k];
            if (to < 0 || to >= R || fr < 0 || fr >= R) GH_LFAIL("gh_lattices_create: arc %d out of range", k);
            if (std::isinf(arc_cost[a0 + k]))  // NaN is kept: the reference treats it as an arc (isinf(nan) is False)
                GH_LFAIL("gh_lattices_create: arc %d has an infinite cost (omit absent arcs, decode.py:106)", k);
        }
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {
            const int tx = arc_to[a0 + x], ty = arc_to[a0 + y];
            if (tx != ty) return tx < ty;
            return arc_from[a0 + x] < arc_from[a0 + y];
        });
        const size_t ptr_base = h_ptr.size();
        h_ptr.resize(ptr_base + R + 1, 0);
        int32_t* ptr = h_ptr.data() + ptr_base;
        std::vector<int> level(R, 0);
        for (int k = 0; k < A; ++k) ptr[arc_to[a0 + idx[k]] + 1]++;
        for (int r = 0; r < R; ++r) ptr[r + 1] += ptr[r];
        for (int k = 0; k < A; ++k) {
            const int to = arc_to[a0 + idx[k]], fr = arc_from[a0 + idx[k]];
            uint32_t w = (uint32_t)fr;
            const bool same = h_state[r0 + to] < 0 || h_state[r0 + fr] < 0;  // decode.py:109
            if (same) {
                if (fr == to) lt->has_self_arc = true;
                w |= GH_ARC_SAME;
                if (fr >= to) w |= GH_ARC_DEAD;  // not yet computed in this column: still +inf
            }
            h_prow[a0 + k] = w;
            h_pcost[a0 + k] = arc_cost[a0 + idx[k]];
        }
        // transposed CSR (by origin) for the backward pass; h_sptr shares ptr_base with h_ptr
        {
            h_sptr.resize(ptr_base + R + 1, 0);
            int32_t* sp = h_sptr.data() + ptr_base;
            for (int r = 0; r < R; ++r)
                for (int p = ptr[r]; p < ptr[r + 1]; ++p) sp[(h_prow[a0 + p] & GH_ARC_ROW) + 1]++;
            for (int r = 0; r < R; ++r) sp[r + 1] += sp[r];
            std::vector<int> fill(sp, sp + R);
            for (int r = 0; r < R; ++r)
                for (int p = ptr[r]; p < ptr[r + 1]; ++p) {
                    const uint32_t w = h_prow[a0 + p];
                    const int o = (int)(w & GH_ARC_ROW), k = fill[o]++;
                    h_srow[a0 + k] = (uint32_t)r | (w & ~GH_ARC_ROW);
                    h_scost[a0 + k] = h_pcost[a0 + p];
                }
        }
        // levels: rows ascending, so every live same-column origin (< row) is already levelled
        int nlev = 1;
        for (int r = 0; r < R; ++r) {
            int lv = 0;
            for (int p = ptr[r]; p < ptr[r + 1]; ++p) {
                const uint32_t w = h_prow[a0 + p];
                if ((w & GH_ARC_SAME) && !(w & GH_ARC_DEAD)) lv = std::max(lv, level[w & GH_ARC_ROW] + 1);
            }
            level[r] = lv;
            nlev = std::max(nlev, lv + 1);
        }
        const size_t lev_base = h_lev.size();
        h_lev.resize(lev_base + nlev + 1, 0);
        int32_t* lp = h_lev.data() + lev_base;
        for (int r = 0; r < R; ++r) lp[level[r] + 1]++;
        int max_level_rows = 0;
        for (int k = 0; k < nlev; ++k) { max_level_rows = std::max(max_level_rows, lp[k + 1]); lp[k + 1] += lp[k]; }
        // inside a level: rows with <= GH_LEAN_NARROW_ARCS arcs first, then the wide rows (the lean kernel gives those
        // 16 lanes each); ascending row index inside both groups
        h_narrow.resize(lev_base + nlev + 1, 0);
        int lean_lanes = 0;
        {
            std::vector<int> fill(lp, lp + nlev), nwide(nlev, 0);
            for (int pass = 0; pass < 2; ++pass)
                for (int r = 0; r < R; ++r) {
                    const bool wide = ptr[r + 1] - ptr[r] > GH_LEAN_NARROW_ARCS;
                    if (wide != (pass == 1)) continue;
                    h_order[r0 + fill[level[r]]++] = r;
                    if (wide) nwide[level[r]]++; else h_narrow[lev_base + level[r]]++;
                }
            for (int k = 0; k < nlev; ++k) lean_lanes = std::max(lean_lanes, h_narrow[lev_base + k] + 16 * nwide[k]);
        }
        int max_state = -1;
        for (int r = 0; r < R; ++r) max_state = std::max(max_state, h_state[r0 + r]);
        gh_lattice_host lh;
        lh.R = R; lh.A = A; lh.nlev = nlev; lh.n_start = ns; lh.n_end = ne;
        lh.row_base = r0; lh.arc_base = a0; lh.end_base = end_off[l]; lh.max_state = max_state;
        lt->lat.push_back(lh);
        gh_lattices::desc d;
        d.R = R; d.nlev = nlev; d.n_end = ne; d.pad = max_level_rows; d.lean_lanes = lean_lanes; d.pad2 = 0;
        d.row_base = r0; d.ptr_base = (int64_t)ptr_base; d.arc_base = a0; d.lev_base = (int64_t)lev_base;
        d.end_base = end_off[l];
        lt->h_desc.push_back(d);
        lt->max_R = std::max(lt->max_R, R);
        lt->max_nlev = std::max(lt->max_nlev, nlev);

        // ---- sequence form (gh_seqgraph): K layers of ONE word each, non-emitting rows between them ----
        if (lt->seq_ok) {
            bool ok = R >= 4 && h_state[r0] < 0 && ns == 1 && (h_start[r0] & 1) && !lt->has_nan_arc;
            int N = 0;
            if (ok) {
                int r = 1;
                while (r < R && h_state[r0 + r] >= 0) ++r;
                N = r - 1;
                ok = gh_seq_n_ok(N) && r < R && (R - 1) % (N + 1) == 0 && (lt->seq_N == 0 || lt->seq_N == N);
            }
            const int K = ok ? (R - 1) / (N + 1) : 0;
            ok = ok && K >= 1 && K <= GH_SEQ_MAXK;
            for (int r = 0; ok && r < R; ++r) if ((r % (N + 1) == 0) != (h_state[r0 + r] < 0)) ok = false;
            std::vector<gh_seqword> wl(ok ? K : 0);
            for (auto& wd : wl) {
                memset(&wd, 0, sizeof wd);
                wd.cin = wd.cout = INFINITY;
                for (int sx = 0; sx < GH_LAYERS_MAXN; ++sx) { wd.c0[sx] = wd.c1[sx] = wd.c2[sx] = INFINITY; wd.state[sx] = -1; }
            }
            bool skip = false;
            for (int r = 0; ok && r < R; ++r)
                for (int p = ptr[r]; ok && p < ptr[r + 1]; ++p) {
                    const int o = (int)(h_prow[a0 + p] & GH_ARC_ROW);
                    const double c = h_pcost[a0 + p];
                    const bool r_nes = r % (N + 1) == 0, o_nes = o % (N + 1) == 0;
                    if (!r_nes && !o_nes) {
                        const int k = (r - 1) / (N + 1), sx = (r - 1) % (N + 1), d = r - o;
                        if ((o - 1) / (N + 1) != k || d < 0 || d > 2 || d > sx) { ok = false; break; }
                        double& t = d == 0 ? wl[k].c0[sx] : d == 1 ? wl[k].c1[sx] : wl[k].c2[sx];
                        if (!std::isinf(t)) { ok = false; break; }
                        t = c;
                        wl[k].arcs[sx] |= (uint8_t)(1 << d);
                        if (d == 2) skip = true;
                    } else if (o_nes && !r_nes) {              // non-emitting row k -> state 0 of layer k
                        const int k = (r - 1) / (N + 1), sx = (r - 1) % (N + 1);
                        if (o != k * (N + 1) || sx != 0 || !std::isinf(wl[k].cin)) { ok = false; break; }
                        wl[k].cin = c;
                        wl[k].arcs[0] |= 8;
                    } else if (r_nes && !o_nes) {              // last state of layer k -> non-emitting row k + 1
                        const int k = (o - 1) / (N + 1), sx = (o - 1) % (N + 1);
                        if (r != (k + 1) * (N + 1) || sx != N - 1 || !std::isinf(wl[k].cout)) { ok = false; break; }
                        wl[k].cout = c;
                    } else { ok = false; break; }
                }
            for (int k = 0; ok && k < K; ++k) {
                if (std::isinf(wl[k].cin)) ok = false;        // every layer is entered through its non-emitting row
                for (int sx = 0; sx < N; ++sx) wl[k].state[sx] = h_state[r0 + k * (N + 1) + 1 + sx];
            }
            for (int k = 0; ok && k < ne; ++k) {
                const int r = end_rows[end_off[l] + k];
                if (r % (N + 1) == 0 || seq_slot[r0 + r] >= 0) ok = false;
                else seq_slot[r0 + r] = k;
            }
            if (ok) {
                gh_seqgraph sg;
                memset(&sg, 0, sizeof sg);
                sg.K = K; sg.n_end = ne; sg.row_base = r0; sg.end_base = end_off[l];
                for (int k = 0; k < K; ++k) {
                    const std::string key(reinterpret_cast<const char*>(&wl[k]), sizeof(gh_seqword));
                    int id = -1;
                    for (size_t i = 0; i < seq_keys.size(); ++i) if (seq_keys[i] == key) { id = (int)i; break; }
                    if (id < 0) { id = (int)seq_keys.size(); seq_keys.push_back(key); lt->h_seqwords.push_back(wl[k]); }
                    sg.word[k] = id;
                }
                lt->h_seqgraphs.push_back(sg);
                lt->seq_N = N;
                if (skip) lt->seq_skip = 1;
            } else {
                lt->seq_ok = false;
            }
        }
        // ---- one-word chain form for the forward-backward kernel (see gh_fbchain) ----
        if (lt->fbchain_ok) {
            gh_fbchain fc;
            memset(&fc, 0, sizeof fc);
            std::vector<int> em;  // emitting rows, ascending
            for (int r = 0; r < R; ++r) if (h_state[r0 + r] >= 0) em.push_back(r);
            bool ok = !em.empty() && (int)em.size() <= GH_FBCHAIN_MAX && ne == 1 && end_rows[end_off[l]] == em.back();
            std::vector<int> pos(R, -1);
            for (int j = 0; ok && j < (int)em.size(); ++j) {
                pos[em[j]] = j;
                fc.state[j] = h_state[r0 + em[j]];
                fc.self_c[j] = fc.next_c[j] = fc.skip_c[j] = INFINITY;
                for (int i = 0; i < j; ++i) ok = ok && fc.state[i] != fc.state[j];
            }
            fc.n = (int)em.size();
            fc.c0 = INFINITY;
            int start_row = -1;
            for (int k = 0; ok && k < ns; ++k) {
                if (start_row >= 0) ok = false;
                start_row = start_rows[start_off[l] + k];
            }
            ok = ok && start_row >= 0;
            if (ok && start_row == em[0]) fc.c0 = 0.0;
            else if (ok && h_state[r0 + start_row] >= 0) ok = false;  // starts at an emitting row that is not the first
            for (int k = 0; ok && k < A; ++k) {
                const int to = arc_to[a0 + k], fr = arc_from[a0 + k];
                const double c = arc_cost[a0 + k];
                if (c != c) { ok = false; break; }
                if (pos[to] >= 0 && pos[fr] >= 0) {           // emitting -> emitting: previous column
                    const int dlt = pos[to] - pos[fr];
                    double* slot = dlt == 0 ? &fc.self_c[pos[to]] : dlt == 1 ? &fc.next_c[pos[to]] : dlt == 2 ? &fc.skip_c[pos[to]] : nullptr;
                    if (!slot || !std::isinf(*slot)) ok = false; else *slot = c;
                } else if (pos[to] >= 0) {                    // non-emitting -> emitting: only start row -> first row
                    if (fr == start_row && to == em[0] && std::isinf(fc.c0) && fr < to) fc.c0 = c; else ok = false;
                } else {                                      // anything -> non-emitting row: must not lead anywhere
                    if (pos[fr] < 0) ok = false;              // (non-emitting chains are not modelled)
                }
            }
            // a non-emitting row that is fed may not feed an emitting row (checked above: only the start row does),
            // and the start row must not be fed
            for (int k = 0; ok && k < A; ++k) if (arc_to[a0 + k] == start_row && start_row != em[0]) ok = false;
            ok = ok && !std::isinf(fc.c0);
            for (int j = 0; ok && j < fc.n; ++j) if (!std::isinf(fc.skip_c[j])) fc.pad = 1;   // pad = "has skip arcs" (ok: fc.n <= GH_FBCHAIN_MAX -- found by ASan: a graph with more emitting rows read past the arrays)
            if (ok) lt->h_fbchain.push_back(fc); else { lt->fbchain_ok = false; lt->h_fbchain.clear(); }
        }
#undef GH_LFAIL
    }
    // ---- chain form: one graph, one level, arcs only from r, r-1, r-2, distinct end rows ----
    std::vector<double> ch0, ch1, ch2;
    std::vector<uint8_t> chinfo;
    std::vector<int32_t> chslot, chgroups;
    if (L == 1 && lt->max_nlev == 1 && !lt->has_nan_arc) {
        const int R = lt->lat[0].R;
        const int32_t* ptr = h_ptr.data();
        ch0.assign(R, INFINITY); ch1.assign(R, INFINITY); ch2.assign(R, INFINITY);
        chinfo.assign(R, 3); chslot.assign(R, -1);
        bool ok = true, skip = false;
        for (int r = 0; r < R && ok; ++r) {
            int first = 3;
            for (int p = ptr[r]; p < ptr[r + 1]; ++p) {
                const int o = (int)(h_prow[p] & GH_ARC_ROW), dlt = r - o;
                if (dlt < 0 || dlt > 2 || (h_prow[p] & (GH_ARC_SAME | GH_ARC_DEAD))) { ok = false; break; }
                double& slot = dlt == 0 ? ch0[r] : (dlt == 1 ? ch1[r] : ch2[r]);
                if (!std::isinf(slot)) { ok = false; break; }  // duplicate arc
                slot = h_pcost[p];
                if (dlt == 2) skip = true;
                if (first == 3 || dlt > first) first = dlt;  // lowest origin == largest delta comes first
            }
            chinfo[r] = (uint8_t)(first | (h_start[r] & 1 ? 4 : 0));
        }
        for (int k = 0; ok && k < (int)h_end.size(); ++k) {
            if (chslot[h_end[k]] >= 0) ok = false;  // duplicated end row: generic / lean kernels handle it
            else chslot[h_end[k]] = k;
        }
        if (ok) {  // 64-lane groups made of whole chains (a chain starts where no arc arrives from r-1 / r-2)
            chgroups.push_back(0);
            int gs = 0, cs = 0, unit = 0;       // unit: the common length of all chains (0: they differ)
            bool uniform = true;
            for (int r = 1; r <= R && ok; ++r) {
                const bool chain_start = r == R || (std::isinf(ch1[r]) && std::isinf(ch2[r]) &&
                                                    (r + 1 >= R || std::isinf(ch2[r + 1])));
                if (!chain_start) continue;
                if (r - cs > 64) { ok = false; break; }     // one chain longer than a wave
                if (r - gs > 64) { chgroups.push_back(cs); gs = cs; }
                if (unit == 0) unit = r - cs; else if (r - cs != unit) uniform = false;
                cs = r;
            }
            chgroups.push_back(R);
            lt->chain_unit = (ok && uniform) ? unit : 0;
        }
        if (ok) { lt->chain_ok = true; lt->chain_skip = skip; lt->chain_groups = (int)chgroups.size() - 1; }
    }
    // ---- layer form (gh_layerform): K identical layers of W left-to-right words, non-emitting rows between them ----
    std::vector<int32_t> lf_slot;
    if (L == 1 && !lt->has_nan_arc && !lt->has_self_arc) {
        const int R = lt->lat[0].R, A = lt->lat[0].A;
        gh_layerform& f = lt->h_layers;
        memset(&f, 0, sizeof f);
        bool ok = R >= 4 && h_state[0] < 0;
        int P = 0;
        if (ok) {   // P = rows up to the next non-emitting row
            int r = 1;
            while (r < R && h_state[r] >= 0) ++r;
            P = r - 1;
            ok = P >= 2 && r < R && (R - 1) % (P + 1) == 0;
        }
        const int K = ok ? (R - 1) / (P + 1) : 0;
        ok = ok && K >= 1 && K <= GH_LAYERS_MAXK;
        for (int r = 0; ok && r < R; ++r) {
            const bool nes = r % (P + 1) == 0;
            if (nes != (h_state[r] < 0)) ok = false;
            if (!nes && h_state[r] != h_state[1 + (r - 1) % (P + 1)]) ok = false;   // same states in every layer
        }
        ok = ok && lt->lat[0].n_start == 1 && (h_start[0] & 1);
        // words: a new word starts where the non-emitting row has an arc into the row
        std::vector<int> word_of(P + 1, -1), pos_of(P + 1, -1);
        std::vector<double> tc0(P, INFINITY), tc1(P, INFINITY), tc2(P, INFINITY), tin(P, INFINITY), tout(P, INFINITY);
        std::vector<char> seen(5 * (size_t)std::max(P, 1) * std::max(K, 1), 0);   // every (layer, slot, position) arc at most once
        const int32_t* ptr = h_ptr.data();
        for (int r = 0; ok && r < R; ++r)
            for (int p = ptr[r]; ok && p < ptr[r + 1]; ++p) {
                const int o = (int)(h_prow[p] & GH_ARC_ROW);
                const double c = h_pcost[p];
                const bool r_nes = r % (P + 1) == 0, o_nes = o % (P + 1) == 0;
                int slot, k, pos;
                if (!r_nes && !o_nes) {              // emitting -> emitting: same layer, from s, s-1 or s-2
                    k = (r - 1) / (P + 1); pos = (r - 1) % (P + 1);
                    if ((o - 1) / (P + 1) != k || r - o < 0 || r - o > 2) { ok = false; break; }
                    slot = r - o;
                } else if (o_nes && !r_nes) {        // non-emitting row k -> a row of layer k
                    k = (r - 1) / (P + 1); pos = (r - 1) % (P + 1);
                    if (o != k * (P + 1)) { ok = false; break; }
                    slot = 3;
                } else if (r_nes && !o_nes) {        // a row of layer k -> non-emitting row k + 1
                    k = (o - 1) / (P + 1); pos = (o - 1) % (P + 1);
                    if (r != (k + 1) * (P + 1)) { ok = false; break; }
                    slot = 4;
                } else { ok = false; break; }
                double& t = slot == 0 ? tc0[pos] : slot == 1 ? tc1[pos] : slot == 2 ? tc2[pos] : slot == 3 ? tin[pos] : tout[pos];
                char& sn = seen[((size_t)k * 5 + slot) * P + pos];
                if (sn) { ok = false; break; }
                sn = 1;
                if (k == 0) t = c;
                else if (!(t == c)) { ok = false; break; }   // layers must be identical (bit-equal costs)
            }
        // every layer must have every arc of layer 0
        for (int k = 1; ok && k < K; ++k)
            for (int slot = 0; ok && slot < 5; ++slot)
                for (int pos = 0; ok && pos < P; ++pos)
                    if (seen[((size_t)k * 5 + slot) * P + pos] != seen[(size_t)slot * P + pos]) ok = false;
        int W = 0, N = 0;
        if (ok) {   // cut the layer into words at the rows fed by the non-emitting row
            for (int pos = 0; pos < P; ++pos) {
                if (!std::isinf(tin[pos])) { ++W; if (W == 2) N = pos; }
                if (W == 0) { ok = false; break; }
            }
            if (ok && W == 1) N = P;
            ok = ok && gh_seq_n_ok(N) && W <= GH_LAYERS_MAXW && W * N == P && (K <= 8 || N <= 8);   // (four register sets: N <= 8)
            ok = ok && (W <= GH_LAYERS_ROWW || (K <= 8 && N <= 8));                                   // (more than 16 words: the wide kernel)
        }
        bool skip = false;
        for (int pos = 0; ok && pos < P; ++pos) {
            const int w = pos / N, sx = pos % N;
            if ((sx == 0) != !std::isinf(tin[pos])) ok = false;                                   // entries exactly at state 0
            if (sx == 0 && (!std::isinf(tc1[pos]) || !std::isinf(tc2[pos]))) ok = false;          // no arc across words
            if (sx == 1 && !std::isinf(tc2[pos])) ok = false;
            if (!std::isinf(tout[pos]) && sx == 0) ok = false;                                    // an exit is never an entry
            if (!std::isinf(tout[pos]) && sx != N - 1) ok = false;                                // exits at the last state only
            if (!std::isinf(tc2[pos])) skip = true;
            f.state[w][sx] = h_state[1 + pos];
            f.c0[w][sx] = tc0[pos]; f.c1[w][sx] = tc1[pos]; f.c2[w][sx] = tc2[pos];
            f.arcs[w][sx] = (uint8_t)((!std::isinf(tc0[pos]) ? 1 : 0) | (!std::isinf(tc1[pos]) ? 2 : 0) |
                                      (!std::isinf(tc2[pos]) ? 4 : 0) | (!std::isinf(tin[pos]) ? 8 : 0));
            if (sx == 0) f.cin[w] = tin[pos];
            if (sx == N - 1) f.cout[w] = tout[pos];
        }
        if (ok) {
            lf_slot.assign(R, -1);
            for (int k = 0; ok && k < (int)h_end.size(); ++k) {
                const int r = h_end[k];
                if (r % (P + 1) == 0 || lf_slot[r] >= 0) ok = false;    // a non-emitting or duplicated end row: other kernels
                else lf_slot[r] = k;
            }
        }
        if (ok) {
            f.K = K; f.W = W; f.N = N; f.skip = skip ? 1 : 0; f.P = P; f.R = R; f.loop = 0; f.loop_row = 0;
            for (int w = 0; w < GH_LAYERS_MAXW; ++w) f.cin0[w] = INFINITY;
            for (int w = W; w < GH_LAYERS_MAXW; ++w) {
                f.cin[w] = f.cout[w] = INFINITY;
                for (int sx = 0; sx < GH_LAYERS_MAXN; ++sx) f.c0[w][sx] = f.c1[w][sx] = f.c2[w][sx] = INFINITY;
            }
            lt->layers_ok = true;
        }
    }
    // ---- loop form (gh_layerform with loop = 1): the word-loop grammar ----
    if (!lt->layers_ok && L == 1 && !lt->has_nan_arc && !lt->has_self_arc) {
        const int R = lt->lat[0].R;
        gh_layerform& f = lt->h_layers;
        memset(&f, 0, sizeof f);
        int Lr = -1, n_nes = 0;
        for (int r = 0; r < R; ++r) if (h_state[r] < 0) { ++n_nes; if (r > 0) Lr = r; }
        bool ok = R >= 4 && h_state[0] < 0 && n_nes == 2 && Lr > 1 && Lr < R - 1;
        const int W = ok ? R - 1 - Lr : 0;
        ok = ok && W >= 1 && W <= GH_LAYERS_MAXW && (Lr - 1) % W == 0;
        const int N = ok ? (Lr - 1) / W + 1 : 0;
        ok = ok && gh_seq_n_ok(N) && (W <= GH_LAYERS_ROWW || N <= 8);          // (more than 16 words: the wide kernel, N <= 8)
        ok = ok && lt->lat[0].n_start == 1 && (h_start[0] & 1);
        auto row_of = [&](int w, int sx) { return sx == 0 ? Lr + 1 + w : 1 + w * (N - 1) + (sx - 1); };
        std::vector<int> wof(R, -1), sof(R, -1);
        for (int w = 0; ok && w < W; ++w)
            for (int sx = 0; sx < N; ++sx) { wof[row_of(w, sx)] = w; sof[row_of(w, sx)] = sx; }
        for (int w = 0; w < GH_LAYERS_MAXW; ++w) {
            f.cin[w] = f.cout[w] = f.cin0[w] = INFINITY;
            for (int sx = 0; sx < GH_LAYERS_MAXN; ++sx) f.c0[w][sx] = f.c1[w][sx] = f.c2[w][sx] = INFINITY;
        }
        bool skip = false;
        const int32_t* ptr = h_ptr.data();
        for (int r = 0; ok && r < R; ++r)
            for (int p = ptr[r]; ok && p < ptr[r + 1]; ++p) {
                const int o = (int)(h_prow[p] & GH_ARC_ROW);
                const double c = h_pcost[p];
                if (r == 0) { ok = false; break; }                                   // nothing enters the start row
                if (r == Lr) {                                                       // last state of a word -> loop row
                    if (wof[o] < 0 || sof[o] != N - 1 || !std::isinf(f.cout[wof[o]])) { ok = false; break; }
                    f.cout[wof[o]] = c;
                } else if (o == 0 || o == Lr) {                                      // start / loop row -> state 0
                    if (sof[r] != 0) { ok = false; break; }
                    double& t = o == 0 ? f.cin0[wof[r]] : f.cin[wof[r]];
                    if (!std::isinf(t)) { ok = false; break; }
                    t = c;
                    f.arcs[wof[r]][0] |= o == 0 ? 16 : 8;
                } else {                                                             // inside a word: from s, s-1, s-2
                    const int w = wof[r], d = sof[r] - sof[o];
                    if (w < 0 || wof[o] != w || d < 0 || d > 2) { ok = false; break; }
                    double& t = d == 0 ? f.c0[w][sof[r]] : d == 1 ? f.c1[w][sof[r]] : f.c2[w][sof[r]];
                    if (!std::isinf(t)) { ok = false; break; }
                    t = c;
                    f.arcs[w][sof[r]] |= (uint8_t)(1 << d);
                    if (d == 2) skip = true;
                }
            }
        if (ok) {
            lf_slot.assign(R, -1);
            for (int k = 0; ok && k < (int)h_end.size(); ++k) {
                const int r = h_end[k];
                if (h_state[r] < 0 || lf_slot[r] >= 0) ok = false;
                else lf_slot[r] = k;
            }
        }
        if (ok) {
            for (int w = 0; w < W; ++w)
                for (int sx = 0; sx < N; ++sx) f.state[w][sx] = h_state[row_of(w, sx)];
            f.K = 1; f.W = W; f.N = N; f.skip = skip ? 1 : 0; f.P = W * N; f.R = R; f.loop = 1; f.loop_row = Lr;
            lt->layers_ok = true;
        }
    }
    UploadArena ar;
    std::vector<gh_layerform> one(1, lt->h_layers);
    if (lt->layers_ok) { ar.add(&lt->d_layers, one); ar.add(&lt->d_lf_end_slot, lf_slot); }
    if (lt->chain_ok) {
        ar.add(&lt->d_ch_cost0, ch0); ar.add(&lt->d_ch_cost1, ch1); ar.add(&lt->d_ch_cost2, ch2);
        ar.add(&lt->d_ch_info, chinfo); ar.add(&lt->d_ch_end_slot, chslot); ar.add(&lt->d_ch_group_row0, chgroups);
    }
    ar.add(&lt->d_row_state, h_state); ar.add(&lt->d_row_start, h_start);
    ar.add(&lt->d_pred_ptr, h_ptr); ar.add(&lt->d_pred_row, h_prow); ar.add(&lt->d_pred_cost, h_pcost); ar.add(&lt->d_order, h_order);
    ar.add(&lt->d_succ_ptr, h_sptr); ar.add(&lt->d_succ_row, h_srow); ar.add(&lt->d_succ_cost, h_scost);
    ar.add(&lt->d_level_ptr, h_lev); ar.add(&lt->d_level_narrow, h_narrow); ar.add(&lt->d_end_rows, h_end);
    ar.add(&lt->d_desc, lt->h_desc);
    if (lt->fbchain_ok) ar.add(&lt->d_fbchain, lt->h_fbchain);
    if (lt->seq_ok) { ar.add(&lt->d_seqgraphs, lt->h_seqgraphs); ar.add(&lt->d_seqwords, lt->h_seqwords); ar.add(&lt->d_seq_end_slot, seq_slot); }
    else { lt->h_seqgraphs.clear(); lt->h_seqwords.clear(); }
    const int rc = ar.commit(&lt->d_arena);
    if (rc) {
        gh_lattices_destroy(lt);
        return rc;
    }
    *out = lt;
    return GH_OK;
}

extern "C" void gh_lattices_destroy(gh_lattices* l) {
    if (!l) return;
    hipSetDevice(l->ctx->device);
    hipFree(l->d_arena);
    if (l->full) gh_lattices_destroy(l->full);
    if (l->deferred_src) gh_transcripts_src_free(l->deferred_src);
    delete l;
}

extern "C" int gh_lattices_set_beam(gh_lattices* l, int beam) {
    GH_REQUIRE(l, "gh_lattices_set_beam: NULL argument");
    l->beam = beam > 0 ? beam : 0;
    return GH_OK;
}

extern "C" int gh_lattices_forms(const gh_lattices* l) {
    if (!l) return -1;
    const bool loop = l->layers_ok && l->h_layers.loop;
    return (l->chain_ok ? 1 : 0) | (l->layers_ok && !loop ? 2 : 0) | (loop ? 4 : 0) | (l->seq_ok ? 8 : 0) | (l->fbchain_ok ? 16 : 0);
}

extern "C" int64_t gh_viterbi_path_cap(const gh_lattices* lat, int l, int64_t T) {
    if (!lat || l < 0 || l >= lat->L) return -1;
    return T * lat->lat[l].nlev;  // at most one cell per (column, level)
}

