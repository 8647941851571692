#pragma once
#include "gh_internal.h"

// Kernel argument block of the lattice Viterbi (all pointers are device pointers).
struct gh_vit_args {
    const gh_lattices::desc* descs;
    const int32_t* row_state;
    const uint8_t* row_start;
    const int32_t* pred_ptr;
    const uint32_t* pred_row;
    const double* pred_cost;
    const int32_t* order;
    const int32_t* level_ptr;
    const int32_t* level_narrow;  // rows of each level with <= 2 arcs (first in `order`)
    const int32_t* end_rows;
    const void* nll;  // [N,S] float or double
    int S;
    int r_pad;                // LDS column stride (>= max R, even)
    int em_chunk;             // lean kernel: columns of emissions prefetched per chunk, >= 1
    int arc_cap;              // lean kernel: LDS slots for a graph's arc list
    int lds_bytes;            // lean kernel: dynamic LDS of the launch (the back-trace re-uses all of it)
    int bpc_off;              // lean kernel: LDS byte offset of the 8-column back-pointer staging block
    int beam;                 // generic kernel: rank beam per column (0 = off), gh_lattices_set_beam
    const int64_t* utt_off;   // [U+1] frame offsets
    const int32_t* utt_lat;   // [U] graph of each utterance, or null (graph 0)
    const int64_t* perm;      // launch slot -> utterance (longest first), or null
    int64_t u_begin;          // first launch slot of this chunk
    uint16_t* bp;             // back-pointer scratch of this chunk
    const int64_t* bp_off;    // [slots] offset of each launch slot's [T,R] block
    double* end_cost;         // [sum n_end]
    const int64_t* end_off;   // [U]
    int32_t* best_end;        // [U]
    int32_t* path;            // [.., 2]
    const int64_t* path_off;  // [U+1]
    int32_t* path_len;        // [U]
    double* costs;            // optional full matrices
    const int64_t* costs_off; // [U+1]
    int* flag;
};

int gh_launch_viterbi(gh_ctx* ctx, const gh_vit_args& a, int64_t n_utts, int block, size_t lds_bytes,
                      bool f64, bool want_path);
// lean kernel (gh_viterbi_lean.hip): <= 3 levels, one row per lane per level, no NaN / self arcs
int gh_launch_viterbi_lean(gh_ctx* ctx, const gh_vit_args& a, int64_t n_utts, int block, size_t lds_bytes,
                           bool f64, bool want_path, int levels);

// Chain kernel (gh_viterbi_chain.hip): left-to-right graphs (arcs from r, r-1, r-2 only), one graph
// for the whole batch.  All pointers are device pointers.
struct gh_chain_args {
    const double* cost0;      // [R] self-arc cost, +inf = absent
    const double* cost1;      // [R] arc from r-1
    const double* cost2;      // [R] arc from r-2
    const uint8_t* row_info;  // [R] bits 0-1: code of the first (lowest-origin) arc, 3 = none; bit 2: start row
    const int32_t* row_state; // [R]
    const int32_t* end_slot;  // [R] position in the end list or -1
    const int32_t* end_rows;  // [n_end]
    const int32_t* group_row0;  // [n_groups+1] row ranges of the 64-lane groups (whole chains)
    int n_groups, R, S, n_end;
    const void* nll;
    const int64_t* utt_off;
    const int64_t* perm;
    int64_t slot0;
    uint8_t* bp;              // [T,R] bytes per launch slot
    const int64_t* bp_off;
    double* end_cost;         // [U, n_end]
    int32_t* best_end;        // [U]
    int32_t* path;
    const int64_t* path_off;
    int32_t* path_len;
    double* costs;
    const int64_t* costs_off;
    int* flag;
};
int gh_launch_viterbi_chain(gh_ctx* ctx, const gh_chain_args& a, int64_t u_begin, int64_t n_utts, bool f64,
                            bool want_bp, bool want_costs, bool skip);
int gh_launch_chain_backtrace(gh_ctx* ctx, const gh_chain_args& a, int64_t u_begin, int64_t n_utts);

// Fused single-Gaussian decode (gh_viterbi_fused.hip): the chain kernel's graph and outputs, but every lane scores its
// own state's Gaussian against the frame instead of reading the [N, S] likelihood matrix (c.nll / c.S unused).
struct gh_fused_args {
    gh_chain_args c;
    const void* feats;   // [N, D] features of the batch's dtype
    const double* par;   // [2 DVp + 2][Rp]: sqrt(1/(2 var)) rows, -mean sqrt(1/(2 var)) rows, -logc, underflow threshold
    int D, Rp, skip;     // feature dimension, padded row count, any r-2 -> r arc
    int DVp;             // rows per half of the constants table (gh_fused_dv(D))
    int select_end;      // 1: the sweep also picks the best end row (one lane group, no path wanted): no back-trace launch
    int lin;             // 1: the linear-domain underflow rule of GMM.evaluate is on (finite threshold in the table)
    int64_t n_items;     // (utterance, row group) pairs of the launch
    int pw, period;      // packed form: rows per wave window (0: one utterance per wave), windows per repeat of the row pattern
    int64_t n_windows;
};
int gh_fused_window(int R, int unit, int* period_out);   // rows per packed window (0: one utterance per wave)
int gh_fused_dv(int D);  // rows per half of the constants table for D dimensions (0: D not covered)
int gh_launch_fused_params(gh_ctx* ctx, const gh_gmm* g, const int32_t* d_row_state, int R, int Rp, int DV, double thr,
                           double* d_par);
int gh_launch_viterbi_fused(gh_ctx* ctx, const gh_fused_args& fa, int64_t u_begin, int64_t n_utts, bool f64,
                            bool want_bp, bool want_costs);

// Layer-form kernel (gh_viterbi_layers.hip): K identical layers of W words x N states (gh_layerform), one graph for
// the whole batch, one wave per utterance.  All pointers are device pointers.
struct gh_layers_args {
    const gh_layerform* lf;
    const int32_t* end_slot;   // [R] position of a row in the end list or -1
    const int32_t* end_rows;   // [n_end]
    int n_end, S;
    const void* nll;
    const int64_t* utt_off;
    const int64_t* perm;
    int64_t slot0;
    uint16_t* bp;              // decision words; bp_off in uint16 units (multiples of 8)
    const int64_t* bp_off;
    double* end_cost;          // [U, n_end]
    int32_t* best_end;         // [U]
    int32_t* path;
    const int64_t* path_off;
    int32_t* path_len;
    // sequence form (gh_seq.hip): per-utterance graphs made of shared word templates
    const gh_seqgraph* seqgraphs;
    const gh_seqword* seqwords;
    const int32_t* utt_lat;    // [U] graph of every utterance, or null (graph 0)
    const int64_t* end_off;    // [U] offset of the utterance's end costs, or null (u * n_end)
    int seq_N;
    const int32_t* row_label;  // label mode (gh_viterbi_labels): label per row, < 0 on non-emitting rows
    int32_t* labels;           // utterance u at label_off[u]
    const int64_t* label_off;  // [U+1]
    int32_t* n_labels;         // [U]
    int* flag;
};
size_t gh_layers_bp_entries(const gh_layerform& f, int64_t T);
int gh_launch_viterbi_layers(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts,
                             bool f64, bool want_path);
int gh_launch_lattice_backtrace(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts);
// more than GH_LAYERS_ROWW words per layer (gh_viterbi_layers_wide.hip: lane = word); the two launchers above hand over to these
size_t gh_layers_wide_bp_entries(int64_t T);
size_t gh_loop_wide_bp_entries(int64_t T);
int gh_launch_viterbi_layers_wide(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts,
                                  bool f64, bool want_path);
int gh_launch_lattice_backtrace_wide(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts);
// sequence form (forced-alignment lattices, gh_seq.hip): forward sweep, four utterances per wave, and its back-trace
size_t gh_seq_bp_entries(int N, int skip, int64_t T);
int gh_launch_viterbi_seq(gh_ctx* ctx, const gh_layers_args& a, int N, int skip, int64_t u_begin, int64_t n_utts, bool f64,
                          bool want_path);
int gh_launch_seq_backtrace(gh_ctx* ctx, const gh_layers_args& a, int N, int skip, int64_t u_begin, int64_t n_utts);
