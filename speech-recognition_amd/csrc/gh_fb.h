#pragma once
#include "gh_internal.h"

// Kernel argument block of the forward-backward kernel (device pointers).
struct gh_fb_args {
    const gh_lattices::desc* descs;
    const int32_t* row_state;
    const uint8_t* row_flag;  // bit0 = start row, bit1 = end row
    const int32_t* pred_ptr;
    const uint32_t* pred_row;
    const double* pred_cost;
    const int32_t* succ_ptr;
    const uint32_t* succ_row;
    const double* succ_cost;
    const int32_t* order;
    const int32_t* level_ptr;
    const int32_t* end_rows;
    const void* nll;
    int S;
    int r_pad;
    int arc_cap;                 // LDS slots for a graph's arc lists (even)
    int lev_cap;                 // LDS slots for a graph's level offsets
    const int64_t* utt_off;
    const int32_t* utt_lat;
    const int64_t* perm;
    int64_t u_begin;
    double* alpha_scratch;       // [T,R] per launch slot
    const int64_t* scratch_off;  // [slots]
    double* logp;                // [U]
    double* out_alpha;           // optional [R,T] per utterance at mat_off[u]
    double* out_beta;
    double* out_gamma;
    const int64_t* mat_off;      // [U+1]
    double* occ;                 // optional [N,S] frame x state occupancies
    double* self_xi;             // optional [S]: expected number of self transitions per state (double atomics)
};

int gh_launch_fb(gh_ctx* ctx, const gh_fb_args& a, int64_t n_utts, int block, size_t lds_bytes, bool f64);

// one-word chain graphs: one lane per utterance (fb_chain_kernel)
struct gh_fbchain_args {
    const gh_fbchain* chains;    // [L]
    const void* nll;
    int S;
    const int64_t* utt_off;
    const int32_t* utt_lat;      // or null (graph 0)
    const int64_t* perm;         // launch slot -> utterance (longest first)
    int64_t U;
    double* alpha_scratch;       // [T, n] per launch slot
    const int64_t* scratch_off;  // [slots]
    double* logp;                // [U]
    double* occ;                 // optional [N,S], zeroed by the caller
    double* gam;                 // optional [N, lanes]: gamma compact, column = chain row (instead of occ)
    double* self_xi_utt;         // optional [U, GH_FBCHAIN_MAX]: expected self transitions of every chain row of every utterance
                                 //   (entries behind the utterance's chain length are not written)
    int lanes;                   // 8 or 16 lanes per utterance (gh_fbchain_lanes): >= the longest chain
    int32_t* occ_rng;            // optional [U, GH_FBCHAIN_MAX, 2] (with gam): first / last frame (utterance-local) of every chain
    double rng_floor;            //   row whose gamma exceeds rng_floor (or is NaN); none: (T, -1) -- the block lists of the
                                 //   fused statistics kernel (gh_bw_fused.hip) without a pass over gamma
};
int gh_launch_fb_chain(gh_ctx* ctx, const gh_fbchain_args& a, bool f64);
// Two-way form (fb_chain2_kernel): compact gamma only (a.gam), forward and backward recursions side by side in one wave.
// Scratch per utterance in doubles, T x n cells (both forms keep an utterance's piece on cache lines of its own):
// the form gh_launch_fb_chain takes: two-way for compact gamma on batches too small to fill the chip with one-way waves
bool gh_fbchain_two_way(const gh_ctx* ctx, bool compact_gamma, bool occupancy_matrix, int64_t U, int lanes);
inline size_t gh_fbchain_scratch(size_t cells, bool two_way) {
    const size_t need = two_way ? 3 * cells + 2       // alpha, beta mantissas [T, n] each + both exponent arrays (int32) + P
                                : cells + (cells + 1) / 2;   // alpha mantissas + exponents
    return (need + 15) & ~size_t(15);
}

// forced-alignment graphs in sequence form (gh_seqgraph): four utterances per wave, lane = layer (fb_seq_kernel, gh_seq.hip)
#define GH_FBSEQ_XI_PARTS 256
struct gh_fbseq_args {
    const gh_seqgraph* graphs;
    const gh_seqword* words;
    const int32_t* end_slot;     // [rows of all graphs] >= 0 on end rows
    const void* nll;
    int S;
    const int64_t* utt_off;
    const int32_t* utt_lat;      // or null (graph 0)
    const int64_t* perm;
    int64_t slot0;
    double* alpha_scratch;       // [T, K, N] per launch slot
    const int64_t* scratch_off;  // [slots]
    double* logp;                // [U]
    double* occ;                 // optional [N,S]
    int occ_in_lds;              // 1: a frame's S occupancies are summed in LDS and stored as one row (every frame of every
                                 //    utterance is written: no zeroing needed); 0: double atomics on occ, zeroed by the caller
    double* self_xi_parts;       // optional [GH_FBSEQ_XI_PARTS, S], zeroed by the caller; summed by the caller
    int32_t* seg_lo;             // optional [U, GH_SEQ_MAXK] (with occ): first / last frame of every layer whose occupancy
    int32_t* seg_hi;             //   exceeds occ_floor (hi < lo: none) -- the segments of the fused statistics kernel
    double occ_floor;
    int max_cells;               // most cells (layers x states) any graph of the launch has; 0 = not known.  <= 64: one
                                 //   utterance per wave, lane = cell (fb_seq_cell_kernel) instead of four with lane = layer
    int32_t* row_lo;             // optional [U, GH_SEQ_MAXK, GH_LAYERS_MAXN] (lane = cell kernel only): frame range of every
    int32_t* row_hi;             //   cell with occupancy above occ_floor
};
int gh_fb_seq_by_cell(const gh_fbseq_args& a, int N);
int gh_launch_fb_seq(gh_ctx* ctx, const gh_fbseq_args& a, int N, int skip, int64_t u_begin, int64_t n_utts, bool f64);
