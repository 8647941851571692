#pragma once
#include "gh_internal.h"

// Kernel argument block of the template DP (device pointers).
struct gh_dtw_args {
    const double* x;         // [N,D] frames (fp64) or null when E is given
    const double* E;         // optional caller-supplied distance matrices, utterance u at e_off[u], [n,T_u]
    const int64_t* e_off;    // [U]
    const int64_t* utt_off;  // [U+1]
    int n, D, beam;          // beam <= 0: no pruning
    const double* y;         // [n,D] template rows
    const double* var;       // [n,D] per-row variance or null (Euclidean)
    const double* logdet;    // [n] 0.5*log((2pi)^D prod var)
    const double* trans;     // [n,n] dense, +inf = no arc
    uint8_t* bp;             // scratch: origin row per cell, bytes for n <= 255, uint16 above
    const int64_t* bp_off;   // [U]
    double* costs;           // optional [n,T] per utterance
    const int64_t* costs_off;
    int32_t* path;
    const int64_t* path_off;  // [U+1]
    int32_t* path_len;
    int* flag;
    // several template sets in one launch (gh_fit_dtw: the segmental k-means of many word models in lock-step):
    // utterance u is matched against set utt_model[u] -- y / var / logdet / trans are then [W, ...] arrays -- and sets
    // whose model_active entry is 0 are skipped; frame_row (optional, [N]): the template row every frame is aligned to
    // (the path as one int per frame: rows of columns 0 .. T-2 from the back-trace, the last frame on row n - 1)
    const int32_t* utt_model;
    const uint8_t* model_active;
    int32_t* frame_row;
};

int gh_launch_dtw(gh_ctx* ctx, const gh_dtw_args& a, int64_t U);
