// Streaming matrix-core kernels of the device-resident refit (gh_refit_mfma.hip), launched by gh_lockstep.hip's
// gh_fit_kmeans / gh_fit_em: the lock-step E-step of hmm_state.py:122-159 and the assignment sweep of kmeans.py:180-186
// with their per-state sums on v_mfma_f64_4x4x4_4b_f64, the per-iteration reduction, M-step / centroid update, stop rule
// and the next iteration's operands folded into the kernel's tail (last workgroup of a state to arrive).
#pragma once
#include "gh_internal.h"

// One workgroup's share of a state's frames: [first, first + count) of the gathered batch, count <= RF_ITEM_FRAMES
// (a state without frames still has one item, count 0: its tail runs the update the reference would run on no data).
struct rf_item { int64_t first; int32_t state, count; };

constexpr int RF_WAVES = 4;            // waves per workgroup; a wave takes the 16-frame slabs wave, wave + 4, ...
constexpr int RF_MAXCG = 8;            // component groups of 4: k <= 32

// geometry shared by host and device: features are padded with the constant column 1 (feature D) to KS steps of 4
__host__ __device__ inline int rf_steps(int D) { return (D + 4) >> 2; }              // ceil((D + 1) / 4)
__host__ __device__ inline int rf_row_stride(int D) { return 4 * rf_steps(D) + 2; }   // LDS doubles per frame: = 2 mod 4
__host__ __device__ inline int rf_col_groups(int D) { return (D + 16) >> 4; }        // ceil((D + 1) / 16)
__host__ __device__ inline int rf_comp_groups(int k) { return (k + 3) >> 2; }
__host__ __device__ inline int rf_em_pstride(int k, int D) { return rf_comp_groups(k) * 2 * rf_steps(D) * 16; }
__host__ __device__ inline int rf_km_pstride(int k, int D) { return (rf_comp_groups(k) + 1) * rf_steps(D) * 16; }

struct rf_common {
    const double* X;            // [N, D] gathered frames
    int D, k, S;
    const rf_item* items;
    const int32_t* item_ptr;    // [S + 1]
    const double* shift;        // [S, D]: the frames and parameters of a state are taken relative to this point
    uint8_t* active;            // [S]
    int32_t* done;              // [S] arrival counters (zero between launches)
    double* partial;            // [n_items][plen]
    int* counter;               // [0] unused, [1] error bits, [2 + (it & 7)] states still active after iteration it
    int it;                     // iteration number of this launch (its first one, with n_iter > 1)
    int fused;                  // 1: the last arriver also runs the update + packs the next operands (no collective in between)
    // TAIL launches (few states left, every one of their workgroups resident at once): the grid is the items item_ids[0 ..]
    // and a workgroup stays for up to n_iter iterations, waiting on its state's generation word between them
    const int32_t* item_ids;    // null: the grid is all items
    int32_t* gen;               // [S] zero at launch; j + 1 when iteration j of the launch is done for the state, -1 when it stopped
    int n_iter;                 // iterations this launch runs (1: the ordinary launch)
};

struct rf_em_args {
    rf_common c;
    double* P;                  // [S][rf_em_pstride]: packed operands, scaled log domain (GH_LSE_SCALE64)
    const double* exp_tab;      // [128] 2^(j/128)
    double* stats;              // [S][k (1 + 2D) + 1] centred on `shift`: occupancy | sum r (x - shift) | sum r (x - shift)^2
    // update (hmm_state.py:134-159)
    const double* nframes;
    double *mean, *var, *weight, *old_mu, *old_sigma, *old_w;
    int32_t* conv_at;
    const int32_t* hard_ids;    // HARD pass only: the group of every frame (the random partition)
};

struct rf_km_args {
    rf_common c;
    double* P;                  // [S][rf_km_pstride]
    double* kscale;             // [S] magnitude of the terms every distance is made of (the tie band is relative to it)
    const double* cent_in;      // unused by the kernel proper (the exact re-test reads `cent`)
    double* cent;               // [S, k, D]
    const double* var;          // [S, k, D]: row 0 of a state is the variance of its distance (kmeans.py:183)
    const double* logdet;       // [S]
    int32_t* ids;               // [N] in/out
    double* sums;               // [S][k (D + 1) + 1] centred on `shift`: sum (x - shift) | count per cluster, then the changed count
    int32_t* iters;
};

int rf_launch_em(gh_ctx* ctx, const rf_em_args& a, int n_items);
int rf_launch_em_update(gh_ctx* ctx, const rf_em_args& a, int pack_only);      // one block per state: (update +) pack
int rf_launch_partition_sums(gh_ctx* ctx, const rf_em_args& a, int n_items);   // count | sum x' | sum x'^2 of given groups -> a.stats
int rf_launch_partvar(gh_ctx* ctx, int S, int k, int D, const double* stats, int first_only, double* cov);
int rf_launch_km(gh_ctx* ctx, const rf_km_args& a, int n_items);
int rf_launch_km_update(gh_ctx* ctx, const rf_km_args& a, int pack_only);
bool rf_supported(int k, int D);
size_t rf_em_lds(int k, int D);
size_t rf_km_lds(int k, int D);
