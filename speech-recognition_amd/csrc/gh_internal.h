// Internal structures of libgmmhmm (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/gmmhmm.h"

void gh_set_error(const char* fmt, ...);

#define GH_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            gh_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return (e_ == hipErrorOutOfMemory) ? GH_ERR_NOMEM : GH_ERR_HIP;              \
        }                                                                                \
    } while (0)

#define GH_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            gh_set_error(__VA_ARGS__);        \
            return GH_ERR_INVALID;            \
        }                                     \
    } while (0)

// lean Viterbi kernel: rows with up to this many arcs keep them in registers (one lane per row); rows with more
// (non-emitting rows collecting every word end) get 16 lanes and an LDS-resident arc list
#define GH_LEAN_NARROW_ARCS 3

// Scaled log domain of the MFMA likelihood kernel (gh_loglik_mfma.hip): the packed operands
// Apk / Cpk carry K * (log-density terms), K = 128/ln2 (fp64) or 1/ln2 (fp32); a component that
// is switched off (weight 0, mixture padding) carries the finite constant OFF instead of -inf.
#define GH_LSE_SCALE64 184.66496523378731614
#define GH_LSE_SCALE32 1.4426950408889634074
#define GH_LSE_OFF64 (-1e300)
#define GH_LSE_OFF32 (-1e30f)

struct gh_ctx {
    int device;
    hipStream_t stream;
    int n_cu;
    // growable device scratch (back-pointers, partial statistics, ...)
    void* scratch;
    size_t scratch_bytes;
    int* d_flag;  // device error flag (self-pointing DP cell etc.)
    void* pinned;         // growable page-locked host staging buffer (one D2H copy per call instead of several)
    size_t pinned_bytes;
    double* d_fp64_tables;  // [384] exp2 / inv / -log tables of the fp64 log-sum-exp (gh_loglik_mfma.hip)
    int compat;             // gh_ctx_set_compat: bit 0 = linear-domain underflow of the reference's GMM.evaluate (+inf)
    int last_fused = 0;     // 1: the last gh_viterbi_fused call on this context ran the fused kernel (0: gh_loglik + gh_viterbi)
    int last_chunks;        // launches the last gh_viterbi* / gh_forward_backward call on this context was cut into (scratch budget)
    size_t budget_cache = 0;     // gh_scratch_budget's last answer (hipMemGetInfo is ~0.2 ms: asked again only when the arena
    int budget_age = 0;          //   changes size, an allocation fails, or after 64 calls)
    // the device arena and the two page-locked blocks of the refit session closed last, for the next gh_fit_create on this
    // context (continuous_train opens one session per outer iteration: hipMalloc + 2 hipHostMalloc + their frees were
    // ~1.5 ms of each); an arena above GMMHMM_FIT_KEEP_MB (default 4096) is not kept
    void* fit_arena = nullptr;
    size_t fit_arena_bytes = 0;
    void* fit_pin = nullptr;
    void* fit_act = nullptr;
    size_t fit_act_bytes = 0;
};

int gh_scratch(gh_ctx* ctx, size_t bytes, void** out);
size_t gh_scratch_budget(gh_ctx* ctx, bool fresh = false);   // fresh: ask the driver again
int gh_pinned(gh_ctx* ctx, size_t bytes, void** out);

// Parameter layout shared by the likelihood kernels (GEMM form):
//   ll[g] = C[g] + sum_d ( A[g,d] * x_d^2 + B[g,d] * x_d )
//   A = -0.5/var, B = mean/var, C = log w - 0.5*(D log 2pi + sum log var + sum mean^2/var)
// g = s*M + m.  KP = D rounded up to a multiple of 4 (zero padded).
struct gh_gmm {
    gh_ctx* ctx;
    void* d_arena;   // the one device allocation all the d* pointers below point into
    int S, M, D, KP;
    std::vector<double> hA, hB, hC;  // host fp64 master copies [G,KP], [G,KP], [G] (of gh_gmm_create's parameters)
    bool host_stale = false;         // gh_gmm_update[_dev] rewrote the device arrays: hA / hB / hC describe an older model
    double *dA64, *dB64, *dC64;      // device fp64
    float *dA32, *dB32, *dC32;       // device fp32 -- for CENTRED features x - cen (see dCen32)
    // fp32 only: the GEMM form cancels terms of size (x^2 + mean^2)/var, which costs fp32 its digits once the
    // features sit far from zero.  The likelihood is shift invariant, so the fp32 operands are packed for
    // x' = x - cen, mean' = mean - cen with cen = the average component mean per dimension (rounded to fp32: the
    // kernels subtract exactly the value the host packed for); fp64 operands stay un-centred (cen = 0).
    float* dCen32;                   // [KP]
    // plain parameters (fp64) for the training kernels: mean, inv_var [G,D], logc [G]
    double *dMean, *dIvar, *dLogc;
    // 1 when some component's log(w * normaliser) is positive (tight variances): the linear-domain underflow rule of
    // GMM.evaluate then needs the per-component test of gh_loglik_underflow_fix (written by gh_gmm_create and by
    // every device-side re-pack)
    int* dAnyPos = nullptr;
    int any_pos_host = -1;   // what the host knows of *dAnyPos: 0 / 1 after gh_gmm_create, -1 after a re-pack on the device
    // MFMA operand packing (gh_loglik_mfma.hip): mixtures padded to M_pad components,
    // Gaussians to n_tiles*16 rows; Apk[tile][kstep][lane] fragments, Cpk[tile*16 + j]
    int M_pad, n_tiles;
    double *dApk64, *dCpk64;
    float *dApk32, *dCpk32;
    // stamp of the parameter set (new at creation and at every in-place update): a batch remembers whose likelihoods its
    // [N, S] matrix holds (gh_batch::nll_serial), so that a kernel that reuses them as mixture denominators can tell
    uint64_t serial = 0;
};
uint64_t gh_next_serial();

// A graph that is ONE left-to-right chain of <= 16 emitting rows with distinct states (arcs from r, r-1, r-2 in the
// previous column), started at its first row (directly, or through a non-emitting start row: cost c0) and ended
// at its last row -- what a one-word forced-alignment lattice is.  gh_forward_backward runs these with one LANE
// per chain row (fb_chain_kernel): 8 lanes per utterance when every chain of the call has <= 8 rows (BASELINE
// configs[1-2]: 5), 16 otherwise (configs[3]: 16) -- gh_fbchain_lanes; the recursion lives in registers.
#define GH_FBCHAIN_MAX 16
struct gh_fbchain {
    int32_t n, pad;                       // pad = 1 when the chain has r-2 -> r (skip) arcs
    int32_t state[GH_FBCHAIN_MAX];
    double c0;
    double self_c[GH_FBCHAIN_MAX], next_c[GH_FBCHAIN_MAX], skip_c[GH_FBCHAIN_MAX];   // +inf = no such arc
};

struct gh_batch {
    gh_ctx* ctx;
    gh_dtype dtype;
    int D;
    int64_t N, U;
    void* feats;  // device [N,D]
    bool owns_feats;
    std::vector<int64_t> offsets;  // host [U+1]
    int64_t* d_offsets;            // device [U+1]
    int64_t max_T;
    bool any_T1 = false;           // some utterance has exactly one frame (the reference's column wrap: special kernels)
    void* nll;  // device [N,S] (dtype) after gh_loglik
    int nll_S;
    uint64_t nll_serial = 0;   // gh_gmm::serial of the model whose likelihoods `nll` holds (0: none yet)
    double* occ;  // device [N,occ_S] fp64 frame x state occupancies after gh_forward_backward(want_occ)
    int occ_S;    // state count `occ` was allocated for (reallocated when a model of another size is used)
    // after a chain-form forward-backward: the states that can carry occupancy in each utterance ([U][8], -1 padded),
    // so that gh_bw_accumulate need not scan the occupancy matrix for them; null otherwise
    int32_t* d_occ_states;
    // after a chain-form forward-backward that nobody asked the full matrix of: gamma compact, one column per chain
    // row ([N, gam_lanes], gam_lanes = 8 or 16: gh_fbchain_lanes), the chains it belongs to and every utterance's chain
    // -- what the fused Baum-Welch statistics kernel consumes (gh_bw_fused.hip); `occ` is then stale (occ_valid false)
    double* gam;
    int gam_lanes = 0;         // columns `gam` was allocated for
    bool occ_valid;
    std::vector<gh_fbchain> gam_chains;
    std::vector<int32_t> gam_utt_graph;
    // launch order of the DP kernels: utterances sorted longest first (computed once)
    std::vector<int64_t> perm;
    int64_t* d_perm;
    // after a sequence-form forward-backward with occupancies (gh_seq.hip): per (utterance, layer) the frame range that
    // carries occupancy, the word of every layer and the word chains -- what lets the fused Baum-Welch statistics
    // kernel walk (utterance, layer) segments grouped by word and read gamma from `occ` (gh_bw_fused.hip)
    bool seq_seg_valid = false;
    std::vector<int32_t> seq_seg_lo, seq_seg_hi;      // [U, GH_SEQ_MAXK] first / last frame (utterance-local), hi < lo: none
    std::vector<int32_t> seq_utt_K, seq_utt_word;     // [U], [U, GH_SEQ_MAXK] word template of every layer
    std::vector<gh_fbchain> seq_word_chains;          // per word template: its states as a chain (costs unused)
    // k-means assignments that stay on the device between lock-step iterations (gh_kmeans_assign_multi with
    // clusters_io == NULL; gh_kmeans_resident_clusters resets / fetches them)
    int32_t* d_clusters = nullptr;
};

// arc flag bits stored in pred_row
#define GH_ARC_SAME 0x80000000u  // reads the SAME column (touches a non-emitting row)
#define GH_ARC_DEAD 0x40000000u  // same-column origin >= destination: reads +inf
#define GH_ARC_ROW 0x3fffffffu

struct gh_lattice_host {
    int R, A, nlev, n_start, n_end;
    int64_t row_base, arc_base, end_base;  // offsets into the concatenated device arrays
    int max_state;
};


// Layer form of a word lattice (build_state_sequences, continuous_speech.py:13-53, with the SAME W words in each of
// its K layers -- the decode lattice of main.py:35): row 0 = non-emitting start; layer k = rows
// k*(P+1)+1 .. k*(P+1)+P (P = W*N: word w, state s at offset w*N + s), followed by one non-emitting row (k+1)*(P+1);
// inside a word only arcs from s, s-1, s-2 (previous column); non-emitting row k -> state 0 of every word of layer k
// and last state of every word of layer k -> non-emitting row k+1 (same column, decode.py:109-111); identical costs
// and states in every layer.  gh_viterbi runs such graphs with one WAVE per utterance: lane = (layer mod 4, word),
// the N states of a word in registers (gh_viterbi_layers.hip).
#define GH_LAYERS_MAXW 64          // words per layer of the layer form: up to GH_LAYERS_ROWW on the narrow kernel (lane = (layer, word)),
#define GH_LAYERS_ROWW 16          //   up to 64 on the wide one (lane = word; K <= 8, N <= 8: gh_viterbi_layers_wide.hip); loop form alike
#define GH_LAYERS_MAXN 16          // states per word the word templates hold (Viterbi, all three forms: 2..8, 12, 16: gh_seq_n_ok)
#define GH_LAYERFORM_MAXN 8        // ... that the sequence-form forward-backward (and the EM session on it) is built for
#define GH_LAYERS_MAXK 16         // layers of the layer form: up to 8 with any word model above, 9 .. 16 with up to 8 states per word
// states per word the word-template Viterbi kernels are instantiated for (gh_seq.hip, gh_viterbi_layers.hip): every count up to 8, and the 12 and 16
// of wide word models (BASELINE configs[3]: 16 states per word), whose N costs still live in one lane's registers
static inline bool gh_seq_n_ok(int N) { return (N >= 2 && N <= 8) || N == 12 || N == 16; }
// LOOP form (K = 1, loop = 1): the word-loop grammar of continuous_speech.build_loop_grammar -- row 0 = non-emitting
// start; rows 1 .. W*(N-1) = states 1..N-1 of every word; row loop_row = 1 + W*(N-1) = the non-emitting loop row;
// rows loop_row+1+w = state 0 of word w.  Last states feed the loop row, the start row (cost cin0) and the loop row
// (cost cin) feed the first states, all in the same column.  gh_viterbi runs it with FOUR utterances per wave: DPP
// row = utterance, lane = word (gh_viterbi_layers.hip, viterbi_loop_kernel).
struct gh_layerform {
    int32_t K, W, N, skip;      // layers, words per layer, states per word, any s-2 arc
    int32_t P, R, loop, loop_row;
    int32_t state[GH_LAYERS_MAXW][GH_LAYERS_MAXN];
    uint8_t arcs[GH_LAYERS_MAXW][GH_LAYERS_MAXN];   // bit0 self, bit1 from s-1, bit2 from s-2, bit3 from the non-emitting row
    double c0[GH_LAYERS_MAXW][GH_LAYERS_MAXN], c1[GH_LAYERS_MAXW][GH_LAYERS_MAXN], c2[GH_LAYERS_MAXW][GH_LAYERS_MAXN];
    double cin[GH_LAYERS_MAXW], cout[GH_LAYERS_MAXW];   // +inf = no such arc
    double cin0[GH_LAYERS_MAXW];                        // loop form: arc from the start row into state 0 (arcs bit4)
};

// SEQUENCE form: a forced-alignment lattice (continuous_speech.py:80: build_state_sequences(models, [[l] for l in labels])) --
// K <= 16 layers with ONE word each (different words in different layers), rows as in the layer form with W = 1:
// row 0 non-emitting start, layer k = rows k*(N+1)+1 .. k*(N+1)+N, non-emitting row (k+1)*(N+1) behind it.  Word
// templates are shared between graphs (thousands of transcripts use the same few words).  gh_viterbi and
// gh_forward_backward run such graphs with FOUR utterances per wave: DPP row = utterance, lane = layer, the N states
// of the layer's word in registers, the non-emitting row handed to the next lane by row_shr:1 (gh_seq.hip).
#define GH_SEQ_MAXK 16
struct gh_seqword {
    int32_t state[GH_LAYERS_MAXN];
    uint8_t arcs[GH_LAYERS_MAXN];          // bit0 self, bit1 from s-1, bit2 from s-2, bit3 from the non-emitting row
    double c0[GH_LAYERS_MAXN], c1[GH_LAYERS_MAXN], c2[GH_LAYERS_MAXN];
    double cin, cout;
};
struct gh_seqgraph {
    int32_t K, n_end;
    int64_t row_base, end_base;
    int32_t word[GH_SEQ_MAXK];             // template of every layer's word
};

struct gh_transcripts_src;   // gh_transcripts.hip: what a handle made by gh_lattices_create_transcripts was made from
void gh_transcripts_src_free(gh_transcripts_src* s);

struct gh_lattices {
    gh_ctx* ctx;
    // gh_lattices_create_transcripts: only the sequence form (and the row / end tables) is filled in; the row-per-lane
    // arrays live in `full`, an ordinary handle expanded from `deferred_src` the first time a fallback needs it
    gh_transcripts_src* deferred_src = nullptr;
    gh_lattices* full = nullptr;
    void* d_arena;   // the one device allocation all the d_* pointers below point into
    int L;
    std::vector<gh_lattice_host> lat;
    // concatenated device arrays
    int32_t* d_row_state;  // [Rtot] state or -1
    uint8_t* d_row_start;  // [Rtot] 1 = start row
    int32_t* d_pred_ptr;   // [Rtot + L] CSR per graph (R_l + 1 entries each, local to arc_base)
    uint32_t* d_pred_row;  // [Atot] origin row | flags, ascending origin per destination
    double* d_pred_cost;   // [Atot]
    // transposed arcs for the backward pass: CSR by ORIGIN row, same flag bits on the destination
    int32_t* d_succ_ptr;   // [Rtot + L]
    uint32_t* d_succ_row;  // [Atot] destination row | flags
    double* d_succ_cost;   // [Atot]
    int32_t* d_order;      // [Rtot] rows sorted by (level, row)
    int32_t* d_level_ptr;  // per graph: nlev+1 entries at lev_base
    int32_t* d_level_narrow;  // per graph, same indexing: rows of the level with <= 2 arcs (they come first in d_order)
    int32_t* d_end_rows;   // [Etot]
    // per-graph descriptor table on device
    struct desc {
        int32_t R, nlev, n_end, pad;  // pad = max rows of a level
        int32_t lean_lanes, pad2;     // max over levels of (narrow rows + 16 * wide rows): lean kernel block
        int64_t row_base, ptr_base, arc_base, lev_base, end_base;
    };
    desc* d_desc;
    std::vector<desc> h_desc;
    int max_R, max_nlev;
    // chain form (gh_viterbi_chain.hip), only for L == 1 graphs whose arcs all come from r, r-1, r-2
    bool chain_ok, chain_skip;
    int chain_groups;
    int chain_unit = 0;              // all chains of the chain form have this many rows (0: lengths differ)
    double *d_ch_cost0, *d_ch_cost1, *d_ch_cost2;
    uint8_t* d_ch_info;
    int32_t *d_ch_end_slot, *d_ch_group_row0;
    bool fbchain_ok;                 // every graph is a gh_fbchain
    std::vector<gh_fbchain> h_fbchain;
    gh_fbchain* d_fbchain;
    bool layers_ok;                  // L == 1 and the graph is a gh_layerform (layers or loop: h_layers.loop)
    gh_layerform h_layers;
    gh_layerform* d_layers;
    int32_t* d_lf_end_slot;          // [R] position of a row in the end list or -1
    bool seq_ok;                     // every graph is in sequence form with the same N
    int seq_N, seq_skip;
    std::vector<gh_seqgraph> h_seqgraphs;
    std::vector<gh_seqword> h_seqwords;
    gh_seqgraph* d_seqgraphs;
    gh_seqword* d_seqwords;
    int32_t* d_seq_end_slot;         // [Rtot] position of a row in its graph's end list or -1
    int beam;           // rank beam per column of gh_viterbi (0 = off): gh_lattices_set_beam; generic kernel only
    bool has_nan_arc;   // a NaN arc cost needs np.argmin's NaN-first rule: generic kernel only
    bool has_self_arc;  // a same-column self arc can raise the reference's NameError: generic kernel only
};

int gh_gmm_update_dev(gh_ctx* ctx, gh_gmm* g, const double* d_mean, const double* d_var, const double* d_weight, int* d_flag);
int gh_batch_ensure_nll(gh_ctx* ctx, gh_batch* b, int S, bool zero);
// compat bit 0, second half: +inf where EVERY component's exp(-q/2) or weighted density rounds to 0 although the largest
// total logarithm is still above ln 2^-1075 (components with w * norm > 1); a no-op launch for ordinary models
int gh_loglik_underflow_fix(gh_ctx* ctx, const gh_gmm* g, gh_batch* b);
struct gh_comm;
int gh_comm_allreduce_enqueue(gh_comm* c, double* dev, int64_t n);
gh_ctx* gh_comm_context(const gh_comm* c);
// hipStreamSynchronize of the context's stream -- behind a collective (c != NULL) with RCCL's asynchronous errors and the
// GMMHMM_COMM_TIMEOUT deadline: GH_ERR_COMM (communicator aborted) instead of a hang when a peer rank is gone
int gh_stream_wait(gh_ctx* ctx, gh_comm* c, const char* who);

// the handle whose row-per-lane arrays are valid: `l` itself, or the lazily expanded twin of a transcripts handle
int gh_lattices_full(const gh_lattices* l, const gh_lattices** out);

// kernels (gh_loglik.hip / gh_viterbi.hip)
int gh_launch_loglik(gh_ctx* ctx, const gh_gmm* g, gh_batch* b);
// 1 = shape not covered; st_lo / st_hi (per utterance, or null): only the states [lo, hi) of every utterance are needed
// gh_bw_fused.hip: 1 = shapes not covered
int gh_bw_accumulate_fused(gh_ctx* ctx, const gh_gmm* g, const gh_batch* b, double occ_floor, double* stats_dev, double** d_result,
                           bool seq = false);
int gh_bw_expand_gamma(gh_ctx* ctx, gh_batch* b, int S);
// work lists of the fused statistics kernel on the device: built per call (context scratch) or once per trainer (own arena)
struct gh_bwf_plan {
    int KS, lt, S, M, D, L, n_wgs, n_pairs, tile_len, max_pairs;
    int slot_shift;            // a workgroup's waves: 1 << slot_shift >= the most column groups any word has
    int32_t *d_ulist, *d_seglen;
    int64_t* d_segfirst;
    void *d_wgs, *d_pairs;
    gh_fbchain* d_chains;
    double *d_part, *d_gsum, *d_own;
    void* d_arena;
};
int gh_bwf_plan_build(gh_ctx* ctx, int S, int M, int D, int KP, const std::vector<gh_fbchain>& chains,
                      const std::vector<int64_t>& seg_first, const std::vector<int32_t>& seg_len,
                      const std::vector<std::vector<int32_t>>& by_graph, bool persistent, gh_bwf_plan* out);   // 1 = shapes not covered
void gh_bwf_plan_free(gh_bwf_plan* p);
int gh_bwf_launch(gh_ctx* ctx, const gh_bwf_plan& pl, const gh_gmm* g, const double* feats, const double* gam, int gam_stride,
                  int gam_by_state, double occ_floor, const gh_fbchain* d_chains, double* d_out,
                  const double* nll = nullptr, int nll_S = 0,   // nll: the batch's likelihoods under g (needed when M > 8)
                  const int32_t* rng = nullptr);               // [U, GH_FBCHAIN_MAX, 2] occupancy ranges of fb_chain_kernel (same floor)
// occupancy ranges of a sequence-form forward-backward (seg_lo / seg_hi [U, GH_SEQ_MAXK]) -> rng [slots, GH_FBCHAIN_MAX, 2]
int gh_bwf_seq_ranges_launch(gh_ctx* ctx, const struct gh_seqgraph* graphs, const int32_t* utt_graph, const int64_t* slot_off,
                             const int32_t* seg_lo, const int32_t* seg_hi, const int32_t* row_lo /*or null*/,
                             const int32_t* row_hi, int64_t U, int n, int32_t* rng);
// lanes per utterance of the chain forward-backward (= columns of the compact gamma matrix): 8, or 16 when a chain is longer
inline int gh_fbchain_lanes(const std::vector<gh_fbchain>& chains) {
    for (const gh_fbchain& c : chains) if (c.n > 8) return 16;
    return 8;
}
// persistent block table of the subset likelihood kernel (built once by a trainer whose transcripts never change)
struct gh_loglik_plan { void* d_blk; int64_t n_blk; int max_tiles; };
int gh_loglik_plan_build(gh_ctx* ctx, const gh_gmm* g, const gh_batch* b, const int32_t* st_lo, const int32_t* st_hi,
                         const int64_t* rng_off, gh_loglik_plan* out, const uint8_t* include = nullptr);   // 1 = not covered; include [U]: part of the batch
void gh_loglik_plan_free(gh_loglik_plan* p);
int gh_launch_loglik_mfma(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int32_t* st_lo = nullptr,
                          const int32_t* st_hi = nullptr, const int64_t* rng_off = nullptr,   // rng_off: several ranges per utterance
                          const gh_loglik_plan* plan = nullptr);
