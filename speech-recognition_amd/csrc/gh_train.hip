// Training-side kernels (fp64):
//   * k-means assignment  (reference: inner N x k loop of kmeans, kmeans.py:180-186)
//   * mixture-EM E-step sufficient statistics (reference: GMM.em, hmm_state.py:127-143)
#include "gh_internal.h"
#include <cstring>
#include <cmath>

namespace {

// ---------------------------------------------------------------- k-means assign
// One frame per lane; centroids (and the shared variance) are staged in LDS and read
// by broadcast.  dist = 0.5*log((2pi)^D prod var) + 0.5*sum (c-x)/var*(c-x)
// (mahalanobis(centroid, x, cov[0]), kmeans.py:183) or ||c-x|| (default dist_fun);
// np.argmin => first minimum.
__global__ void kmeans_assign_kernel(const double* __restrict__ X, int64_t N, int D, int k,
                                     const double* __restrict__ cent, const double* __restrict__ var,
                                     double logdet, int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* sc = sm;           // [k,D]
    double* sv = sm + k * D;   // [D]
    for (int i = threadIdx.x; i < k * D; i += blockDim.x) sc[i] = cent[i];
    if (var) for (int i = threadIdx.x; i < D; i += blockDim.x) sv[i] = var[i];
    __syncthreads();
    const int64_t nidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nidx >= N) return;
    const double* x = X + nidx * D;
    double best = 0;
    int bi = 0;
    for (int c = 0; c < k; ++c) {
        double q = 0, dist;
        if (var) {
            for (int d = 0; d < D; ++d) { const double t = sc[c * D + d] - x[d]; q += t / sv[d] * t; }
            dist = logdet + 0.5 * q;
        } else {
            for (int d = 0; d < D; ++d) { const double t = sc[c * D + d] - x[d]; q = fma(t, t, q); }
            dist = sqrt(q);
        }
        if (c == 0 || dist < best || (dist != dist && best == best)) { best = dist; bi = c; }  // np.argmin
    }
    out[nidx] = bi;
}

// ------------------------------------------------------------------ EM statistics
// Tile of F frames per iteration.  Phase 1 (lane = frame): weighted component log
// densities of the first k components, responsibilities r = softmax over those k
// (hmm_state.py:128-133; components the reference's linear-domain product rounds to 0 get none) into LDS.  Phase 2 (lane = (component, dim) pair):
// S0[c] += r, S1[c,d] += r (x_d - mean_cd), S2[c,d] += r (x_d - mean_cd)^2 from the LDS tile
// -- no shuffles, no atomics.  Centring on the component's current mean keeps the
// single-pass variance S2/S0 - (S1/S0)^2 free of cancellation (the reference makes a
// second pass over the data around the new mean, hmm_state.py:141-143).  Each workgroup writes one partial [k, 1+2D]; the host adds the partials in
// a fixed order (deterministic).
constexpr int EM_MAXP = 8;  // pairs per lane: k*(D+1) <= 8*256
constexpr double EM_LN_UNDERFLOW = -745.1332191019412;  // exp(x) rounds to +0 in fp64 below this

__global__ __launch_bounds__(256) void em_stats_kernel(const double* __restrict__ X, int64_t N, int D, int k,
                                                       const double* __restrict__ mean,
                                                       const double* __restrict__ ivar,
                                                       const double* __restrict__ logc, int F,
                                                       double* __restrict__ partial, double* __restrict__ ll_partial) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* xt = sm;               // [F,D]
    double* rt = sm + F * D;       // [k,F]
    double* pm = rt + k * F;       // [k,D] mean
    double* pv = pm + k * D;       // [k,D] inverse variance
    double* pc = pv + k * D;       // [k]
    const int tid = threadIdx.x;
    for (int i = tid; i < k * D; i += 256) { pm[i] = mean[i]; pv[i] = ivar[i]; }
    for (int i = tid; i < k; i += 256) pc[i] = logc[i];
    const int P = k * (D + 1);
    double s1[EM_MAXP], s2[EM_MAXP];
#pragma unroll
    for (int q = 0; q < EM_MAXP; ++q) { s1[q] = 0; s2[q] = 0; }
    double ll_sum = 0;
    const int64_t ntiles = (N + F - 1) / F;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * F;
        const int nf = (int)((N - n0 < F) ? (N - n0) : F);
        __syncthreads();
        for (int i = tid; i < nf * D; i += 256) xt[i] = X[n0 * D + i];
        __syncthreads();
        for (int f = tid; f < F; f += 256) {
            if (f < nf) {
                const double* x = xt + f * D;
                double mx = -INFINITY;
                bool bad = false;  // a NaN density makes the whole row NaN (p /= NaN row sum, hmm_state.py:130-133)
                for (int c = 0; c < k; ++c) {
                    double q = 0;
                    for (int d = 0; d < D; ++d) { const double t = x[d] - pm[c * D + d]; q = fma(t * pv[c * D + d], t, q); }
                    double ll = pc[c] - 0.5 * q;
                    bad |= (ll != ll);
                    // The reference forms w * norm * np.exp(-0.5 q) in the LINEAR domain (hmm_state.py:42-43,115): once
                    // exp() -- or the product -- rounds to 0 the component gets no share of the frame, and a frame whose
                    // every component underflowed keeps an all-zero row (row sum 0 -> 1e-5, :130-133): it moves nothing.
                    if (-0.5 * q < EM_LN_UNDERFLOW || ll < EM_LN_UNDERFLOW) ll = -INFINITY;
                    rt[c * F + f] = ll;
                    mx = fmax(mx, ll);
                }
                double sum = 0;
                for (int c = 0; c < k; ++c) {
                    const double e = (mx == -INFINITY) ? 0.0 : exp(rt[c * F + f] - mx);
                    rt[c * F + f] = e;
                    sum += e;
                }
                const double inv = bad ? NAN : (sum > 0 ? 1.0 / sum : 0.0);
                for (int c = 0; c < k; ++c) rt[c * F + f] = bad ? NAN : rt[c * F + f] * inv;
                if (bad) ll_sum = NAN;
                else if (sum > 0) ll_sum += mx + log(sum);
            } else {
                for (int c = 0; c < k; ++c) rt[c * F + f] = 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EM_MAXP; ++q) {
            const int p = tid + q * 256;
            if (p < P) {
                const int c = p / (D + 1), d = p % (D + 1);
                const double* r = rt + c * F;
                double a1 = 0, a2 = 0;
                if (d == D) {
                    for (int f = 0; f < nf; ++f) a1 += r[f];
                } else {
                    for (int f = 0; f < nf; ++f) {
                        const double xv = xt[f * D + d] - pm[c * D + d];  // centred on the current mean
                        const double rx = r[f] * xv;
                        a1 += rx;
                        a2 = fma(rx, xv, a2);
                    }
                }
                s1[q] += a1;
                s2[q] += a2;
            }
        }
    }
    double* out = partial + (int64_t)blockIdx.x * k * (1 + 2 * D);
#pragma unroll
    for (int q = 0; q < EM_MAXP; ++q) {
        const int p = tid + q * 256;
        if (p < P) {
            const int c = p / (D + 1), d = p % (D + 1);
            double* o = out + c * (1 + 2 * D);
            if (d == D) o[0] = s1[q];
            else { o[1 + d] = s1[q]; o[1 + D + d] = s2[q]; }
        }
    }
    // per-block sum of frame log-likelihoods (over the first k components)
    __shared__ double red[256];
    red[tid] = ll_sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) ll_partial[blockIdx.x] = red[0];
}


// ------------------------------------------------- Baum-Welch (soft) statistics
// Every frame contributes to every state s with weight occ[n,s] (the forward-backward state
// posterior): r_nsm = occ[n,s] * w_sm pdf_sm(x_n) / sum_m' w_sm' pdf_sm'(x_n).  Posteriors are
// sparse (mass sits on the states near the alignment), so a tile of F frames first votes which
// states are active at all and only those are evaluated.  Accumulation is deterministic: each
// workgroup owns a private [S,M,1+2D] slab in HBM (L2 resident), a second kernel adds the slabs
// in a fixed order -- no float atomics.  Statistics are centred on the current means.
constexpr int BW_MAXP = 8;  // (component, dimension) pairs per lane of bw_stats_kernel: M*(D+1) <= 8*256

// workgroup barrier that orders LDS traffic only (no wait for outstanding global loads / store acknowledgements)
__device__ __forceinline__ void bw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MF = accumulate on the matrix cores (needs M <= 8 and 2D + 1 <= 128): see the accumulation phase below.
// MAXCT / MAXP bound the column tiles and the (component, dimension) pairs per lane at compile time (register arrays).
template <bool MF, int MAXCT, int MAXP>
__global__ __launch_bounds__(256) void bw_stats_kernel(const double* __restrict__ X, int64_t N, int D, int S, int M,
                                                       const double* __restrict__ mean, const double* __restrict__ ivar,
                                                       const double* __restrict__ logc, const double* __restrict__ occ,
                                                       double occ_floor, int F, const int64_t* __restrict__ utt_off,
                                                       int64_t U, double* __restrict__ slabs,
                                                       const int32_t* __restrict__ utt_states) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* xt = sm;                 // [F,D]   frames of the tile (MF: rows >= nf zeroed up to a multiple of 4)
    double* rt = xt + F * D;         // [M,F]   weighted responsibilities of the current state (MF: [F,8], frame major)
    double* pm = rt + (MF ? 8 : M) * F;  // [M,D] mean      } of the current state,
    double* pv = pm + M * D;         // [M,D]   1/variance  } staged once per (tile, state)
    double* pc = pv + M * D;         // [M]     log-constant
    double* wt = pc + M;             // [F]     occupancy of the current state on the tile's frames
    const int NCT = (2 * D + 1 + 15) / 16;                    // MF: 16-column tiles of Z = [1 | x - c | (x - c)^2]
    double* gt = wt + F;                                       // MF: [8][NCT*16] product tile
    int* s_list = reinterpret_cast<int*>(gt + (MF ? 8 * NCT * 16 : 0));  // [S] active states of the tile, compacted
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int W = 1 + 2 * D;
    const int RS = MF ? 8 : 0;                                 // MF: row stride of rt
    double* slab = slabs + (int64_t)blockIdx.x * S * M * W;
    // tiles never straddle utterances: an utterance only occupies the states of its own graph, so a
    // tile's active set stays small (5 states for an isolated word instead of the 10-15 of a 128-frame
    // window over 2-3 utterances)
    for (int64_t u = blockIdx.x; u < U; u += gridDim.x)
    for (int64_t n0 = utt_off[u]; n0 < utt_off[u + 1]; n0 += F) {
        const int nf = (int)((utt_off[u + 1] - n0 < F) ? (utt_off[u + 1] - n0) : F);
        __syncthreads();
#pragma unroll 8
        for (int i = tid; i < nf * D; i += 256) xt[i] = X[n0 * D + i];   // (unrolled: 8 loads in flight per lane)
        if (MF) for (int i = nf * D + tid; i < ((nf + 3) & ~3) * D; i += 256) xt[i] = 0.0;   // 0 * garbage could be NaN
        // which states have any occupancy on this tile?  All lanes sweep the [nf, S] block of the occupancy
        // matrix linearly (coalesced, 4 loads in flight per lane) and raise a flag per state -- one lane per
        // state walking down its column costs one memory round trip per frame.
        if (utt_states) {   // a chain-form forward-backward left the list of states that can be occupied (<= 8)
            __syncthreads();
            if (tid == 0) {
                int c = 0;
                for (int j = 0; j < GH_FBCHAIN_MAX; ++j) { const int s = utt_states[u * GH_FBCHAIN_MAX + j]; if (s >= 0) s_list[c++] = s; }
                s_count = c;
            }
            __syncthreads();
        } else {
        for (int s = tid; s < S; s += 256) s_list[s] = 0;
        __syncthreads();
        {
            const double* ob = occ + n0 * S;
            const int total = nf * S;
#pragma unroll 4
            for (int i = tid; i < total; i += 256) {
                const double w = ob[i];
                if ((w > occ_floor) || (w != w)) s_list[i % S] = 1;   // benign race: every writer stores 1
            }
        }
        __syncthreads();
        if (tid == 0) {
            int c = 0;
            for (int s = 0; s < S; ++s)
                if (s_list[s]) s_list[c++] = s;
            s_count = c;
        }
        __syncthreads();
        }
        const int count = s_count;
        for (int k = 0; k < count; ++k) {
            const int s = s_list[k];
            // this lane's slab entries ((component, dimension) pairs p = tid and tid + 256) start their trip from
            // L2 / HBM now and are added to at the end of the pair: the read-modify-write latency of the 253 KB
            // per-workgroup slab was the largest single cost of the kernel (33 k of 80 k cycles per pair)
            const int P = M * (D + 1);
            double so1[MAXP], so2[MAXP];
#pragma unroll
            for (int h = 0; h < MAXP; ++h) {
                so1[h] = 0; so2[h] = 0;
                const int p = tid + 256 * h;
                if (p < P) {
                    const int m = p / (D + 1), d = p % (D + 1);
                    const double* o = slab + ((int64_t)s * M + m) * W;
                    so1[h] = (d == D) ? o[0] : o[1 + d];
                    so2[h] = (d == D) ? 0.0 : o[1 + D + d];
                }
            }
            for (int i = tid; i < M * D; i += 256) {
                pm[i] = mean[(int64_t)s * M * D + i];
                pv[i] = ivar[(int64_t)s * M * D + i];
            }
            if (tid < M) pc[tid] = logc[(int64_t)s * M + tid];
            if (MF) for (int i = tid; i < 8 * NCT * 16; i += 256) gt[i] = 0.0;
            bw_lds_barrier();
            // component log-densities of every (frame, component) pair of the tile on all 256 lanes (the lanes of a
            // frame read the same feature row: LDS broadcast), four independent partial sums per pair; then one
            // lane per frame normalises over the components
            for (int f = tid; f < F; f += 256) wt[f] = (f < nf) ? occ[(n0 + f) * S + s] : 0.0;
            bw_lds_barrier();
            if (MF) {
                // densities AND the normalisation over the components in one phase: lane = (frame, component slot of
                // 8); max / sum over the 8 slots by xor shuffles -- no intermediate LDS round trip, no second barrier
                const int npair = ((nf + 3) & ~3) * 8;
                for (int p = tid; p < npair; p += 256) {
                    const int f = p >> 3, m = p & 7;
                    const double wgt = (f < nf) ? wt[f] : 0.0;
                    const bool on = (wgt > occ_floor || wgt != wgt);
                    double ll = -INFINITY;
                    if (on && m < M) {
                        const double* x = xt + f * D;
                        const double* mu = pm + m * D;
                        const double* iv = pv + m * D;
                        double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                        int d = 0;
                        for (; d + 3 < D; d += 4) {
                            const double t0 = x[d] - mu[d], t1 = x[d + 1] - mu[d + 1], t2 = x[d + 2] - mu[d + 2], t3 = x[d + 3] - mu[d + 3];
                            q0 = fma(t0 * iv[d], t0, q0); q1 = fma(t1 * iv[d + 1], t1, q1);
                            q2 = fma(t2 * iv[d + 2], t2, q2); q3 = fma(t3 * iv[d + 3], t3, q3);
                        }
                        for (; d < D; ++d) { const double t = x[d] - mu[d]; q0 = fma(t * iv[d], t, q0); }
                        ll = pc[m] - 0.5 * ((q0 + q1) + (q2 + q3));
                    }
                    // NaN anywhere in the frame's densities poisons the frame (as the sequential version did)
                    double bad = (ll != ll) ? 1.0 : 0.0, mx = (ll != ll) ? -INFINITY : ll;
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) { mx = fmax(mx, __shfl_xor(mx, o)); bad += __shfl_xor(bad, o); }
                    double e = (mx == -INFINITY || m >= M || ll != ll) ? 0.0 : exp(ll - mx);
                    double sum = e;
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
                    double r = 0.0;
                    if (on && m < M) r = (bad > 0.0) ? NAN : (sum > 0 ? e * (wgt / sum) : 0.0);
                    rt[f * RS + m] = r;
                }
            } else {
            for (int p = tid; p < nf * M; p += 256) {
                const int f = p / M, m = p - f * M;
                const double wgt = wt[f];
                double ll = 0.0;
                if (wgt > occ_floor || wgt != wgt) {
                    const double* x = xt + f * D;
                    const double* mu = pm + m * D;
                    const double* iv = pv + m * D;
                    double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                    int d = 0;
                    for (; d + 3 < D; d += 4) {
                        const double t0 = x[d] - mu[d], t1 = x[d + 1] - mu[d + 1], t2 = x[d + 2] - mu[d + 2], t3 = x[d + 3] - mu[d + 3];
                        q0 = fma(t0 * iv[d], t0, q0); q1 = fma(t1 * iv[d + 1], t1, q1);
                        q2 = fma(t2 * iv[d + 2], t2, q2); q3 = fma(t3 * iv[d + 3], t3, q3);
                    }
                    for (; d < D; ++d) { const double t = x[d] - mu[d]; q0 = fma(t * iv[d], t, q0); }
                    ll = pc[m] - 0.5 * ((q0 + q1) + (q2 + q3));
                }
                rt[MF ? f * RS + m : m * F + f] = ll;
            }
            bw_lds_barrier();
            if (tid < F) {
                const int f = tid;
                const double wgt = wt[f];
                double* rf = MF ? rt + f * RS : rt + f;          // this frame's responsibilities, stride rs
                const int rs = MF ? 1 : F;
                if (f < nf && (wgt > occ_floor || wgt != wgt)) {
                    double mx = -INFINITY;
                    bool bad = false;
                    for (int m = 0; m < M; ++m) {
                        const double ll = rf[m * rs];
                        bad |= (ll != ll);
                        mx = fmax(mx, ll);
                    }
                    double sum = 0;
                    for (int m = 0; m < M; ++m) {
                        const double e = (mx == -INFINITY) ? 0.0 : exp(rf[m * rs] - mx);
                        rf[m * rs] = e;
                        sum += e;
                    }
                    const double inv = bad ? NAN : (sum > 0 ? wgt / sum : 0.0);
                    for (int m = 0; m < M; ++m) rf[m * rs] = bad ? NAN : rf[m * rs] * inv;
                    if (MF) for (int m = M; m < 8; ++m) rf[m] = 0.0;
                } else if (MF) {
                    if (f < ((nf + 3) & ~3)) for (int m = 0; m < 8; ++m) rf[m] = 0.0;
                } else {
                    for (int m = 0; m < M; ++m) rf[m * rs] = 0.0;
                }
            }
            }   // !MF
            bw_lds_barrier();
            if (MF) {
                // ---- accumulation on the matrix cores: G[m, c] = sum_f r[f, m] Z[f, c] with ONE operand for all
                // components, Z[f] = [1 | x_f - c_s | (x_f - c_s)^2] centred on the state's first component mean
                // c_s; the component-centred sums follow in the epilogue: with delta = mean_m - c_s,
                //   S1 = G1 - delta S0,   S2 = G2 - 2 delta G1 + delta^2 S0.
                // v_mfma_f64_16x16x4: A[i, k] = r[f0 + k][i] (rows 8..15 zero), B[k, j] = Z[f0 + k][16 ct + j]; wave w
                // owns column tiles w, w + 4.  (The per-lane frame loops of the vector version were 33 k cycles per
                // tile and state.)
                // Work split: wave w takes the k-steps (groups of 4 frames) w, w + 4, ... for ALL column tiles -- one A
                // fragment feeds NCT independent MFMAs, whose issue time hides the LDS latency of the next operands --
                // and adds its partial tile into the shared product tile with LDS fp64 atomics.
                typedef double v4d __attribute__((ext_vector_type(4)));
                const int lane = tid & 63, wv = tid >> 6;
                const int i = lane & 15, kq = lane >> 4;
                const int nk = (nf + 3) >> 2;
                int kind[MAXCT], dd[MAXCT];
                double cs[MAXCT];
                v4d acc[MAXCT];
#pragma unroll
                for (int ct = 0; ct < MAXCT; ++ct) {
                    const int c = ct * 16 + i;                       // column of Z this lane feeds in tile ct
                    kind[ct] = (ct >= NCT) ? 3 : (c == 0) ? 0 : (c <= D) ? 1 : (c <= 2 * D) ? 2 : 3;
                    dd[ct] = (kind[ct] == 1) ? c - 1 : (kind[ct] == 2) ? c - 1 - D : 0;
                    cs[ct] = pm[dd[ct]];
                    acc[ct] = (v4d){0, 0, 0, 0};
                }
                for (int k = wv; k < nk; k += 4) {
                    const int f = 4 * k + kq;
                    const double av = (i < 8) ? rt[f * RS + i] : 0.0;
                    const double* xr = xt + f * D;
#pragma unroll
                    for (int ct = 0; ct < MAXCT; ++ct) {
                        if (ct < NCT) {
                            const double xv = xr[dd[ct]] - cs[ct];
                            const double bv = (kind[ct] == 0) ? 1.0 : (kind[ct] == 1) ? xv : (kind[ct] == 2) ? xv * xv : 0.0;
                            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[ct], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < MAXCT; ++ct) {
                    if (ct < NCT) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {             // D[row = kq + 4 r4][col = i]
                            const int row = kq + 4 * r4;
                            if (row < 8) atomicAdd(&gt[row * (NCT * 16) + ct * 16 + i], acc[ct][r4]);
                        }
                    }
                }
                bw_lds_barrier();
#pragma unroll
                for (int h = 0; h < MAXP; ++h) {
                    const int p = tid + 256 * h;
                    if (p >= P) continue;
                    const int m = p / (D + 1), d = p % (D + 1);
                    const double* g = gt + m * (NCT * 16);
                    double* o = slab + ((int64_t)s * M + m) * W;
                    if (d == D) {
                        o[0] = so1[h] + g[0];
                    } else {
                        const double dl = pm[m * D + d] - pm[d], s0 = g[0], g1 = g[1 + d], g2 = g[1 + D + d];
                        o[1 + d] = so1[h] + (g1 - dl * s0);
                        o[1 + D + d] = so2[h] + (g2 - dl * (2.0 * g1 - dl * s0));
                    }
                }
                bw_lds_barrier();
                continue;
            }
#pragma unroll
            for (int h = 0; h < MAXP; ++h) {
                const int p = tid + 256 * h;
                if (p >= P) continue;
                const int m = p / (D + 1), d = p % (D + 1);
                const double* r = rt + m * F;
                double a1 = 0, a2 = 0;
                if (d == D) {
                    for (int f = 0; f < nf; ++f) a1 += r[f];
                } else {
                    const double mu = pm[m * D + d];
                    for (int f = 0; f < nf; ++f) {
                        const double xv = xt[f * D + d] - mu;
                        const double rx = r[f] * xv;
                        a1 += rx;
                        a2 = fma(rx, xv, a2);
                    }
                }
                double* o = slab + ((int64_t)s * M + m) * W;
                if (d == D) o[0] = so1[h] + a1;
                else { o[1 + d] = so1[h] + a1; o[1 + D + d] = so2[h] + a2; }
            }
            bw_lds_barrier();  // rt / pm are rewritten by the next state (the slab stores stay in flight)
        }
    }
}

__global__ void slab_reduce_kernel(const double* __restrict__ slabs, int n_slabs, int64_t len, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double acc = 0;
    for (int g = 0; g < n_slabs; ++g) acc += slabs[(int64_t)g * len + i];
    out[i] = acc;
}

}  // namespace

static int frames_f64(const gh_batch* b, int64_t first, int64_t count, const char* who) {
    GH_REQUIRE(b->dtype == GH_F64, "%s: needs an fp64 batch", who);
    GH_REQUIRE(first >= 0 && count >= 0 && first + count <= b->N, "%s: frame range [%lld,+%lld) outside the batch",
               who, (long long)first, (long long)count);
    return GH_OK;
}

extern "C" int gh_kmeans_assign(gh_ctx* ctx, const gh_batch* b, int64_t first, int64_t count, int k,
                                const double* centroids, const double* var, int32_t* out_clusters) {
    GH_REQUIRE(ctx && b && centroids && out_clusters, "gh_kmeans_assign: NULL argument");
    GH_REQUIRE(k > 0, "gh_kmeans_assign: k=%d", k);
    int rc = frames_f64(b, first, count, "gh_kmeans_assign");
    if (rc) return rc;
    if (count == 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    const int D = b->D;
    double logdet = 0;
    if (var) {
        double prod = 1.0;
        for (int d = 0; d < D; ++d) prod *= var[d];
        logdet = 0.5 * std::log(std::pow(2.0 * M_PI, D) * prod);  // hmm_state.py:58
    }
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t bc = al((size_t)k * D * 8), bv = al((size_t)D * 8);
    void* base;
    rc = gh_scratch(ctx, bc + bv + (size_t)count * 4, &base);
    if (rc) return rc;
    double* dc = (double*)base;
    double* dv = (double*)((char*)base + bc);
    int32_t* dout = (int32_t*)((char*)base + bc + bv);
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(dc, centroids, (size_t)k * D * 8, hipMemcpyHostToDevice, st));
    if (var) GH_HIP(hipMemcpyAsync(dv, var, (size_t)D * 8, hipMemcpyHostToDevice, st));
    const size_t lds = ((size_t)k * D + D) * 8;
    GH_REQUIRE(lds <= 64 * 1024, "gh_kmeans_assign: k*D too large for LDS staging");
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), lds, st,
                       (const double*)b->feats + first * D, count, D, k, dc, var ? dv : nullptr, logdet, dout);
    GH_HIP(hipGetLastError());
    GH_HIP(hipMemcpyAsync(out_clusters, dout, (size_t)count * 4, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

extern "C" int gh_em_accumulate(gh_ctx* ctx, const gh_batch* b, int64_t first, int64_t count, int k,
                                const double* mean, const double* var, const double* weight,
                                double* out_stats, double* out_loglik) {
    GH_REQUIRE(ctx && b && mean && var && weight && out_stats, "gh_em_accumulate: NULL argument");
    int rc = frames_f64(b, first, count, "gh_em_accumulate");
    if (rc) return rc;
    const int D = b->D;
    GH_REQUIRE(k > 0 && k * (D + 1) <= EM_MAXP * 256, "gh_em_accumulate: k=%d x D=%d unsupported", k, D);
    const int W = 1 + 2 * D;
    for (int i = 0; i < k * W; ++i) out_stats[i] = 0.0;
    if (out_loglik) *out_loglik = 0.0;
    if (count == 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    std::vector<double> ivar((size_t)k * D), logc(k);
    const double log2pi = std::log(2.0 * M_PI);
    for (int c = 0; c < k; ++c) {
        double sl = 0;
        for (int d = 0; d < D; ++d) {
            const double v = var[(size_t)c * D + d];
            GH_REQUIRE(v != 0, "gh_em_accumulate: var[%d,%d]=%g (singular covariance)", c, d, v);
            ivar[(size_t)c * D + d] = 1.0 / v;
            sl += std::log(v);
        }
        logc[c] = std::log(weight[c]) - 0.5 * (D * log2pi + sl);
    }
    int F = 256;
    auto lds_need = [&](int f) { return ((size_t)f * D + (size_t)k * f + 2 * (size_t)k * D + k) * 8; };
    while (F > 32 && lds_need(F) > 96 * 1024) F >>= 1;
    GH_REQUIRE(lds_need(F) <= 150 * 1024, "gh_em_accumulate: D=%d k=%d does not fit LDS", D, k);
    const int64_t ntiles = (count + F - 1) / F;
    const int grid = (int)std::min<int64_t>(ntiles, 2 * (int64_t)ctx->n_cu);
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t bm = al((size_t)k * D * 8), bl = al((size_t)k * 8), bp = al((size_t)grid * k * W * 8),
                 bll = al((size_t)grid * 8);
    void* base;
    rc = gh_scratch(ctx, 2 * bm + bl + bp + bll, &base);
    if (rc) return rc;
    char* p = (char*)base;
    double *dm = (double*)p, *div = (double*)(p + bm), *dl = (double*)(p + 2 * bm), *dpart = (double*)(p + 2 * bm + bl),
           *dll = (double*)(p + 2 * bm + bl + bp);
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(dm, mean, (size_t)k * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(div, ivar.data(), (size_t)k * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(dl, logc.data(), (size_t)k * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(em_stats_kernel, dim3(grid), dim3(256), lds_need(F), st, (const double*)b->feats + first * D,
                       count, D, k, dm, div, dl, F, dpart, dll);
    GH_HIP(hipGetLastError());
    std::vector<double> part((size_t)grid * k * W), llp(grid);
    GH_HIP(hipMemcpyAsync(part.data(), dpart, part.size() * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(llp.data(), dll, llp.size() * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    double ll = 0;
    for (int g = 0; g < grid; ++g) {
        for (int i = 0; i < k * W; ++i) out_stats[i] += part[(size_t)g * k * W + i];
        ll += llp[g];
    }
    if (out_loglik) *out_loglik = ll;
    return GH_OK;
}

extern "C" int gh_bw_accumulate(gh_ctx* ctx, const gh_gmm* g, const gh_batch* b, double occ_floor,
                                double* out_stats, double* stats_dev) {
    GH_REQUIRE(ctx && g && b && (out_stats || stats_dev), "gh_bw_accumulate: NULL argument");
    GH_REQUIRE(b->dtype == GH_F64, "gh_bw_accumulate: needs an fp64 batch");
    GH_REQUIRE(g->D == b->D && g->S == b->nll_S, "gh_bw_accumulate: model / batch mismatch");
    GH_HIP(hipSetDevice(ctx->device));
    const int S = g->S, M = g->M, D = g->D, W = 1 + 2 * D;
    const int64_t len = (int64_t)S * M * W;
    hipStream_t st = ctx->stream;
    if (!b->gam_chains.empty() && !b->occ_valid && b->N > 0) {
        // the forward-backward left a compact gamma (chain graphs): densities + accumulation fused on the matrix cores
        double* d_res = nullptr;
        int rc0 = gh_bw_accumulate_fused(ctx, g, b, occ_floor, stats_dev, &d_res);
        if (rc0 < 0) return rc0;
        if (rc0 == 0) {
            if (out_stats) GH_HIP(hipMemcpyAsync(out_stats, d_res, (size_t)len * 8, hipMemcpyDeviceToHost, st));
            GH_HIP(hipStreamSynchronize(st));
            return GH_OK;
        }
        rc0 = gh_bw_expand_gamma(ctx, const_cast<gh_batch*>(b), S);   // shapes it does not cover: generic kernel below
        if (rc0) return rc0;
    }
    if (b->seq_seg_valid && b->occ && b->occ_valid && b->N > 0) {
        // the forward-backward ran in sequence form (multi-word transcripts): the same fused kernel over (utterance,
        // layer) segments grouped by word, gamma read from the occupancy matrix (GMMHMM_BW=generic: the generic kernel)
        const char* e2 = getenv("GMMHMM_BW");
        if (!(e2 && !strcmp(e2, "generic"))) {
            double* d_res = nullptr;
            const int rc0 = gh_bw_accumulate_fused(ctx, g, b, occ_floor, stats_dev, &d_res, true);
            if (rc0 < 0) return rc0;
            if (rc0 == 0) {
                if (out_stats) GH_HIP(hipMemcpyAsync(out_stats, d_res, (size_t)len * 8, hipMemcpyDeviceToHost, st));
                GH_HIP(hipStreamSynchronize(st));
                return GH_OK;
            }
        }
    }
    GH_REQUIRE((b->occ && b->occ_valid) || b->N == 0, "gh_bw_accumulate: run gh_forward_backward(want_occ=1) first");
    GH_REQUIRE(b->N == 0 || b->occ_S == g->S, "gh_bw_accumulate: occupancies were computed for %d states, the model has %d", b->occ_S, g->S);
    GH_REQUIRE(g->M * (g->D + 1) <= BW_MAXP * 256, "gh_bw_accumulate: M=%d x D=%d unsupported", g->M, g->D);
    // The kernel is latency bound per workgroup (global round trips between barriers), so workgroups per CU
    // decide: frame tiles are sized for as many resident workgroups as the 160 KB of LDS allow (a tile one
    // entry too large silently drops a CU from 3 to 2 workgroups: measured 4.4 -> 5.0 ms).
    const bool mf = M <= 8 && 2 * D + 1 <= 128;   // accumulation on the matrix cores
    const int nct = (2 * D + 1 + 15) / 16;
    auto lds_need = [&](int f) {
        return ((size_t)f * D + (size_t)(mf ? 8 : M) * f + 2 * (size_t)M * D + M + f + (mf ? 8 * nct * 16 : 0)) * 8 + (size_t)S * 4 + 16;
    };
    int F = (int)std::max<int64_t>(32, std::min<int64_t>(160, (b->max_T + 7) & ~int64_t(7)));  // one tile per utterance when it fits
    if (const char* e = getenv("GMMHMM_BW_F")) F = std::max(16, std::min(256, atoi(e)));   // tuning knob
    F = (F + 3) & ~3;
    while (F > 16 && lds_need(F) > 80 * 1024) F >>= 1;
    GH_REQUIRE(lds_need(F) <= 150 * 1024, "gh_bw_accumulate: D=%d M=%d does not fit LDS", D, M);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, ((size_t)160 * 1024) / (lds_need(F) + 256)));
    const int64_t ntiles = b->U;  // one utterance (in chunks of F frames) per workgroup pass
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, per_cu * (int64_t)ctx->n_cu));
    void* base;
    const size_t slab_bytes = (size_t)grid * len * 8, out_bytes = ((size_t)len * 8 + 255) & ~size_t(255);
    int rc = gh_scratch(ctx, out_bytes + slab_bytes, &base);
    if (rc) return rc;
    double* d_out = stats_dev ? stats_dev : (double*)base;
    double* d_slabs = (double*)((char*)base + out_bytes);
    GH_HIP(hipMemsetAsync(d_slabs, 0, slab_bytes, st));
    if (ntiles > 0) {
        auto kern = !mf ? bw_stats_kernel<false, 1, BW_MAXP>
                        : (nct <= 5 && M * (D + 1) <= 512) ? bw_stats_kernel<true, 5, 2> : bw_stats_kernel<true, 8, BW_MAXP>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_need(F), st, (const double*)b->feats, b->N, D, S, M,
                           g->dMean, g->dIvar, g->dLogc, b->occ, occ_floor, F, b->d_offsets, b->U, d_slabs, b->d_occ_states);
        GH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, d_slabs, grid, len, d_out);
    GH_HIP(hipGetLastError());
    if (out_stats) GH_HIP(hipMemcpyAsync(out_stats, d_out, (size_t)len * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}
