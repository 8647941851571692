// The two per-iteration kernels of the device-resident refit (gh_fit_kmeans / gh_fit_em, gh_lockstep.hip) as STREAMING
// matrix-core kernels: the E-step of the mixture EM (hmm_state.py:122-159: densities, responsibilities, weighted sums) and
// the assignment sweep of k-means (kmeans.py:180-186: distances, arg-min, cluster sums).
//
// Round 2-4 ran both as one workgroup per 64-frame tile, frame = lane, chains of dependent fp64 VALU operations with the
// parameters read from LDS by broadcast: 0.13 / 0.15 of the HBM rate for a pass over 1.4 M frames, 0.28 with the
// arithmetic compiled out (one tile of loads in flight per workgroup), and four to eight small launches around every pass.
// Here:
//   * a workgroup (4 waves) owns an ITEM of up to a few thousand consecutive frames of one state; a wave walks 16-frame
//     SLABS (wave, wave + 4, ...) through a private LDS buffer with the next slab's loads in flight in registers while
//     the current one is computed -- 16+ waves per CU, every one of them with 5 KB on its way;
//   * both contractions run on v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: the granularity of 4 components wastes
//     nothing at k = 4, 8, where the 16x16x4 form would pad k to 16; measured 73 TF against 77.8 for 16x16x4,
//     tools/mfma_f64_4x4.hip; fp64 MFMA and fp64 VALU share one pipe on gfx950, so the matrix form does not add flops,
//     it removes the operand traffic and the dependent chains):
//       phase 1   L[frame, comp] = sum_j  Z[frame, j] P[comp, j]       Z = [x^2 | x | 1]   (scores in the D layout:
//                 lane = 16 i + 4 b + j holds frame F(b, i), component 4 cg + j)
//       softmax / arg-min over the components of a frame: registers + two DPP quad steps
//       phase 2   S[col, comp]  += sum_frames Y[frame, col] R[frame, comp]    Y = [x | 1 | x^2]
//                 R = the phase-1 result registers themselves as the B operand (block b = the slab's frames
//                 F(b, 0..3)); a block accumulates its own four frames, so S is four partial sums per entry, kept in
//                 2 KS accumulators per component group for the whole item and added up when the item is written
//                 (CBSZ/ABID, which would broadcast one block's R to all four, is ignored by the fp64 form on gfx950:
//                 tools/mfma_f64_4x4.hip);
//   * frames and parameters are taken relative to a per-state point (`shift`), so the expanded quadratic form
//     x^2 A + x B + C loses nothing to cancellation; the update converts the sums to the reference's centred form;
//   * the item's sums go to a slab; the LAST workgroup of a state to arrive adds the slabs in a fixed order (deterministic),
//     runs GMM.em_update / the centroid update with its stop rule, and packs the operands of the next iteration:
//     ONE launch per lock-step iteration (three with a communicator: sums, collective, update).
// k-means keeps the reference's assignments: a frame whose two smallest distances are closer than the rounding of
// either computation (or that meets a non-finite number) is re-tested with the reference's own operations
// (t / var * t, summed in order, np.argmin's first minimum) -- see km_exact().
#include "gh_refit.h"
#include "gh_host.h"

namespace {

constexpr int RF_TAIL_POLLS = 1 << 22;                    // several seconds of waiting for a state's generation word, then an error bit
constexpr double RF_LN_UNDERFLOW = -745.1332191019412;   // exp(x) rounds to +0 below this (hmm_state.py's linear domain)

__device__ __forceinline__ double rf_mfma(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
template <int CTRL> __device__ __forceinline__ double rf_dpp(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ int rf_dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ double rf_vmax(double a, double b) {   // one instruction; a NaN operand loses
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 2^(y/128) for y <= 0 (or NaN), as in gh_loglik_mfma.hip: table of 2^(j/128) + degree-4 polynomial
__device__ __forceinline__ double rf_exp2s(double y, const double* __restrict__ tab) {
    const double n = __builtin_rint(y);
    const double r = y - n;
    const int ni = (int)n;
    const double t = tab[ni & 127];
    double p = fma(r, 3.583032305400251285e-11, 2.6466421444330968834e-08);
    p = fma(p, r, 1.4662262387640424337e-05);
    p = fma(p, r, 5.4152123481245727298e-03);
    p = p * r;
    return __builtin_ldexp(fma(t, p, t), ni >> 7);
}
__device__ __forceinline__ double rf_rcp(double s) {              // 1 / s for s in [1, 32] (or NaN)
    double y = __builtin_amdgcn_rcp(s);
    y = fma(fma(-s, y, 1.0), y, y);
    return fma(fma(-s, y, 1.0), y, y);
}

// frame of lane group (b, i) inside a slab: any bijection onto 0..15 serves both contractions; with this one the eight
// frames a half-wave reads in phase 2 (i = 0, 1 or i = 2, 3; four consecutive columns each) are the even (odd) rows,
// which a row stride = 2 mod 4 puts on eight disjoint groups of four banks; phase 1 reads two consecutive columns of all
// sixteen rows per half-wave, conflict-free for any stride whose half is odd
__device__ __forceinline__ int rf_frame(int b, int i) { return 2 * (4 * (i & 1) + b) + (i >> 1); }

// ------------------------------------------------------------------------------------------- slab staging
// A slab is 16 D consecutive doubles; lane l moves elements l, l + 64, ...: coalesced loads, all in flight at once,
// every one of them unconditional (branches around single loads and stores cost the kernel its pipelining: the first
// version had one per element), and the stores of elements beyond the slab go to a spare slot behind it.
template <int NIT>
struct rf_stage {
    int off[NIT];        // LDS slot of element lane + 64 it of a slab (the spare slot 16 TS: beyond the 16 frames)
    double mu[NIT];      // the shift of its dimension
    double r[NIT];       // the element on its way
    __device__ __forceinline__ void init(int lane, int D, int TS, const double* __restrict__ shift) {
        int f = lane / D, d = lane - f * D;
        const int q64 = 64 / D, r64 = 64 - q64 * D;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool in = lane + 64 * it < 16 * D;
            off[it] = in ? f * TS + d : 16 * TS;
            mu[it] = shift[in ? d : 0];
            d += r64;
            f += q64;
            if (d >= D) { d -= D; ++f; }
        }
    }
    // BUFFER loads: the descriptor's size is the slab's, an element beyond it reads as 0 (a frame beyond the item's end then
    // sits at -shift: finite, and its weight is 0) -- no clamp, no select, no branch
    __device__ __forceinline__ void load(int lane, const double* __restrict__ base, int nelem) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nelem * 8, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            r[it] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, it * 512, 0));
    }
    __device__ __forceinline__ void store(double* tb) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) tb[off[it]] = r[it] - mu[it];
    }
};

// constant column (feature D) = 1, padding columns = 0: written once, the staging never touches them
__device__ __forceinline__ void rf_init_pads(double* __restrict__ tb, int lane, int D, int TS) {
    const int f = lane & 15;
    for (int c = D + (lane >> 4); c < TS; c += 4) tb[f * TS + c] = (c == D) ? 1.0 : 0.0;
}

// An item's slab is stored WRITE-THROUGH (agent-scope relaxed atomic stores = global_store ... sc1): it needs no release
// fence -- the fence writes back every dirty line of the XCD's L2, 6.5 us per workgroup with a few KB freshly written and
// four times that with four workgroups per CU; it was most of the 35 us a workgroup spent outside its slabs.  Then:
// every wave drains its stores, barrier, one lane draws a ticket; the state's LAST workgroup acquires (invalidates its
// CU's L1) and reads the slabs with plain loads.  True in every thread of that workgroup.
__device__ __forceinline__ void rf_publish(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rf_publish(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rf_publish(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Everything a state's LAST workgroup writes (statistics, parameters, stop flags, the next operands) is stored write-through
// as well: in a TAIL launch the next iteration's workgroups -- on other CUs, other XCDs -- read it after one acquire.
__device__ __forceinline__ bool rf_arrive(int32_t* done, int s, int n_items, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(done + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == n_items - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(done + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// out[i] = sum over the slabs t = i0 .. i1 - 1 of partial[t][i], i < n, by the whole block in a FIXED order: wave w takes
// the slabs i0 + w, i0 + w + 4, ... with eight loads in flight per thread (one thread walking a column slab after slab
// paid one memory round trip per slab: 0.45 us each, 37 us more per launch with 110 slabs per state than with 28), and
// the four waves' sums are added in wave order through LDS
__device__ void rf_reduce_slabs(const double* __restrict__ partial, int plen, int i0, int i1, int n, double* __restrict__ out,
                                double* __restrict__ lds /*[RF_WAVES][64]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        double acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0;
        if (i < n) {
            const double* src = partial + i;
            int t = i0 + wv;
            for (; t + 7 * RF_WAVES < i1; t += 8 * RF_WAVES) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(t + j * RF_WAVES) * plen];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (t + j * RF_WAVES < i1) ? src[(int64_t)min(t + j * RF_WAVES, i1 - 1) * plen] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        lds[wv * 64 + lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (wv == 0 && i < n) rf_publish(out + i, (lds[lane] + lds[64 + lane]) + (lds[128 + lane] + lds[192 + lane]));
        __syncthreads();
    }
}

__device__ __forceinline__ bool rf_close(double a, double b) {   // np.isclose(a, b), default tolerances
    if (a == b) return true;
    if (!(a - a == 0.0) || !(b - b == 0.0)) return false;
    return fabs(a - b) <= 1e-8 + 1e-5 * fabs(b);
}

// =========================================================================================== mixture EM
// operands of one state for the next pass, by the whole block: P[((cg 2 + h) KS + st) 16 + 4 j + kk] = coefficient of
// component 4 cg + j at feature 4 st + kk, h = 0: of x^2 (-iv / 2), h = 1: of x (iv (mean - shift)), the constant
// log w - (D log 2 pi + sum log var + sum iv (mean - shift)^2) / 2 at feature D -- all times 128 / ln 2
__device__ void em_pack_state(const rf_em_args& a, int s, double* __restrict__ lds /*[2 k D + k]*/) {
    const int D = a.c.D, k = a.c.k, KS = rf_steps(D), CG = rf_comp_groups(k), tid = threadIdx.x;
    const double* mean = a.mean + (int64_t)s * k * D;
    const double* var = a.var + (int64_t)s * k * D;
    const double* sh = a.c.shift + (int64_t)s * D;
    double* cst = lds + 2 * k * D;
    for (int i = tid; i < k * D; i += blockDim.x) {             // the terms of every component's constant, side by side
        const int d = i % D;
        const double v = var[i], m = mean[i] - sh[d];
        if (v == 0) atomicOr(a.c.counter + 1, 16);              // np.linalg.inv raises LinAlgError (hmm_state.py:17)
        lds[i] = log(v);
        lds[k * D + i] = m * (1.0 / v) * m;
    }
    __syncthreads();
    if (tid < k) {
        const double log2pi = 1.8378770664093454836;
        double sl = 0, sq = 0;
        for (int d = 0; d < D; ++d) { sl += lds[tid * D + d]; sq += lds[k * D + tid * D + d]; }
        const double c = log(a.weight[(int64_t)s * k + tid]) - 0.5 * (D * log2pi + sl) - 0.5 * sq;
        cst[tid] = (c == -INFINITY) ? GH_LSE_OFF64 : (c != c ? c : fmax(c * GH_LSE_SCALE64, GH_LSE_OFF64));
    }
    __syncthreads();
    double* P = a.P + (int64_t)s * (CG * 2 * KS * 16);
    for (int i = tid; i < CG * 2 * KS * 16; i += blockDim.x) {
        const int kk = i & 3, j = (i >> 2) & 3, st = (i >> 4) % KS, h = ((i >> 4) / KS) & 1, cg = (i >> 4) / (2 * KS);
        const int c = 4 * cg + j, d = 4 * st + kk;
        double v = 0.0;
        if (c < k) {
            if (d < D) {
                const double iv = 1.0 / var[c * D + d];
                v = h == 0 ? -0.5 * iv * GH_LSE_SCALE64 : iv * (mean[c * D + d] - sh[d]) * GH_LSE_SCALE64;
            } else if (d == D && h == 1) v = cst[c];
        } else if (d == D && h == 1) v = GH_LSE_OFF64;              // padding component: never weighs anything
        rf_publish(P + i, v);
    }
}

// GMM.em_update (hmm_state.py:134-159) of one state by the whole block, from sums centred on `shift`
// (st: occupancy | sum r (x - shift) | sum r (x - shift)^2 per component).  Returns (block-uniform) whether the state goes on.
__device__ bool em_update_state(const rf_em_args& a, int s, int it, const double* __restrict__ st, int* __restrict__ lds_i /*[2]*/) {
    const int D = a.c.D, k = a.c.k, Wd = 1 + 2 * D, tid = threadIdx.x;
    if (tid == 0) { lds_i[0] = 0; lds_i[1] = 0; }
    __syncthreads();
    const double* sh = a.c.shift + (int64_t)s * D;
    int mine = 0, diff = 0;
    auto same = [](double x, double y) { return x == y || (x != x && y != y); };
    double mu_l[2], sg_l[2], w_l = 0.0;            // this thread's entries (k D <= 512 = 2 per thread of a 256-thread block)
    int n_l = 0;
    unsigned onept = 0xffu;                        // bit c: every entry of component c this thread holds says "one point"
    for (int i = tid; i < k * D; i += blockDim.x) {
        const int c = i / D, d = i - c * D;
        const int64_t at = ((int64_t)s * k + c) * D + d;
        const double m0 = a.mean[at], dm = m0 - sh[d];
        const double s0 = st[c * Wd], T1 = st[c * Wd + 1 + d], T2 = st[c * Wd + 1 + D + d];
        // all of the component's weight on ONE point (a frame, or identical ones): sum r y^2 . sum r = (sum r y)^2 up to the
        // rounding of the three sums -- and only then in every dimension at once (several points: the two sides differ by
        // s0^2 times the weighted variance, ~1 against 1e-15)
        // (as weighted mean and mean square, not as products of the sums: a component nobody is close to has s0 ~ 1e-200, and
        //  T1 T1, T2 s0 would both underflow to 0 and "agree")
        {
            const double mq = T1 / s0, qq = T2 / s0;
            if (!(s0 >= 1e-290 && fabs(mq * mq - qq) <= 3.6e-15 * fabs(qq))) onept &= ~(1u << c);
        }
        const double S1 = T1 - dm * s0;                                 // sum r (x - m0)
        const double S2 = T2 - dm * (2.0 * T1 - dm * s0);               // sum r (x - m0)^2
        const double occ = (s0 == 0) ? 1e-5 : s0;
        const double mu = (m0 * s0 + S1) / occ;
        const double dl = mu - m0;
        const double sg = (S2 - dl * (2.0 * S1 - dl * s0)) / occ;
        diff += !same(mu, m0) + !same(sg, a.var[at]);
        // update_models installs the covariance BEFORE the convergence test (hmm_state.py:149): a zero variance raises
        // LinAlgError there, whether or not the state would go on (the pack of the next pass only sees the states that do)
        // (sg < 0: a variance below the rounding noise of the centred sums -- a component with ~1e-22 of the weight, whose true
        //  variance of ~1e-19 the reference inverts for one more iteration before it collapses onto a frame and raises: numerically
        //  singular here already, and log(sg) would turn the whole state into NaN)
        //  The same for a variance within the noise ABOVE zero: sg <= 128 eps E[(x - shift)^2] -- a feature that is constant over the
        //  component's frames (the reference: exactly 0, LinAlgError) comes out as +-1e-16 y^2 here
        if (sg <= 0 || sg <= 2.8e-14 * (T2 / occ)) atomicOr(a.c.counter + 1, 16);
        rf_publish(a.mean + at, mu);
        rf_publish(a.var + at, sg);
        if (n_l < 2) { mu_l[n_l] = mu; sg_l[n_l] = sg; }
        ++n_l;
        mine += !rf_close(mu, a.old_mu[at]) + !rf_close(sg, a.old_sigma[at]);
    }
    if (tid < k) {
        const double w = st[tid * Wd] / a.nframes[s];
        diff += !same(w, a.weight[(int64_t)s * k + tid]);
        rf_publish(a.weight + (int64_t)s * k + tid, w);
        w_l = w;
        mine += !rf_close(w, a.old_w[(int64_t)s * k + tid]);
    }
    if (mine) atomicAdd(&lds_i[0], mine);
    if (diff) atomicAdd(&lds_i[1], diff);
    // A component that sits on one point: the reference's M-step (hmm_state.py:134-148) gets mu = that frame and a variance of
    // EXACTLY 0 -- update_models raises LinAlgError -- or, when mu comes out one ulp beside the frame or denormal-small weights
    // are left on other frames, a variance of 1e-32 .. 1e-74 of the spread, and NaN parameters one iteration later.  The centred
    // sums here leave rounding noise (+-1e-16 of the spread, either sign) in both cases, which `sg == 0` cannot see and which
    // would train on silently: both raise.
    for (int c = 0; c < k; ++c)
        if (__syncthreads_and((int)((onept >> c) & 1u)) && tid == 0) atomicOr(a.c.counter + 1, 16);
    __syncthreads();
    const int bad = lds_i[0], moved = lds_i[1];
    if (bad == 0) {                                  // np.allclose on all three: converged, the old values stay
        if (tid == 0) { rf_publish(a.c.active + s, (uint8_t)0); rf_publish(a.conv_at + s, (int32_t)it); }
        return false;
    }
    n_l = 0;
    for (int i = tid; i < k * D; i += blockDim.x, ++n_l) {
        const int64_t at = (int64_t)s * k * D + i;
        rf_publish(a.old_mu + at, n_l < 2 ? mu_l[n_l] : a.mean[at]);
        rf_publish(a.old_sigma + at, n_l < 2 ? sg_l[n_l] : a.var[at]);
    }
    if (tid < k) rf_publish(a.old_w + (int64_t)s * k + tid, w_l);
    // bit for bit the parameters that went in (NaN = NaN): a fixed point the allclose test cannot see (see
    // fit_em_update_kernel in gh_lockstep.hip) -- the state stops with the values the full loop would end with
    if (moved == 0) {
        if (tid == 0) rf_publish(a.c.active + s, (uint8_t)0);
        return false;
    }
    if (tid == 0) atomicAdd(a.c.counter + 2 + (it & 7), 1);
    return true;
}

// KSM: accumulator columns (steps of 4 features) the instantiation holds; EXACT: the state has exactly KSM steps, every
// loop over them is straight-line code (the shapes that matter: D = 12..15 and 36..39); otherwise KS <= KSM at run time
// HARD: the responsibilities are GIVEN -- 1 for the component a.hard_ids names for the frame, 0 for the others -- and the pass
// only sums: count | sum (x - shift) | sum (x - shift)^2 per (state, group), the one-pass variances of kmeans.py:171-177's
// random partitions (no operands, no densities, no update in the tail).
// TAIL: the multi-iteration form (its own instantiation: the loop around the kernel body and the values it keeps alive cost the
// ordinary launch 10-40 % when both forms shared one -- 121 -> 178 us for a full pass at k = 4)
template <int CG, int KSM, bool EXACT, bool HARD = false, bool TAIL = false>
__global__ __launch_bounds__(64 * RF_WAVES) __attribute__((amdgpu_waves_per_eu((CG == 1 || TAIL) && KSM <= 10 ? 3 : (CG == 2 && KSM > 10 ? 1 : 2), 8))) void refit_em_kernel(const rf_em_args a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = a.c.D, k = a.c.k, KS = EXACT ? KSM : rf_steps(D), TS = 4 * KS + 2;
    const int pstride = CG * 2 * KS * 16, bufsz = 16 * TS + 8;
    double* sP = sm;                              // [pstride]
    double* sT = sP + pstride;                    // [128]
    double* sBuf = sT + 128;                      // [RF_WAVES][bufsz]; afterwards the cross-wave sums
    int* sFlag = reinterpret_cast<int*>(sBuf + RF_WAVES * bufsz + 64);   // [4]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar slab addresses)
    constexpr bool tail = TAIL && !HARD;           // a TAIL launch: this workgroup stays for several iterations (see the loop's end)
    if (!tail && blockIdx.x == 0 && tid == 0) a.c.counter[2 + ((a.c.it + 1) & 7)] = 0;   // the next iteration's slot
    const int bid = a.c.item_ids ? a.c.item_ids[blockIdx.x] : (int)blockIdx.x;
    const rf_item item = a.c.items[bid];
    const int s = item.state;
    // the first slab's loads go out before anything else of the prologue (they need nothing but the item)
    const int nsl = (item.count + 15) >> 4;
    int sl = wv;
    rf_stage<KSM> stg;
    if (sl < nsl) stg.load(lane, a.c.X + (item.first + (int64_t)sl * 16) * D, min(16, item.count - sl * 16) * D);
    const uint8_t live = a.c.active[s];           // (tested below: the operands' loads share its round trip)
    if (!HARD) {
        for (int i = tid; i < pstride; i += 64 * RF_WAVES) sP[i] = a.P[(int64_t)s * pstride + i];
        if (tid < 128) sT[tid] = a.exp_tab[tid];
    }
    double* tb = sBuf + wv * bufsz;
    rf_init_pads(tb, lane, D, TS);
    stg.init(lane, D, TS, a.c.shift + (int64_t)s * D);
    if (!live) return;
    __syncthreads();
    // lane roles: operands (k, b, i) = (lane >> 4, (lane >> 2) & 3, lane & 3); results: (i, b, j) in the same places
    const int kk = lane >> 4, bq = (lane >> 2) & 3, lo = lane & 3;
    const int a1 = rf_frame(bq, lo) * TS + kk;                     // phase 1, A: x[F(b, i)][4 st + k]
    const int pb = 4 * lo + kk;                                    // phase 1, B: P[comp 4 cg + j][4 st + k]
    const int a2 = rf_frame(bq, kk) * TS + lo;                     // phase 2, A: x[F(b, k)][4 g + i]
    const int my_frame = rf_frame(bq, kk);                         // phase 1 result / phase 2 B: frame F(b, i), comp 4 cg + j
    const double thr = RF_LN_UNDERFLOW * GH_LSE_SCALE64;
    const int Wd = 1 + 2 * D, plen = k * Wd + 1;
    double* out = a.c.partial + (int64_t)bid * plen;
    const int i0 = a.c.item_ptr[s], i1 = a.c.item_ptr[s + 1];
    for (int j = 0;; ++j) {
    const int it = a.c.it + j;
    double cst[CG];                                                // the constant of component 4 cg + j (exponent-only test)
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) cst[cg] = HARD ? 0.0 : sP[((cg * 2 + 1) * KS + (D >> 2)) * 16 + 4 * lo + (D & 3)];
    double Sx[CG][KSM], Sq[CG][KSM];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int g = 0; g < KSM; ++g) Sx[cg][g] = Sq[cg][g] = 0.0;

    for (; sl < nsl; sl += RF_WAVES) {
        const int cnt = min(16, item.count - sl * 16);
        stg.store(tb);
        if (sl + RF_WAVES < nsl)
            stg.load(lane, a.c.X + (item.first + (int64_t)(sl + RF_WAVES) * 16) * D, min(16, item.count - (sl + RF_WAVES) * 16) * D);
        double l[CG];
        const bool act = my_frame < cnt;
        if (HARD) {
            const int id = act ? a.hard_ids[item.first + (int64_t)sl * 16 + my_frame] : -1;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) l[cg] = (id == 4 * cg + lo) ? 1.0 : 0.0;
        } else {
        // ---- phase 1: scaled log-densities ----
        double aq[CG], al[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) aq[cg] = al[cg] = 0.0;
#pragma unroll
        for (int st = 0; st < KSM; ++st) {
            if (!EXACT && st >= KS) break;
            const double x = tb[a1 + 4 * st], x2 = x * x;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                aq[cg] = rf_mfma(x2, sP[((cg * 2 + 0) * KS + st) * 16 + pb], aq[cg]);
                al[cg] = rf_mfma(x, sP[((cg * 2 + 1) * KS + st) * 16 + pb], al[cg]);
            }
        }
        // ---- responsibilities: the components of a frame sit in the CG registers of a quad of lanes ----
        double mx = -INFINITY;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) { l[cg] = aq[cg] + al[cg]; mx = rf_vmax(mx, l[cg]); }
        mx = rf_vmax(mx, rf_dpp<0xB1>(mx));
        mx = rf_vmax(mx, rf_dpp<0x4E>(mx));
        double sum = 0.0;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            // the reference's linear-domain densities: a product that underflows is exactly 0 (its exponent alone, or with
            // weight and normaliser)
            const bool dead = (l[cg] < thr) || (l[cg] - cst[cg] < thr);
            const double e = rf_exp2s(l[cg] - mx, sT);
            l[cg] = dead ? 0.0 : e;
            sum += l[cg];
        }
        sum += rf_dpp<0xB1>(sum);
        sum += rf_dpp<0x4E>(sum);
        const double inv = (sum == 0.0) ? 0.0 : rf_rcp(sum);          // (a NaN stays a NaN)
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) l[cg] = act ? l[cg] * inv : 0.0;
        }
        // ---- phase 2: S[col][comp] += sum over the block's four frames of x r, x^2 r (col D: the occupancy) ----
#pragma unroll
        for (int g = 0; g < KSM; ++g)
            if (EXACT || g < KS) {
                const double y = tb[a2 + 4 * g], y2 = y * y;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    Sx[cg][g] = rf_mfma(y, l[cg], Sx[cg][g]);
                    Sq[cg][g] = rf_mfma(y2, l[cg], Sq[cg][g]);
                }
            }
    }
    // ---- the item's sums: the four waves added in wave order, the four blocks of an entry when the slab is written ----
    __syncthreads();
    double* sRed = sBuf;                                             // [CG][2][KS][64]
    for (int w = 0; w < RF_WAVES; ++w) {
        if (wv == w) {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int g = 0; g < KSM; ++g)
                    if (EXACT || g < KS) {
                        double* p = sRed + ((cg * 2) * KS + g) * 64 + lane;
                        if (w == 0) { p[0] = Sx[cg][g]; p[KS * 64] = Sq[cg][g]; }
                        else { p[0] += Sx[cg][g]; p[KS * 64] += Sq[cg][g]; }
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < CG * 2 * KS * 16; e += 64 * RF_WAVES) {      // entry (cg, h, g, i, j): lanes 16 i + 4 b + j, b = 0..3
        const int j = e & 3, i = (e >> 2) & 3, g = (e >> 4) % KS, h = ((e >> 4) / KS) & 1, cg = (e >> 4) / (2 * KS);
        const int c = 4 * cg + j, col = 4 * g + i;
        const double* p = sRed + ((cg * 2 + h) * KS + g) * 64 + 16 * i + j;
        const double v = (p[0] + p[4]) + (p[8] + p[12]);
        if (c < k) {
            if (col < D) rf_publish(out + c * Wd + 1 + h * D + col, v);
            else if (col == D && h == 0) rf_publish(out + c * Wd, v);
        }
    }
    const bool last = rf_arrive(a.c.done, s, i1 - i0, sFlag);
    bool alive = false;
    if (last) {
        // ---- last workgroup of the state: slabs in a fixed order -> statistics -> update -> next operands ----
        double* st = a.stats + (int64_t)s * plen;
        rf_reduce_slabs(a.c.partial, plen, i0, i1, k * Wd, st, sBuf);
        if (tid == 0) rf_publish(st + k * Wd, 0.0);       // (log-likelihood column of the call-by-call layout: not computed here)
        if (!HARD && a.c.fused) {
            __syncthreads();
            alive = em_update_state(a, s, it, st, sFlag);
            if (alive) {
                __syncthreads();
                em_pack_state(a, s, sBuf);
            }
        }
    }
    if (!tail || j + 1 >= a.c.n_iter) return;
    // ---- TAIL launch: every workgroup of the state waits for its last one to finish the iteration (the state's generation
    //      word: j + 1 = go on, -1 = the state has stopped), then takes the new operands; polls are bounded ----
    if (last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            rf_publish(a.c.gen + s, alive ? j + 1 : -1);
            sFlag[2] = alive ? j + 1 : -1;
        }
    } else if (tid == 0) {
        int g = j, polls = 0;
        while ((g = __hip_atomic_load(a.c.gen + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == j && ++polls < RF_TAIL_POLLS)
            __builtin_amdgcn_s_sleep(8);
        if (g == j) { atomicOr(a.c.counter + 1, 32); g = -1; }     // (never seen: some workgroup of the grid was not resident)
        sFlag[2] = g;
    }
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (sFlag[2] < 0) return;
    sl = wv;
    if (sl < nsl) stg.load(lane, a.c.X + (item.first + (int64_t)sl * 16) * D, min(16, item.count - sl * 16) * D);
    for (int i = tid; i < pstride; i += 64 * RF_WAVES) sP[i] = a.P[(int64_t)s * pstride + i];
    rf_init_pads(tb, lane, D, TS);                 // (the slab buffers served as scratch for the sums since)
    __syncthreads();
    }
}

// variances of the random partitions (kmeans.py:171-177: np.cov(...).diagonal(), ddof 1) from the HARD pass's sums:
// (sum x'^2 - (sum x')^2 / n) / (n - 1), x' = x - shift (the shift sits inside the data: nothing cancels); n <= 1 -> NaN
// like numpy's 0 * (1 / 0).  first_only: cluster 0's variance for every cluster (the sharded rule, sums over all ranks).
__global__ void refit_partvar_kernel(int S, int k, int D, const double* __restrict__ stats, int first_only, double* __restrict__ cov) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * k * D) return;
    const int s = i / (k * D), c = first_only ? 0 : (i / D) % k, d = i % D, Wd = 1 + 2 * D;
    const double* st = stats + (int64_t)s * (k * Wd + 1) + (int64_t)c * Wd;
    const double n = st[0], s1 = st[1 + d], s2 = st[1 + D + d];
    cov[i] = n > 1.0 ? (s2 - s1 * s1 / n) / (n - 1.0) : NAN;
}

// one block per state: (update +) pack; the first pack of a session, and the update behind a collective
__global__ __launch_bounds__(256) void refit_em_update_kernel(const rf_em_args a, int pack_only) {
    __shared__ double lds[2 * 8 * 64 + 8];
    __shared__ int lds_i[2];
    const int s = blockIdx.x;
    if (s == 0 && threadIdx.x == 0 && !pack_only) a.c.counter[2 + ((a.c.it + 1) & 7)] = 0;
    if (!a.c.active[s]) return;
    if (!pack_only) {
        const int plen = a.c.k * (1 + 2 * a.c.D) + 1;
        if (!em_update_state(a, s, a.c.it, a.stats + (int64_t)s * plen, lds_i)) return;
        __syncthreads();
    }
    em_pack_state(a, s, lds);
}

// =========================================================================================== k-means
// operands: P[(cg KS + st) 16 + 4 j + kk] = -2 iv c' of centroid 4 cg + j at feature 4 st + kk (c' = centroid - shift), at
// feature D the constant sum iv c'^2; then one more group [KS][16]: iv at every j (the frame's own term sum iv x'^2).
// kscale[s]: 2 max_c sum iv c'^2 + 2 |logdet| -- with 2 sum iv x'^2 the size of everything a distance is added up from.
__device__ void km_pack_state(const rf_km_args& a, int s, double* __restrict__ lds /*[k D + k]*/) {
    const int D = a.c.D, k = a.c.k, KS = rf_steps(D), CG = rf_comp_groups(k), tid = threadIdx.x;
    const double* cen = a.cent + (int64_t)s * k * D;
    const double* var = a.var + (int64_t)s * k * D;            // row 0
    const double* sh = a.c.shift + (int64_t)s * D;
    double* cc = lds + k * D;
    for (int i = tid; i < k * D; i += blockDim.x) {
        const int d = i % D;
        const double c = cen[i] - sh[d];
        lds[i] = c * (1.0 / var[d]) * c;
    }
    __syncthreads();
    if (tid < k) {
        double acc = 0;
        for (int d = 0; d < D; ++d) acc += lds[tid * D + d];
        cc[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        double m = 0;
        for (int c = 0; c < k; ++c) { const double v = fabs(cc[c]); if (v == v && v > m && v < INFINITY) m = v; }
        rf_publish(a.kscale + s, 2.0 * m + 2.0 * fabs(a.logdet[s]));
    }
    double* P = a.P + (int64_t)s * ((CG + 1) * KS * 16);
    for (int i = tid; i < (CG + 1) * KS * 16; i += blockDim.x) {
        const int kk = i & 3, j = (i >> 2) & 3, st = (i >> 4) % KS, cg = (i >> 4) / KS;
        const int c = 4 * cg + j, d = 4 * st + kk;
        double v = 0.0;
        if (cg == CG) v = d < D ? 1.0 / var[d] : 0.0;
        else if (c < k) {
            if (d < D) v = -2.0 * (1.0 / var[d]) * (cen[c * D + d] - sh[d]);
            else if (d == D) v = cc[c];
        } else if (d == D) v = 1e300;                               // padding centroid: never the nearest
        rf_publish(P + i, v);
    }
}

// centroid update + stop rule of one state by the whole block (kmeans.py:187-192 with lockstep's rule "no assignment
// changed"): su = sum (x - shift) | count per cluster, then the number of assignments that changed
__device__ bool km_update_state(const rf_km_args& a, int s, int it, const double* __restrict__ su) {
    const int D = a.c.D, k = a.c.k, tid = threadIdx.x;
    const double* sh = a.c.shift + (int64_t)s * D;
    double* ce = a.cent + (int64_t)s * k * D;
    for (int i = tid; i < k * D; i += blockDim.x) {
        const int c = i / D, d = i - c * D;
        rf_publish(ce + i, sh[d] + su[c * (D + 1) + d] / su[c * (D + 1) + D]);   // an empty cluster: 0 / 0 = NaN, like np.mean of nothing
    }
    const bool stop = su[k * (D + 1)] == 0.0;
    if (tid == 0) {
        rf_publish(a.iters + s, a.iters[s] + 1);
        if (stop) rf_publish(a.c.active + s, (uint8_t)0); else atomicAdd(a.c.counter + 2 + (it & 7), 1);
    }
    return !stop;
}

// The reference's own arithmetic for ONE frame (kmeans.py:183 -> mahalanobis, hmm_state.py:58: m / variance * m summed in
// order, + log-determinant term, np.argmin: the first minimum, a NaN before everything)
__device__ int km_exact(const double* __restrict__ x, const double* __restrict__ cen, const double* __restrict__ var, double ld, int k, int D) {
    double best = 0;
    int bi = 0;
    for (int c = 0; c < k; ++c) {
        double q = 0;
        for (int d = 0; d < D; ++d) { const double t = cen[c * D + d] - x[d]; q += t / var[d] * t; }
        const double dist = ld + 0.5 * q;
        if (c == 0 || dist < best || (dist != dist && best == best)) { best = dist; bi = c; }
    }
    return bi;
}

template <int CG, int KSM, bool EXACT, bool TAIL = false>
__global__ __launch_bounds__(64 * RF_WAVES) __attribute__((amdgpu_waves_per_eu(KSM <= 10 ? 3 : 2, 8))) void refit_km_kernel(const rf_km_args a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = a.c.D, k = a.c.k, KS = EXACT ? KSM : rf_steps(D), TS = 4 * KS + 2;
    const int pstride = (CG + 1) * KS * 16, bufsz = 16 * TS + 8;
    double* sP = sm;                              // [pstride]
    double* sBuf = sP + pstride;                  // [RF_WAVES][bufsz]; afterwards the cross-wave sums
    int* sFlag = reinterpret_cast<int*>(sBuf + RF_WAVES * bufsz + 64);   // [8]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar slab addresses)
    constexpr bool tail = TAIL;                    // a TAIL launch: see refit_em_kernel
    if (!tail && blockIdx.x == 0 && tid == 0) a.c.counter[2 + ((a.c.it + 1) & 7)] = 0;
    const int bid = a.c.item_ids ? a.c.item_ids[blockIdx.x] : (int)blockIdx.x;
    const rf_item item = a.c.items[bid];
    const int s = item.state;
    const int nsl = (item.count + 15) >> 4;
    int sl = wv;
    rf_stage<KSM> stg;
    if (sl < nsl) stg.load(lane, a.c.X + (item.first + (int64_t)sl * 16) * D, min(16, item.count - sl * 16) * D);
    const uint8_t live = a.c.active[s];
    for (int i = tid; i < pstride; i += 64 * RF_WAVES) sP[i] = a.P[(int64_t)s * pstride + i];
    double* tb = sBuf + wv * bufsz;
    rf_init_pads(tb, lane, D, TS);
    stg.init(lane, D, TS, a.c.shift + (int64_t)s * D);
    if (tid < 4) sFlag[4 + tid] = 0;
    if (!live) return;
    __syncthreads();
    const int kk = lane >> 4, bq = (lane >> 2) & 3, lo = lane & 3;
    const int a1 = rf_frame(bq, lo) * TS + kk;
    const int pb = 4 * lo + kk;
    const int a2 = rf_frame(bq, kk) * TS + lo;
    const int my_frame = rf_frame(bq, kk);
    const double* cen = a.cent + (int64_t)s * k * D;
    const double* var = a.var + (int64_t)s * k * D;
    const double ld = a.logdet[s];
    const int plen = k * (D + 1) + 1;
    double* out = a.c.partial + (int64_t)bid * plen;
    const int i0 = a.c.item_ptr[s], i1 = a.c.item_ptr[s + 1];
    for (int j = 0;; ++j) {
    const int it = a.c.it + j;
    const double kscale = a.kscale[s];
    double Sx[CG][KSM];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int g = 0; g < KSM; ++g) Sx[cg][g] = 0.0;
    int n_changed = 0;

    for (; sl < nsl; sl += RF_WAVES) {
        const int cnt = min(16, item.count - sl * 16);
        const int64_t f0 = item.first + (int64_t)sl * 16;
        stg.store(tb);
        if (sl + RF_WAVES < nsl)
            stg.load(lane, a.c.X + (item.first + (int64_t)(sl + RF_WAVES) * 16) * D, min(16, item.count - (sl + RF_WAVES) * 16) * D);
        const bool act = my_frame < cnt;
        const int old_id = (act && lo == 0) ? a.ids[f0 + my_frame] : 0;
        // ---- phase 1: g[c] = sum iv c'^2 - 2 sum iv c' x' (the distance up to the frame's own term xx and the state's constants) ----
        double gq[CG], xx = 0.0;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) gq[cg] = 0.0;
#pragma unroll
        for (int st = 0; st < KSM; ++st) {
            if (!EXACT && st >= KS) break;
            const double x = tb[a1 + 4 * st];
            xx = rf_mfma(x * x, sP[(CG * KS + st) * 16 + pb], xx);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) gq[cg] = rf_mfma(x, sP[(cg * KS + st) * 16 + pb], gq[cg]);
        }
        // ---- arg-min over the quad's 4 CG scores: np.argmin = the first minimum, a NaN before everything ----
        double best = INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            gq[cg] = (gq[cg] != gq[cg]) ? -INFINITY : gq[cg];       // (a NaN centroid: distance NaN -> wins; the lowest such index)
            if (gq[cg] < best) { best = gq[cg]; bi = 4 * cg + lo; }
        }
#define RF_AMIN(CTRL)                                                                  \
        { const double ob = rf_dpp<CTRL>(best); const int oi = rf_dpp<CTRL>(bi);         \
          if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; } }
        RF_AMIN(0xB1) RF_AMIN(0x4E)
#undef RF_AMIN
        // how many scores lie within the rounding of either computation of the best one?
        const double tau = 9.094947017729282e-13 * (2.0 * xx + kscale);          // 2^-40 of the terms' size
        int near = 0;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) near += (4 * cg + lo < k) && (gq[cg] <= best + tau);
        near += rf_dpp<0xB1>(near);
        near += rf_dpp<0x4E>(near);
        const bool redo = (best > -INFINITY && near > 1) || !(xx - xx == 0.0) || !(best < INFINITY);
        if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {              // rare: the reference's own operations decide
            if (redo && act && lo == 0) bi = km_exact(a.c.X + (f0 + my_frame) * D, cen, var, ld, k, D);
            const int b0 = rf_dpp<0x00>(bi);                          // (lane j = 0 of the quad holds the answer)
            if (redo) bi = b0;
        }
        if (act && lo == 0) {
            a.ids[f0 + my_frame] = bi;
            n_changed += old_id != bi;
        }
        // ---- phase 2: sums of the frames of every cluster (col D: the count) ----
        double r[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) r[cg] = (act && bi == 4 * cg + lo) ? 1.0 : 0.0;
#pragma unroll
        for (int g = 0; g < KSM; ++g)
            if (EXACT || g < KS) {
                const double y = tb[a2 + 4 * g];
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) Sx[cg][g] = rf_mfma(y, r[cg], Sx[cg][g]);
            }
    }
    {   // assignments of this wave that changed
        int n = n_changed;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
        if (lane == 0) sFlag[4 + wv] = n;
    }
    __syncthreads();
    double* sRed = sBuf;                                             // [CG][KS][64]
    for (int w = 0; w < RF_WAVES; ++w) {
        if (wv == w) {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int g = 0; g < KSM; ++g)
                    if (EXACT || g < KS) {
                        double* p = sRed + (cg * KS + g) * 64 + lane;
                        if (w == 0) p[0] = Sx[cg][g]; else p[0] += Sx[cg][g];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < CG * KS * 16; e += 64 * RF_WAVES) {
        const int j = e & 3, i = (e >> 2) & 3, g = (e >> 4) % KS, cg = (e >> 4) / KS;
        const int c = 4 * cg + j, col = 4 * g + i;
        const double* p = sRed + (cg * KS + g) * 64 + 16 * i + j;
        if (c < k && col <= D) rf_publish(out + c * (D + 1) + col, (p[0] + p[4]) + (p[8] + p[12]));
    }
    if (tid == 0) rf_publish(out + k * (D + 1), (double)(sFlag[4] + sFlag[5] + sFlag[6] + sFlag[7]));
    const bool last = rf_arrive(a.c.done, s, i1 - i0, sFlag);
    bool alive = false;
    if (last) {
        double* su = a.sums + (int64_t)s * plen;
        rf_reduce_slabs(a.c.partial, plen, i0, i1, plen, su, sBuf);
        if (a.c.fused) {
            __syncthreads();
            alive = km_update_state(a, s, it, su);
            if (alive) {
                __syncthreads();
                km_pack_state(a, s, sBuf);
            }
        }
    }
    if (!tail || j + 1 >= a.c.n_iter) return;
    // ---- TAIL launch: on to the state's next iteration, or out (see refit_em_kernel) ----
    if (last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            rf_publish(a.c.gen + s, alive ? j + 1 : -1);
            sFlag[2] = alive ? j + 1 : -1;
        }
    } else if (tid == 0) {
        int g = j, polls = 0;
        while ((g = __hip_atomic_load(a.c.gen + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == j && ++polls < RF_TAIL_POLLS)
            __builtin_amdgcn_s_sleep(8);
        if (g == j) { atomicOr(a.c.counter + 1, 32); g = -1; }
        sFlag[2] = g;
    }
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (sFlag[2] < 0) return;
    sl = wv;
    if (sl < nsl) stg.load(lane, a.c.X + (item.first + (int64_t)sl * 16) * D, min(16, item.count - sl * 16) * D);
    for (int i = tid; i < pstride; i += 64 * RF_WAVES) sP[i] = a.P[(int64_t)s * pstride + i];
    rf_init_pads(tb, lane, D, TS);
    n_changed = 0;
    __syncthreads();
    }
}

__global__ __launch_bounds__(256) void refit_km_update_kernel(const rf_km_args a, int pack_only) {
    __shared__ double lds[8 * 64 + 8];
    const int s = blockIdx.x;
    if (s == 0 && threadIdx.x == 0 && !pack_only) a.c.counter[2 + ((a.c.it + 1) & 7)] = 0;
    if (!a.c.active[s]) return;
    if (!pack_only) {
        const int plen = a.c.k * (a.c.D + 1) + 1;
        if (!km_update_state(a, s, a.c.it, a.sums + (int64_t)s * plen)) return;
        __syncthreads();
    }
    km_pack_state(a, s, lds);
}

int rf_ksm(int D) { const int ks = rf_steps(D); return ks <= 4 ? 4 : (ks <= 10 ? 10 : 17); }

}  // namespace

// two component groups: the 2 KS accumulators per group fill the registers beyond that (k > 8 keeps the tile kernels)
bool rf_supported(int k, int D) { return k >= 1 && k <= 8 && D >= 2 && D <= 64; }

size_t rf_em_lds(int k, int D) {
    return ((size_t)rf_em_pstride(k, D) + 128 + (size_t)RF_WAVES * (16 * rf_row_stride(D) + 8) + 64) * 8 + 64;
}
size_t rf_km_lds(int k, int D) {
    return ((size_t)rf_km_pstride(k, D) + (size_t)RF_WAVES * (16 * rf_row_stride(D) + 8) + 64) * 8 + 64;
}

// A TAIL launch (a.c.n_iter > 1) is only made when every workgroup of the grid is resident at once -- its workgroups wait
// for each other between iterations: the instantiation's own occupancy times the CUs, with a quarter held back; otherwise
// `refused` and the caller keeps the ordinary launches.
#define RF_GO(KFN, LDS)                                                                                         \
    do {                                                                                                        \
        if (a.c.n_iter > 1) {                                                                                   \
            int occ = 0;                                                                                        \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)KFN, 64 * RF_WAVES, LDS) != hipSuccess) occ = 0; \
            if ((int64_t)n_items * 4 > (int64_t)occ * ctx->n_cu * 3) { refused = true; break; }                 \
        }                                                                                                       \
        hipLaunchKernelGGL(KFN, dim3((unsigned)n_items), dim3(64 * RF_WAVES), LDS, ctx->stream, a);             \
        launched = true;                                                                                        \
    } while (0)
#define RF_DISPATCH_EM(CGV, KSV, EX, LDS)                                                                       \
    do {                                                                                                        \
        if (!launched && !refused && (CGV) == c_g && (KSV) == ksm && (!(EX) || rf_steps(a.c.D) == (KSV))) {     \
            if (a.c.n_iter > 1) RF_GO((refit_em_kernel<CGV, KSV, EX, false, true>), LDS);                       \
            else RF_GO((refit_em_kernel<CGV, KSV, EX, false, false>), LDS);                                     \
        }                                                                                                       \
    } while (0)
#define RF_DISPATCH_KM(CGV, KSV, EX, LDS)                                                                       \
    do {                                                                                                        \
        if (!launched && !refused && (CGV) == c_g && (KSV) == ksm && (!(EX) || rf_steps(a.c.D) == (KSV))) {     \
            if (a.c.n_iter > 1) RF_GO((refit_km_kernel<CGV, KSV, EX, true>), LDS);                              \
            else RF_GO((refit_km_kernel<CGV, KSV, EX, false>), LDS);                                            \
        }                                                                                                       \
    } while (0)
#define RF_DISPATCH_HARD(CGV, KSV, EX, LDS)                                                                     \
    do {                                                                                                        \
        if (!launched && (CGV) == c_g && (KSV) == ksm && (!(EX) || rf_steps(a.c.D) == (KSV))) {                 \
            hipLaunchKernelGGL((refit_em_kernel<CGV, KSV, EX, true>), dim3((unsigned)n_items), dim3(64 * RF_WAVES), LDS, ctx->stream, a); \
            launched = true;                                                                                    \
        }                                                                                                       \
    } while (0)
#define RF_ALL(DISPATCH, LDS)                                                                \
    DISPATCH(1, 4, true, LDS); DISPATCH(2, 4, true, LDS);                                    \
    DISPATCH(1, 10, true, LDS); DISPATCH(2, 10, true, LDS);                                  \
    DISPATCH(1, 4, false, LDS); DISPATCH(2, 4, false, LDS);                                  \
    DISPATCH(1, 10, false, LDS); DISPATCH(2, 10, false, LDS);                                \
    DISPATCH(1, 17, false, LDS); DISPATCH(2, 17, false, LDS)

// returns 1 when a TAIL launch was refused (not every workgroup would be resident): nothing was launched
int rf_launch_em(gh_ctx* ctx, const rf_em_args& a, int n_items) {
    if (n_items <= 0) return GH_OK;
    const int c_g = rf_comp_groups(a.c.k), ksm = rf_ksm(a.c.D);
    const size_t lds = rf_em_lds(a.c.k, a.c.D);
    bool launched = false, refused = false;
    RF_ALL(RF_DISPATCH_EM, lds);
    if (refused) return 1;
    GH_REQUIRE(launched, "refit: k=%d, D=%d has no kernel", a.c.k, a.c.D);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int rf_launch_partition_sums(gh_ctx* ctx, const rf_em_args& a, int n_items) {
    if (n_items <= 0) return GH_OK;
    const int c_g = rf_comp_groups(a.c.k), ksm = rf_ksm(a.c.D);
    const size_t lds = rf_em_lds(a.c.k, a.c.D);
    bool launched = false;
    RF_ALL(RF_DISPATCH_HARD, lds);
    GH_REQUIRE(launched, "refit: k=%d, D=%d has no kernel", a.c.k, a.c.D);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int rf_launch_partvar(gh_ctx* ctx, int S, int k, int D, const double* stats, int first_only, double* cov) {
    hipLaunchKernelGGL(refit_partvar_kernel, dim3((unsigned)((S * k * D + 255) / 256)), dim3(256), 0, ctx->stream, S, k, D, stats, first_only, cov);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int rf_launch_km(gh_ctx* ctx, const rf_km_args& a, int n_items) {
    if (n_items <= 0) return GH_OK;
    const int c_g = rf_comp_groups(a.c.k), ksm = rf_ksm(a.c.D);
    const size_t lds = rf_km_lds(a.c.k, a.c.D);
    bool launched = false, refused = false;
    RF_ALL(RF_DISPATCH_KM, lds);
    if (refused) return 1;
    GH_REQUIRE(launched, "refit: k=%d, D=%d has no kernel", a.c.k, a.c.D);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int rf_launch_em_update(gh_ctx* ctx, const rf_em_args& a, int pack_only) {
    hipLaunchKernelGGL(refit_em_update_kernel, dim3((unsigned)a.c.S), dim3(256), 0, ctx->stream, a, pack_only);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int rf_launch_km_update(gh_ctx* ctx, const rf_km_args& a, int pack_only) {
    hipLaunchKernelGGL(refit_km_update_kernel, dim3((unsigned)a.c.S), dim3(256), 0, ctx->stream, a, pack_only);
    GH_HIP(hipGetLastError());
    return GH_OK;
}
