// MFCC front-end on the GPU (first half): PCM -> 40 log mel filterbank
// energies -> 13 cepstra per frame, the arithmetic of mfcc_features (sr/feature/feature.py:43-82) in
// fp64.  One wave per frame, four frames per workgroup:
//   pre-emphasis (:45-46, product and difference rounded separately like numpy) -> the frame's samples
//   into a zero padded power-of-two buffer, centred (:25-40) -> Hamming window over the padded length
//   (:52) -> 512-point radix-2 FFT in LDS -> power spectrum / 512 (:54-56) -> mel filters (:58-75;
//   every filter only over its own bin range) -> log10 with eps for zeros (:76-78) -> DCT-II ortho
//   rows 1..13 (:80-81).
// All tables (window, twiddles, filterbank, DCT rows) are built on the host exactly as the reference
// builds them (np.hamming, np.linspace, floor((NFFT+1) hz / rate), scipy's ortho DCT-II scaling).
#include "gh_internal.h"
#include "gh_host.h"

namespace {

constexpr int NFFT = 512, NBIN = NFFT / 2 + 1, NFILT = 40, NCEPS = 13;

struct MfccTables {           // device pointers into one scratch block
    const double* window;     // [NFFT]   hamming(pad_w) in [0, pad_w), 0 behind
    const double* tw;         // [NFFT][2] cos / -sin of 2 pi k / NFFT
    const double* wup;        // [NBIN] weight of bin k in the ASCENDING half of the filter that peaks right of it
    const double* wdn;        // [NBIN] weight of bin k in the DESCENDING half of the filter that peaks at / left of it
    const int* seg;           // [NFILT + 2] the mel bin points: segment s = bins [seg[s], seg[s+1])
    const double* dct;        // [NCEPS][NFILT]
};

struct MfccArgs {
    const void* pcm; int fmt;
    const int64_t* s_off; const int64_t* f_off; const int32_t* f_utt;
    int64_t U, N;
    int flen, fstep, pad_left;
    MfccTables t;
    double* out_fb; double* out_mfcc;
};

template <int FMT> __device__ __forceinline__ double sample_at(const void* pcm, int64_t i) {
    if (FMT == 0) return (double)static_cast<const int16_t*>(pcm)[i];
    if (FMT == 1) return (double)static_cast<const float*>(pcm)[i];
    return static_cast<const double*>(pcm)[i];
}

typedef double c2 __attribute__((ext_vector_type(2)));   // (re, im)

__device__ __forceinline__ c2 mul_negi(c2 v) { return (c2){v.y, -v.x}; }                       // v * (-i)
__device__ __forceinline__ c2 cmul(c2 a, c2 w) { return (c2){a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }

// in-place 8-point forward DFT (decimation in frequency), natural output order
__device__ __forceinline__ void dft8(c2 (&v)[8]) {
    constexpr double R = 0.70710678118654752440;
    c2 t[4], u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = v[j] + v[j + 4]; u[j] = v[j] - v[j + 4]; }
    u[1] = (c2){(u[1].x + u[1].y) * R, (u[1].y - u[1].x) * R};      // * W8^1
    u[2] = mul_negi(u[2]);                                          // * W8^2
    u[3] = (c2){(u[3].y - u[3].x) * R, -(u[3].x + u[3].y) * R};     // * W8^3
    auto dft4 = [](const c2 (&x)[4], c2& o0, c2& o1, c2& o2, c2& o3) {
        const c2 s0 = x[0] + x[2], s1 = x[0] - x[2], s2 = x[1] + x[3], s3 = mul_negi(x[1] - x[3]);
        o0 = s0 + s2; o2 = s0 - s2; o1 = s1 + s3; o3 = s1 - s3;
    };
    dft4(t, v[0], v[2], v[4], v[6]);
    dft4(u, v[1], v[3], v[5], v[7]);
}

// One wave per PAIR of frames (A, B): z = a + i b goes through ONE 512-point complex FFT and the two
// real spectra are separated afterwards.  512 = 8 x 8 x 8: three radix-8 passes in registers (each lane
// holds 8 points), two transposes through LDS (padded, 16-byte accesses) -- instead of 9 radix-2 stages
// with a barrier and 8 LDS accesses per butterfly each.
template <int FMT>
__global__ __launch_bounds__(256) void mfcc_kernel(MfccArgs a) {
    constexpr int S1 = 72, S2 = 9;                       // padded strides of the two transposes (elements)
    __shared__ double s_x[4][8 * S1];                    // 4608 B per wave (the transposes move re and im one after
                                                         // the other: half the LDS, twice the resident waves), reused by every phase
    __shared__ double s_lfb[4][2][NFILT];
    __shared__ double s_wup[NBIN], s_wdn[NBIN];          // per-bin filter weights, shared by the block
    for (int k = threadIdx.x; k < NBIN; k += 256) { s_wup[k] = a.t.wup[k]; s_wdn[k] = a.t.wdn[k]; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* ex = s_x[wv];
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wv) * 2;   // frames n0 (A) and n0 + 1 (B)
    int64_t s0[2] = {0, 0}, slen[2] = {0, 0}, sbase[2] = {0, 0};
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t n = n0 + h;
        live[h] = n < a.N;
        if (live[h]) {
            const int u = a.f_utt[n];          // utterance of this frame (host-built table)
            sbase[h] = a.s_off[u];
            slen[h] = a.s_off[u + 1] - sbase[h];
            s0[h] = (n - a.f_off[u]) * a.fstep;
        }
    }
#ifdef GH_MFCC_TIMING
    long long tk[8]; int ti = 0;
#define TK() tk[ti++] = clock64()
#else
#define TK()
#endif
    TK();
    // ---- windowed, zero padded frames: lane l holds z[l + 64 j], j = 0..7 ----
    // (every load is unconditional on a clamped address, so all 32 of them are in flight together;
    //  a load behind a branch costs one memory round trip each)
    c2 v[8];
    double cur[2][8], prv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t last = sbase[h] + (slen[h] > 0 ? slen[h] - 1 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int64_t p = sbase[h] + s0[h] + (lane + 64 * j - a.pad_left);
            p = p < sbase[h] ? sbase[h] : (p > last ? last : p);
            cur[h][j] = sample_at<FMT>(a.pcm, p);
            prv[h][j] = sample_at<FMT>(a.pcm, p > sbase[h] ? p - 1 : p);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int js = lane + 64 * j - a.pad_left;
        const double w = a.t.window[lane + 64 * j];
        double xs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool in = live[h] && js >= 0 && js < a.flen && s0[h] + js < slen[h];
            const double x = (s0[h] + js == 0) ? cur[h][j] : __dsub_rn(cur[h][j], __dmul_rn(0.97, prv[h][j]));
            xs[h] = in ? x * w : 0.0;
        }
        v[j] = (c2){xs[0], xs[1]};
    }
    TK();
    // ---- pass 1: DFT over j, twiddle W512^(l q); transpose so that lane (l1 + 8 q) holds l2 = 0..7 ----
    dft8(v);
#pragma unroll
    for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], *reinterpret_cast<const c2*>(a.t.tw + 2 * (lane * q)));
    // component-wise transpose through LDS: v[i] goes to slot wr(i), comes back from slot rd(i)
    auto transpose = [&](auto wr, auto rd) {
        double tx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ex[wr(i)] = v[i].x;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) tx[i] = ex[rd(i)];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) ex[wr(i)] = v[i].y;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c2){tx[i], ex[rd(i)]};
        __syncthreads();
    };
    const int l1 = lane & 7, qq = lane >> 3;
    transpose([&](int q) { return q * S1 + lane; }, [&](int l2) { return qq * S1 + l1 + 8 * l2; });
    // ---- pass 2: DFT over l2, twiddle W64^(l1 q'); transpose so that lane (q + 8 q') holds l1 = 0..7 ----
    dft8(v);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) v[q2] = cmul(v[q2], *reinterpret_cast<const c2*>(a.t.tw + 2 * (8 * l1 * q2)));
    transpose([&](int q2) { return (qq + 8 * q2) * S2 + l1; }, [&](int i) { return lane * S2 + i; });
    // ---- pass 3: DFT over l1: register p holds Z[lane + 64 p] ----
    dft8(v);
    TK();
    // ---- separate the two real spectra, power / NFFT for bins 0..256: the partner Z[N - k] comes through LDS ----
    double wx[5], wy[5];
#pragma unroll
    for (int p = 0; p < 8; ++p) ex[lane + 64 * p] = v[p].x;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) wx[i] = ex[(NFFT - (lane + 64 * i)) & (NFFT - 1)];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; ++p) ex[lane + 64 * p] = v[p].y;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) wy[i] = ex[(NFFT - (lane + 64 * i)) & (NFFT - 1)];
    __syncthreads();
    double* pw = ex;                                         // [2][NBIN + pad]
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int k = lane + 64 * i;
        const c2 z = (i < 4) ? v[i] : v[4];                  // k = 256 sits in lane 0, register 4
        const double ar = z.x + wx[i], ai = z.y - wy[i], br = z.x - wx[i], bi = z.y + wy[i];
        if (k < NBIN) {
            pw[k] = (ar * ar + ai * ai) * (0.25 / NFFT);
            pw[264 + k] = (br * br + bi * bi) * (0.25 / NFFT);
        }
    }
    __syncthreads();
    TK();
    // ---- mel filterbank: lane s sums segment s = [seg[s], seg[s+1]) once with the ascending weights (filter
    // s + 1) and once with the descending ones (filter s); filter m = up(segment m - 1) + down(segment m) ----
    {
        double up0 = 0.0, up1 = 0.0, dn0 = 0.0, dn1 = 0.0;
        if (lane <= NFILT) {
            const int kb = a.t.seg[lane], ke = a.t.seg[lane + 1];
            for (int k = kb; k < ke; ++k) {
                const double gu = s_wup[k], gd = s_wdn[k], x0 = pw[k], x1 = pw[264 + k];
                up0 = fma(x0, gu, up0); dn0 = fma(x0, gd, dn0);
                up1 = fma(x1, gu, up1); dn1 = fma(x1, gd, dn1);
            }
        }
        const double pu0 = __shfl_up(up0, 1), pu1 = __shfl_up(up1, 1);   // ascending half lives one segment to the left
        if (lane >= 1 && lane <= NFILT) {
            double acc0 = pu0 + dn0, acc1 = pu1 + dn1;
            if (acc0 == 0.0) acc0 = 2.220446049250313e-16;  // np.finfo(float).eps
            if (acc1 == 0.0) acc1 = 2.220446049250313e-16;
            const double f0 = log10(acc0), f1 = log10(acc1);
            s_lfb[wv][0][lane - 1] = f0;
            s_lfb[wv][1][lane - 1] = f1;
            if (a.out_fb) {
                if (live[0]) a.out_fb[n0 * NFILT + lane - 1] = f0;
                if (live[1]) a.out_fb[(n0 + 1) * NFILT + lane - 1] = f1;
            }
        }
    }
    __syncthreads();
    TK();
    // ---- DCT-II (ortho), coefficients 1..13: lane = coefficient + 16 * quarter of the 40 filters ----
    {
        const int c = lane & 15, part = lane >> 4;
        double acc0 = 0.0, acc1 = 0.0;
        if (c < NCEPS) {
            const double* row = a.t.dct + c * NFILT + part * (NFILT / 4);
            const double* l0 = s_lfb[wv][0] + part * (NFILT / 4);
            const double* l1f = s_lfb[wv][1] + part * (NFILT / 4);
#pragma unroll
            for (int m = 0; m < NFILT / 4; ++m) { acc0 = fma(row[m], l0[m], acc0); acc1 = fma(row[m], l1f[m], acc1); }
        }
        acc0 += __shfl_xor(acc0, 16); acc0 += __shfl_xor(acc0, 32);
        acc1 += __shfl_xor(acc1, 16); acc1 += __shfl_xor(acc1, 32);
        if (lane < NCEPS && a.out_mfcc) {
            if (live[0]) a.out_mfcc[n0 * NCEPS + lane] = acc0;
            if (live[1]) a.out_mfcc[(n0 + 1) * NCEPS + lane] = acc1;
        }
    }
    TK();
#ifdef GH_MFCC_TIMING
    if (lane == 0 && a.out_fb && live[0]) for (int i = 0; i < 6; ++i) a.out_fb[n0 * NFILT + i] = (double)(tk[i] - tk[0]);
#endif
}

struct HostTables {
    std::vector<double> window, tw, wup, wdn, dct;
    std::vector<int> seg;
    std::vector<int32_t> f_utt;   // utterance of every frame (filled by check_inputs)
    int flen, fstep, pad_left;
};

// tables built the way the reference builds them (feature.py:25-40,52,58-75,80)
int build_tables(int sample_rate, double frame_size, double frame_stride, double low_freq, double high_freq,
                 HostTables& h) {
    h.flen = (int)(frame_size * sample_rate);
    h.fstep = (int)(frame_stride * sample_rate);
    GH_REQUIRE(sample_rate > 0 && h.flen >= 1 && h.fstep >= 1, "gh_mfcc: sample_rate=%d frame=%d step=%d samples",
               sample_rate, h.flen, h.fstep);
    int pad_w = 1;
    while (pad_w < h.flen) pad_w <<= 1;  // 1 << (width - 1).bit_length()
    if (pad_w > NFFT) {
        gh_set_error("gh_mfcc: frames of %d samples exceed the reference's NFFT = %d", h.flen, NFFT);
        return GH_ERR_UNSUPPORTED;
    }
    h.pad_left = (pad_w - h.flen) / 2;
    h.window.assign(NFFT, 0.0);
    for (int k = 0; k < pad_w; ++k)
        h.window[k] = pad_w == 1 ? 1.0 : 0.54 - 0.46 * std::cos(2.0 * M_PI * k / (pad_w - 1));
    h.tw.resize(2 * NFFT);
    for (int k = 0; k < NFFT; ++k) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / NFFT;
        h.tw[2 * k] = (double)cosl(ang);
        h.tw[2 * k + 1] = (double)sinl(ang);
    }
    if (!(high_freq > 0)) high_freq = sample_rate / 2.0;
    const double low_mel = 2595 * std::log10(1 + low_freq / 700), high_mel = 2595 * std::log10(1 + high_freq / 700);
    std::vector<double> bin(NFILT + 2);
    const double step = (high_mel - low_mel) / (NFILT + 1);   // np.linspace(start, stop, NFILT + 2)
    for (int i = 0; i < NFILT + 2; ++i) {
        const double mel = (i == NFILT + 1) ? high_mel : low_mel + step * i;
        const double hz = 700 * (std::pow(10.0, mel / 2595) - 1);
        bin[i] = std::floor((NFFT + 1) * hz / sample_rate);
    }
    // the triangles (feature.py:66-75) as per-bin weights: bin k in [bin[m-1], bin[m]) rises towards filter m,
    // bin k in [bin[m], bin[m+1]) falls away from filter m
    h.wup.assign(NBIN, 0.0);
    h.wdn.assign(NBIN, 0.0);
    h.seg.assign(NFILT + 2, 0);
    for (int i = 0; i < NFILT + 2; ++i) {
        GH_REQUIRE(bin[i] >= 0 && bin[i] <= NBIN && (i == 0 || bin[i] >= bin[i - 1]),
                   "gh_mfcc: mel point %d falls on bin %g outside the spectrum", i, bin[i]);
        h.seg[i] = (int)bin[i];
    }
    for (int m = 1; m <= NFILT; ++m) {
        const int lo = (int)bin[m - 1], ce = (int)bin[m], hi = (int)bin[m + 1];
        for (int k = lo; k < ce; ++k) h.wup[k] = (k - bin[m - 1]) / (bin[m] - bin[m - 1]);
        for (int k = ce; k < hi; ++k) h.wdn[k] = (bin[m + 1] - k) / (bin[m + 1] - bin[m]);
    }
    h.dct.resize((size_t)NCEPS * NFILT);
    for (int c = 1; c <= NCEPS; ++c)
        for (int m = 0; m < NFILT; ++m)
            h.dct[(size_t)(c - 1) * NFILT + m] = std::sqrt(2.0 / NFILT) * std::cos(M_PI * c * (2 * m + 1) / (2.0 * NFILT));
    return GH_OK;
}

void launch_mfcc(const MfccArgs& a, int64_t N, hipStream_t st) {
    const dim3 grid((unsigned)((N + 7) / 8)), block(256);
    if (a.fmt == 0) hipLaunchKernelGGL(mfcc_kernel<0>, grid, block, 0, st, a);
    else if (a.fmt == 1) hipLaunchKernelGGL(mfcc_kernel<1>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(mfcc_kernel<2>, grid, block, 0, st, a);
}

size_t fmt_size(int fmt) { return fmt == 0 ? 2 : (fmt == 1 ? 4 : 8); }

int check_inputs(const char* who, int fmt, int64_t U, const void* samples, const int64_t* s_off, const int64_t* f_off,
                 HostTables& h) {
    GH_REQUIRE(fmt >= 0 && fmt <= 2, "%s: sample_fmt=%d (0 int16, 1 float32, 2 float64)", who, fmt);
    GH_REQUIRE(U >= 0 && s_off && f_off && s_off[0] == 0 && f_off[0] == 0, "%s: offsets must start at 0", who);
    GH_REQUIRE(samples || s_off[U] == 0, "%s: samples is NULL", who);
    for (int64_t u = 0; u < U; ++u) {
        const int64_t len = s_off[u + 1] - s_off[u];
        GH_REQUIRE(len >= 1, "%s: utterance %lld is empty (the reference reads signal[0])", who, (long long)u);
        const int64_t nf = (len + h.fstep - 1) / h.fstep;
        GH_REQUIRE(f_off[u + 1] - f_off[u] == nf, "%s: frame_off gives utterance %lld %lld frames, ceil(%lld / %d) = %lld",
                   who, (long long)u, (long long)(f_off[u + 1] - f_off[u]), (long long)len, h.fstep, (long long)nf);
    }
    h.f_utt.resize((size_t)f_off[U]);
    for (int64_t u = 0; u < U; ++u) std::fill(h.f_utt.begin() + f_off[u], h.f_utt.begin() + f_off[u + 1], (int32_t)u);
    return GH_OK;
}

// carve the inputs + tables out of one block, upload, return the kernel arguments
size_t table_bytes(int fmt, int64_t n_samples, int64_t U, int64_t N) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    return al((size_t)n_samples * fmt_size(fmt)) + 2 * al((size_t)(U + 1) * 8) + al((size_t)N * 4) + al(NFFT * 8) + al(2 * NFFT * 8) +
           2 * al(NBIN * 8) + al((NFILT + 2) * 4) + al((size_t)NCEPS * NFILT * 8);
}

hipError_t upload_inputs(char* base, hipStream_t st, int fmt, int64_t U, const void* samples, const int64_t* s_off,
                         const int64_t* f_off, const HostTables& h, MfccArgs& a) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    char* p = base;
    hipError_t e = hipSuccess;
    auto put = [&](const void* src, size_t bytes) -> void* {
        void* dst = p;
        if (bytes && e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
        p += al(bytes);
        return dst;
    };
    a.pcm = put(samples, (size_t)s_off[U] * fmt_size(fmt));
    a.s_off = (const int64_t*)put(s_off, (size_t)(U + 1) * 8);
    a.f_off = (const int64_t*)put(f_off, (size_t)(U + 1) * 8);
    a.f_utt = (const int32_t*)put(h.f_utt.data(), h.f_utt.size() * 4);
    a.t.window = (const double*)put(h.window.data(), NFFT * 8);
    a.t.tw = (const double*)put(h.tw.data(), 2 * NFFT * 8);
    a.t.wup = (const double*)put(h.wup.data(), NBIN * 8);
    a.t.wdn = (const double*)put(h.wdn.data(), NBIN * 8);
    a.t.seg = (const int*)put(h.seg.data(), (NFILT + 2) * 4);
    a.t.dct = (const double*)put(h.dct.data(), (size_t)NCEPS * NFILT * 8);
    a.fmt = fmt; a.U = U; a.N = f_off[U];
    a.flen = h.flen; a.fstep = h.fstep; a.pad_left = h.pad_left;
    return e;
}

}  // namespace

extern "C" int64_t gh_mfcc_frames(int64_t n_samples, int sample_rate, double frame_stride) {
    const int64_t step = (int64_t)(frame_stride * sample_rate);
    if (n_samples < 1 || step < 1) return -1;
    return (n_samples + step - 1) / step;  // math.ceil(slen / frame_step1), feature.py:13
}

extern "C" int gh_mfcc(gh_ctx* ctx, int sample_fmt, int sample_rate, double frame_size, double frame_stride,
                       double low_freq, double high_freq, int64_t U, const void* samples, const int64_t* sample_off,
                       const int64_t* frame_off, double* out_fbank, double* out_mfcc) {
    GH_REQUIRE(ctx, "gh_mfcc: ctx is NULL");
    HostTables h;
    int rc = build_tables(sample_rate, frame_size, frame_stride, low_freq, high_freq, h);
    if (rc) return rc;
    if ((rc = check_inputs("gh_mfcc", sample_fmt, U, samples, sample_off, frame_off, h))) return rc;
    const int64_t N = frame_off[U];
    if (N == 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    char* d_in;
    double *d_fb, *d_mf;
    Carver cv;
    cv.add(&d_in, table_bytes(sample_fmt, sample_off[U], U, N));
    cv.add(&d_fb, (size_t)N * NFILT);
    cv.add(&d_mf, (size_t)N * NCEPS);
    if ((rc = cv.commit(ctx))) return rc;
    hipStream_t st = ctx->stream;
    MfccArgs a;
    GH_HIP(upload_inputs(d_in, st, sample_fmt, U, samples, sample_off, frame_off, h, a));
    a.out_fb = d_fb;
    a.out_mfcc = d_mf;
    launch_mfcc(a, N, st);
    GH_HIP(hipGetLastError());
    if (out_fbank) GH_HIP(hipMemcpyAsync(out_fbank, d_fb, (size_t)N * NFILT * 8, hipMemcpyDeviceToHost, st));
    if (out_mfcc) GH_HIP(hipMemcpyAsync(out_mfcc, d_mf, (size_t)N * NCEPS * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

extern "C" int gh_batch_create_from_pcm(gh_ctx* ctx, gh_dtype dtype, int mode, int sample_fmt, int sample_rate,
                                        double frame_size, double frame_stride, double low_freq, double high_freq,
                                        int64_t U, const void* samples, const int64_t* sample_off,
                                        const int64_t* frame_off, gh_batch** out) {
    GH_REQUIRE(ctx && out, "gh_batch_create_from_pcm: NULL argument");
    HostTables h;
    int rc = build_tables(sample_rate, frame_size, frame_stride, low_freq, high_freq, h);
    if (rc) return rc;
    if ((rc = check_inputs("gh_batch_create_from_pcm", sample_fmt, U, samples, sample_off, frame_off, h))) return rc;
    const int64_t N = frame_off[U];
    void* extra = nullptr;
    return gh_batch_from_device_cepstra(
        ctx, dtype, mode, NCEPS, N, U, frame_off, table_bytes(sample_fmt, sample_off[U], U, N), &extra,
        [&](double* d_ceps, hipStream_t st) {
            MfccArgs a;
            hipError_t e = upload_inputs(static_cast<char*>(extra), st, sample_fmt, U, samples, sample_off, frame_off, h, a);
            if (e != hipSuccess) return e;
            a.out_fb = nullptr;
            a.out_mfcc = d_ceps;
            launch_mfcc(a, N, st);
            return hipGetLastError();
        },
        "gh_batch_create_from_pcm", out);
}
