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
    const double* tw;         // [NFFT/2][2] cos / -sin of 2 pi k / NFFT
    const double* fbank;      // [NFILT][NBIN]
    const int* flo;           // [NFILT] first bin with a non-zero weight
    const int* fhi;           // [NFILT] one past the last
    const double* dct;        // [NCEPS][NFILT]
};

struct MfccArgs {
    const void* pcm; int fmt;
    const int64_t* s_off; const int64_t* f_off;
    int64_t U, N;
    int flen, fstep, pad_left;
    MfccTables t;
    double* out_fb; double* out_mfcc;
};

__device__ __forceinline__ double sample_at(const void* pcm, int fmt, int64_t i) {
    if (fmt == 0) return (double)static_cast<const int16_t*>(pcm)[i];
    if (fmt == 1) return (double)static_cast<const float*>(pcm)[i];
    return static_cast<const double*>(pcm)[i];
}

__global__ __launch_bounds__(256) void mfcc_kernel(MfccArgs a) {
    __shared__ double s_re[4][NFFT], s_im[4][NFFT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 4 + wv;   // frame index over the whole batch
    const bool live = n < a.N;                        // dead waves still take part in the barriers
    double* re = s_re[wv];
    double* im = s_im[wv];
    int64_t s0 = 0, slen = 0, sbase = 0;
    if (live) {  // utterance of this frame: last u with f_off[u] <= n
        int64_t lo = 0, hi = a.U - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (a.f_off[mid] <= n) lo = mid; else hi = mid - 1;
        }
        sbase = a.s_off[lo];
        slen = a.s_off[lo + 1] - sbase;
        s0 = (n - a.f_off[lo]) * a.fstep;
    }
    // ---- windowed, zero padded frame into bit-reversed order ----
#pragma unroll
    for (int i = 0; i < NFFT / 64; ++i) {
        const int k = lane + 64 * i;
        const int j = k - a.pad_left;
        double v = 0.0;
        if (live && j >= 0 && j < a.flen && s0 + j < slen) {
            const int64_t p = sbase + s0 + j;
            const double cur = sample_at(a.pcm, a.fmt, p);
            v = (s0 + j == 0) ? cur : __dsub_rn(cur, __dmul_rn(0.97, sample_at(a.pcm, a.fmt, p - 1)));
            v *= a.t.window[k];
        }
        const int r = (int)(__brev((unsigned)k) >> (32 - 9));
        re[r] = v;
        im[r] = 0.0;
    }
    __syncthreads();
    // ---- 9 radix-2 stages, 4 butterflies per lane and stage ----
    for (int s = 0; s < 9; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int i = 0; i < NFFT / 128; ++i) {
            const int b = lane + 64 * i;
            const int pos = b & (half - 1);
            const int i0 = ((b >> s) << (s + 1)) + pos, i1 = i0 + half;
            const int ti = pos << (8 - s);
            const double wr = a.t.tw[2 * ti], wi = a.t.tw[2 * ti + 1];
            const double xr = re[i1], xi = im[i1];
            const double tr = wr * xr - wi * xi, tim = wr * xi + wi * xr;
            const double ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + tim;
            re[i1] = ur - tr; im[i1] = ui - tim;
        }
        __syncthreads();
    }
    // ---- power spectrum (bins 0..256) ----
    double pw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int k = lane + 64 * i;
        pw[i] = (k < NBIN) ? (re[k] * re[k] + im[k] * im[k]) * (1.0 / NFFT) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int k = lane + 64 * i;
        if (k < NBIN) re[k] = pw[i];
    }
    __syncthreads();
    // ---- mel filterbank, log10 ----
    if (lane < NFILT) {
        const double* wgt = a.t.fbank + lane * NBIN;
        double acc = 0.0;
        for (int k = a.t.flo[lane]; k < a.t.fhi[lane]; ++k) acc = fma(re[k], wgt[k], acc);
        if (acc == 0.0) acc = 2.220446049250313e-16;  // np.finfo(float).eps
        const double lf = log10(acc);
        im[lane] = lf;
        if (live && a.out_fb) a.out_fb[n * NFILT + lane] = lf;
    }
    __syncthreads();
    // ---- DCT-II (ortho), coefficients 1..13 ----
    if (live && lane < NCEPS && a.out_mfcc) {
        const double* row = a.t.dct + lane * NFILT;
        double acc = 0.0;
        for (int m = 0; m < NFILT; ++m) acc = fma(row[m], im[m], acc);
        a.out_mfcc[n * NCEPS + lane] = acc;
    }
}

struct HostTables {
    std::vector<double> window, tw, fbank, dct;
    std::vector<int> flo, fhi;
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
    h.tw.resize(NFFT);
    for (int k = 0; k < NFFT / 2; ++k) {
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
    h.fbank.assign((size_t)NFILT * NBIN, 0.0);
    h.flo.assign(NFILT, 0);
    h.fhi.assign(NFILT, 0);
    for (int m = 1; m <= NFILT; ++m) {
        const int lo = (int)bin[m - 1], ce = (int)bin[m], hi = (int)bin[m + 1];
        GH_REQUIRE(lo >= 0 && hi <= NBIN, "gh_mfcc: mel filter %d covers bins [%d, %d) outside the spectrum", m, lo, hi);
        double* row = h.fbank.data() + (size_t)(m - 1) * NBIN;
        for (int k = lo; k < ce; ++k) row[k] = (k - bin[m - 1]) / (bin[m] - bin[m - 1]);
        for (int k = ce; k < hi; ++k) row[k] = (bin[m + 1] - k) / (bin[m + 1] - bin[m]);
        h.flo[m - 1] = lo;
        h.fhi[m - 1] = std::max(hi, lo);
    }
    h.dct.resize((size_t)NCEPS * NFILT);
    for (int c = 1; c <= NCEPS; ++c)
        for (int m = 0; m < NFILT; ++m)
            h.dct[(size_t)(c - 1) * NFILT + m] = std::sqrt(2.0 / NFILT) * std::cos(M_PI * c * (2 * m + 1) / (2.0 * NFILT));
    return GH_OK;
}

size_t fmt_size(int fmt) { return fmt == 0 ? 2 : (fmt == 1 ? 4 : 8); }

int check_inputs(const char* who, int fmt, int64_t U, const void* samples, const int64_t* s_off, const int64_t* f_off,
                 const HostTables& h) {
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
    return GH_OK;
}

// carve the inputs + tables out of one block, upload, return the kernel arguments
size_t table_bytes(int fmt, int64_t n_samples, int64_t U) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    return al((size_t)n_samples * fmt_size(fmt)) + 2 * al((size_t)(U + 1) * 8) + al(NFFT * 8) + al(NFFT * 8) +
           al((size_t)NFILT * NBIN * 8) + 2 * al(NFILT * 4) + al((size_t)NCEPS * NFILT * 8);
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
    a.t.window = (const double*)put(h.window.data(), NFFT * 8);
    a.t.tw = (const double*)put(h.tw.data(), NFFT * 8);
    a.t.fbank = (const double*)put(h.fbank.data(), (size_t)NFILT * NBIN * 8);
    a.t.flo = (const int*)put(h.flo.data(), NFILT * 4);
    a.t.fhi = (const int*)put(h.fhi.data(), NFILT * 4);
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
    cv.add(&d_in, table_bytes(sample_fmt, sample_off[U], U));
    cv.add(&d_fb, (size_t)N * NFILT);
    cv.add(&d_mf, (size_t)N * NCEPS);
    if ((rc = cv.commit(ctx))) return rc;
    hipStream_t st = ctx->stream;
    MfccArgs a;
    GH_HIP(upload_inputs(d_in, st, sample_fmt, U, samples, sample_off, frame_off, h, a));
    a.out_fb = d_fb;
    a.out_mfcc = d_mf;
    hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, a);
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
        ctx, dtype, mode, NCEPS, N, U, frame_off, table_bytes(sample_fmt, sample_off[U], U), &extra,
        [&](double* d_ceps, hipStream_t st) {
            MfccArgs a;
            hipError_t e = upload_inputs(static_cast<char*>(extra), st, sample_fmt, U, samples, sample_off, frame_off, h, a);
            if (e != hipSuccess) return e;
            a.out_fb = nullptr;
            a.out_mfcc = d_ceps;
            hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, a);
            return hipGetLastError();
        },
        "gh_batch_create_from_pcm", out);
}
