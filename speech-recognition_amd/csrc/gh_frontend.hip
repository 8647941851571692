// Feature stacking in front of the hot path: cepstra [T,C] ->
// [cepstra | delta | delta-delta] -> per-utterance standardisation, written straight into a
// resident gh_batch -- no host round trip between the front-end and the likelihood kernel.
// Reference: delta_feature (sr/core.py:13-22: central difference, one-sided at both ends),
// standardize (sr/feature/feature.py:85-88: (x - mean) / std per column, population std), as
// composed by load_wav_as_mfcc (sr/core.py:41-44).  fp64 arithmetic, HBM-bound.
#include "gh_internal.h"
#include "gh_host.h"
#include <functional>

namespace {

__device__ __forceinline__ double delta_at(const double* __restrict__ f, int t, int T, int C, int c) {
    // f: [T,C] of one utterance
    if (t == 0) return f[(int64_t)C + c] - f[c];
    if (t == T - 1) return f[(int64_t)t * C + c] - f[(int64_t)(t - 1) * C + c];
    return f[(int64_t)(t + 1) * C + c] - f[(int64_t)(t - 1) * C + c];
}

// one workgroup per utterance; raw: [N,3C] fp64 scratch
__global__ __launch_bounds__(256) void stack_kernel(const double* __restrict__ ceps, const int64_t* __restrict__ off, int C,
                                                    double* __restrict__ raw) {
    const int64_t f0 = off[blockIdx.x];
    const int T = (int)(off[blockIdx.x + 1] - f0);
    if (T < 2) return;
    const double* f = ceps + f0 * C;
    double* o = raw + f0 * 3 * C;
    for (int i = threadIdx.x; i < T * C; i += blockDim.x) {
        const int t = i / C, c = i % C;
        const double d = delta_at(f, t, T, C, c);
        double dd;  // delta of the delta track, same edge rules
        if (t == 0) dd = delta_at(f, 1, T, C, c) - d;
        else if (t == T - 1) dd = d - delta_at(f, t - 1, T, C, c);
        else dd = delta_at(f, t + 1, T, C, c) - delta_at(f, t - 1, T, C, c);
        o[(int64_t)t * 3 * C + c] = f[(int64_t)t * C + c];
        o[(int64_t)t * 3 * C + C + c] = d;
        o[(int64_t)t * 3 * C + 2 * C + c] = dd;
    }
}

// one workgroup (4 waves) per utterance: lane = column, wave = frame phase; two-pass mean / std
template <typename OT>
__global__ __launch_bounds__(256) void standardize_kernel(const double* __restrict__ raw, const int64_t* __restrict__ off,
                                                          int D3, OT* __restrict__ out) {
    __shared__ double part[4][64];
    __shared__ double s_mean[64], s_std[64];
    const int64_t f0 = off[blockIdx.x];
    const int T = (int)(off[blockIdx.x + 1] - f0);
    if (T <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double* x = raw + f0 * D3;
    OT* o = out + f0 * D3;
    for (int d0 = 0; d0 < D3; d0 += 64) {
        const int d = d0 + lane;
        const bool act = d < D3;
        double s = 0;
        if (act) for (int t = wv; t < T; t += 4) s += x[(int64_t)t * D3 + d];
        part[wv][lane] = s;
        __syncthreads();
        if (wv == 0) s_mean[lane] = (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]) / T;
        __syncthreads();
        const double mu = s_mean[lane];
        s = 0;
        if (act) for (int t = wv; t < T; t += 4) { const double c = x[(int64_t)t * D3 + d] - mu; s = fma(c, c, s); }
        part[wv][lane] = s;
        __syncthreads();
        if (wv == 0) s_std[lane] = sqrt((part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]) / T);
        __syncthreads();
        const double sd = s_std[lane];
        if (act) for (int t = wv; t < T; t += 4) o[(int64_t)t * D3 + d] = (OT)((x[(int64_t)t * D3 + d] - mu) / sd);
        __syncthreads();
    }
}

template <typename OT>
__global__ void convert_kernel(const double* __restrict__ in, int64_t n, OT* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (OT)in[i];
}

}  // namespace

// tail shared by the cepstra and the PCM entry points: d_ceps [N,C] (device) -> resident batch.
// `fill(d_ceps, stream)` enqueues whatever produces the cepstra (an upload or the MFCC kernel).
int gh_batch_from_device_cepstra(gh_ctx* ctx, gh_dtype dtype, int mode, int C, int64_t N, int64_t U,
                                 const int64_t* utt_offsets, size_t extra_scratch, void** extra,
                                 const std::function<hipError_t(double*, hipStream_t)>& fill, const char* who,
                                 gh_batch** out) {
    GH_REQUIRE(C > 0, "%s: C=%d", who, C);
    GH_REQUIRE(mode >= 0 && mode <= 2, "%s: mode=%d", who, mode);
    for (int64_t u = 0; mode != 2 && u < U; ++u)
        GH_REQUIRE(utt_offsets[u + 1] - utt_offsets[u] >= 2,
                   "%s: utterance %lld has fewer than 2 frames (delta_feature indexes feat[i + 1])", who, (long long)u);
    GH_HIP(hipSetDevice(ctx->device));
    const int D3 = mode == 2 ? C : 3 * C;  // mode 2 standardises the C input columns as they are
    const size_t esz = dtype == GH_F64 ? 8 : 4;
    void* feats = nullptr;
    if (N > 0) GH_HIP(hipMalloc(&feats, (size_t)N * D3 * esz));
    gh_batch* b = nullptr;
    int rc = gh_batch_wrap(ctx, dtype, D3, N, U, feats, utt_offsets, &b);
    if (rc) { if (feats) hipFree(feats); return rc; }
    b->feats = feats;
    b->owns_feats = true;
    if (N > 0) {
        double *d_ceps, *d_raw;
        char* d_extra = nullptr;
        Carver cv;
        cv.add(&d_ceps, (size_t)N * C);
        cv.add(&d_raw, (size_t)N * D3);
        if (extra_scratch) cv.add(&d_extra, extra_scratch);
        rc = cv.commit(ctx);
        if (rc) { gh_batch_destroy(b); return rc; }
        if (extra) *extra = d_extra;
        hipStream_t st = ctx->stream;
        hipError_t e = fill(d_ceps, st);
        if (e == hipSuccess) {
            const double* src = d_ceps;  // what gets standardised / copied
            if (mode != 2) {
                hipLaunchKernelGGL(stack_kernel, dim3((unsigned)U), dim3(256), 0, st, d_ceps, b->d_offsets, C, d_raw);
                src = d_raw;
            }
            if (mode == 1) {  // stack only: raw [ceps | delta | delta-delta]
                if (dtype == GH_F64) e = hipMemcpyAsync(feats, src, (size_t)N * D3 * 8, hipMemcpyDeviceToDevice, st);
                else hipLaunchKernelGGL((convert_kernel<float>), dim3((unsigned)(((size_t)N * D3 + 255) / 256)), dim3(256), 0, st, src, (int64_t)N * D3, (float*)feats);
            } else if (dtype == GH_F64) {
                hipLaunchKernelGGL((standardize_kernel<double>), dim3((unsigned)U), dim3(256), 0, st, src, b->d_offsets, D3, (double*)feats);
            } else {
                hipLaunchKernelGGL((standardize_kernel<float>), dim3((unsigned)U), dim3(256), 0, st, src, b->d_offsets, D3, (float*)feats);
            }
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            gh_set_error("%s: %s", who, hipGetErrorString(e));
            gh_batch_destroy(b);
            return GH_ERR_HIP;
        }
    }
    *out = b;
    return GH_OK;
}

extern "C" int gh_batch_create_from_cepstra(gh_ctx* ctx, gh_dtype dtype, int mode, int C, int64_t N, int64_t U,
                                            const double* ceps_host, const int64_t* utt_offsets, gh_batch** out) {
    GH_REQUIRE(ctx && out && utt_offsets && (ceps_host || N == 0), "gh_batch_create_from_cepstra: NULL argument");
    return gh_batch_from_device_cepstra(
        ctx, dtype, mode, C, N, U, utt_offsets, 0, nullptr,
        [&](double* d_ceps, hipStream_t st) {
            return hipMemcpyAsync(d_ceps, ceps_host, (size_t)N * C * 8, hipMemcpyHostToDevice, st);
        },
        "gh_batch_create_from_cepstra", out);
}

extern "C" int gh_batch_fetch_features(gh_ctx* ctx, const gh_batch* b, void* out_host) {
    GH_REQUIRE(ctx && b && out_host, "gh_batch_fetch_features: NULL argument");
    GH_HIP(hipSetDevice(ctx->device));
    if (b->N == 0) return GH_OK;
    GH_HIP(hipMemcpyAsync(out_host, b->feats, (size_t)b->N * b->D * (b->dtype == GH_F64 ? 8 : 4), hipMemcpyDeviceToHost,
                          ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}
