// mel.hip — STFT/mel front-end helpers (audio/stft.py:52-81,159-178; audio/tools.py:8-15).  The windowed-DFT
// contraction itself (frames x 1026 x 1024) runs on the shared implicit-GEMM kernel over an overlapping-row
// view of the padded waveform (row stride = hop); this file holds the pad/clip and magnitude/mel/log tails.
#include "kernels.h"

namespace dex {

__global__ void wav_pad_kernel(const float* wav, int n, int pad, float* out, int out_len, long out_bstride) {
    wav += (long)blockIdx.y * n; out += (long)blockIdx.y * out_bstride;           // blockIdx.y = utterance
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < out_len; i += gridDim.x * blockDim.x) {
        int j = i - pad;
        float v = 0.f;
        if (i < n + 2 * pad) {
            if (j < 0) j = -j;                       // reflect (no edge repeat), F.pad mode='reflect'
            if (j >= n) j = 2 * (n - 1) - j;
            v = fminf(1.f, fmaxf(-1.f, wav[j]));
        }
        out[i] = v;
    }
}
void launch_wav_pad(const float* wav, int n, int pad, float* out, int out_len, hipStream_t st, int B, long out_bstride) {
    int blocks = (out_len + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wav_pad_kernel, dim3(blocks, B), dim3(256), 0, st, wav, n, pad, out, out_len, out_bstride);
}

// one block per frame: magnitudes into LDS, then 80 mel rows (one wave-strided dot each) and the energy.
__global__ __launch_bounds__(256) void magmel_kernel(const MagMelP p) {
    __shared__ float mag[544];
    __shared__ float esum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;                                                    // utterance
    const float* row = p.spec + ((long)b * p.frames + f) * p.ld;
    float* mel = p.mel + (long)b * p.nmel * p.frames;
    float* energy = p.energy + (long)b * p.frames;
    float e = 0.f;
    for (int k = tid; k < p.nbins; k += 256) {
        const float re = row[k], im = row[p.im_off + k];
        const float m = sqrtf(re * re + im * im);
        mag[k] = m;
        e = fmaf(m, m, e);
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (lane == 0) esum[wave] = e;
    __syncthreads();
    if (tid == 0) energy[f] = sqrtf((esum[0] + esum[1]) + (esum[2] + esum[3]));
    for (int j = wave; j < p.nmel; j += 4) {
        const float* w = p.melW + (long)j * p.nbins;
        float a = 0.f;
        for (int k = lane; k < p.nbins; k += 64) a = fmaf(w[k], mag[k], a);
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) mel[(long)j * p.frames + f] = logf(fmaxf(a, 1e-5f));
    }
}
void launch_magmel(const MagMelP& p, hipStream_t st, int B) {
    hipLaunchKernelGGL(magmel_kernel, dim3(p.frames, B), dim3(256), 0, st, p);
}

// log-f0 normalisation of the DEX style front-end (DEX-TTS/synthesize.py:26-38,55-58): lf0 = log(f0) where f0 != 0, then over the
// entries with lf0 != 0 (an f0 of exactly 1 Hz counts as unvoiced, as in the reference): (lf0 - mean) / (std + 1e-8), or lf0 - mean
// when std == 0; unvoiced entries and everything past the utterance's length stay 0.
// The statistics repeat numpy's fp32 arithmetic OPERATION BY OPERATION — np.mean / np.std of a contiguous float32 array reduce as
// 0 + pairwise_sum(a) with numpy's 8-accumulator / 128-element-block pairwise scheme (checked against np.add.reduce), the variance from the separately
// rounded (a - mean)^2 — because on a (nearly) constant pitch track the reference's std is nothing BUT that round-off (a constant
// 200 Hz track normalises to -0.979, not 0), so only the same operation order reproduces it.  One wave per utterance: the voiced
// values are compacted in order into LDS by ballot, lane 0 runs the two reductions (T is a few hundred frames), all lanes write.
__device__ float np_pairwise_sum_f32(const float* a, int n) {
    if (n < 8) {
        float r = -0.0f;
        for (int i = 0; i < n; ++i) r = __fadd_rn(r, a[i]);
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, a[i]);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_sum_f32(a, n2), np_pairwise_sum_f32(a + n2, n - n2));
}
__device__ float np_add_reduce_f32(const float* a, int n) {       // np.add.reduce of a contiguous array: identity (0) + pairwise_sum(a, n)
    return __fadd_rn(0.f, np_pairwise_sum_f32(a, n));
}

__global__ __launch_bounds__(64) void lf0_normalize_kernel(const float* __restrict__ f0, const int* __restrict__ lengths, int T, float* __restrict__ out) {
    extern __shared__ float lf0_buf[];                      // [T] compacted voiced log-f0, then their squared deviations
    __shared__ float stat[2];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int len = lengths ? min(max(lengths[b], 0), T) : T;
    f0 += (long)b * T; out += (long)b * T;
    // log through fp64, rounded once: the correctly rounded fp32 logarithm (what numpy / torch return for these inputs; the fp32
    // library logf is one ulp off for some, and on a constant track one ulp moves the normalised value by 1 %)
    auto lf = [&](int i) { const float v = f0[i]; return v != 0.f ? (float)log((double)v) : 0.f; };
    int c = 0;
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        const float v = i < len ? lf(i) : 0.f;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(v != 0.f);
        if (v != 0.f) lf0_buf[c + __popcll(m & ((1ull << lane) - 1ull))] = v;
        c += __popcll(m);
    }
    __syncthreads();
    if (lane == 0 && c > 0) {
        const float fc = (float)c;
        const float mean = __fdiv_rn(np_add_reduce_f32(lf0_buf, c), fc);
        for (int i = 0; i < c; ++i) { const float d = __fsub_rn(lf0_buf[i], mean); lf0_buf[i] = __fmul_rn(d, d); }
        stat[0] = mean;
        stat[1] = __fsqrt_rn(__fdiv_rn(np_add_reduce_f32(lf0_buf, c), fc));
    }
    __syncthreads();
    const float mean = stat[0], sd = stat[1];
    const float den = __fadd_rn(sd, 1e-8f);
    for (int i = lane; i < T; i += 64) {
        float v = i < len ? lf(i) : 0.f;
        if (v != 0.f && c > 0) v = sd == 0.f ? __fsub_rn(v, mean) : __fdiv_rn(__fsub_rn(v, mean), den);
        out[i] = v;
    }
}
void launch_lf0_normalize(const float* f0, const int* lengths, int B, int T, float* out, hipStream_t st) {
    hipLaunchKernelGGL(lf0_normalize_kernel, dim3(B), dim3(64), (size_t)T * sizeof(float), st, f0, lengths, T, out);
}

}  // namespace dex
