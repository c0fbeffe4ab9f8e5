// mel.hip — STFT/mel front-end helpers (audio/stft.py:52-81,159-178; audio/tools.py:8-15).  The windowed-DFT
// contraction itself (frames x 1026 x 1024) runs on the shared implicit-GEMM kernel over an overlapping-row
// view of the padded waveform (row stride = hop); this file holds the pad/clip and magnitude/mel/log tails.
#include "kernels.h"

namespace dex {

__global__ void wav_pad_kernel(const float* wav, int n, int pad, float* out, int out_len) {
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
void launch_wav_pad(const float* wav, int n, int pad, float* out, int out_len, hipStream_t st) {
    int blocks = (out_len + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wav_pad_kernel, dim3(blocks), dim3(256), 0, st, wav, n, pad, out, out_len);
}

// one block per frame: magnitudes into LDS, then 80 mel rows (one wave-strided dot each) and the energy.
__global__ __launch_bounds__(256) void magmel_kernel(const MagMelP p) {
    __shared__ float mag[544];
    __shared__ float esum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = p.spec + (long)f * p.ld;
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
    if (tid == 0) p.energy[f] = sqrtf((esum[0] + esum[1]) + (esum[2] + esum[3]));
    for (int j = wave; j < p.nmel; j += 4) {
        const float* w = p.melW + (long)j * p.nbins;
        float a = 0.f;
        for (int k = lane; k < p.nbins; k += 64) a = fmaf(w[k], mag[k], a);
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) p.mel[(long)j * p.frames + f] = logf(fmaxf(a, 1e-5f));
    }
}
void launch_magmel(const MagMelP& p, hipStream_t st) {
    hipLaunchKernelGGL(magmel_kernel, dim3(p.frames), dim3(256), 0, st, p);
}

}  // namespace dex
