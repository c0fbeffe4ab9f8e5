// conv_res2.h — the first ResnetBlock's shortcut recomputed inside its consumer (Conv3P::res2_*; see kernels.h).
// Bit-for-bit the arithmetic of first_conv_kernel's 1x1 branch (unet_elem.hip): a = b1; a = fma(plane_q * mask, w1[q], a), q = 0..planes-1,
// with plane 1 pre-multiplied by c_in.  The 4 x 64 coefficient table (bias, three weight rows) lives in LDS - as registers
// (33 per thread) it pushed the fused-tail kernels over their occupancy step.  `at(row, c)` returns the float4 of channels
// c..c+3 of table row `row` (0 bias, 1..3 weights); `put(row, c)` the address of one entry.
// No include guard on purpose: included once per namespace build of the two conv files.
template <class Put>
__device__ __forceinline__ void res2_fill(const Conv3P& p, int tid, Put put) {
    if (tid < 64) {
        *put(0, tid) = p.res2_b[tid];
#pragma unroll
        for (int q = 0; q < 3; ++q) *put(1 + q, tid) = q < p.res2_planes ? p.res2_w[q * 64 + tid] : 0.f;
    }
}
// plane values of one pixel -> the 8 shortcut channels c8..c8+7 of this thread
template <class At>
__device__ __forceinline__ void res2_eval(At at, int c8, int planes, float c_in, float mu, float x, float spk, float mk, float (&out)[8]) {
    const float v0 = mu * mk;
    const float t1 = x * c_in;
    const float v1 = t1 * mk;
    const float v2 = spk * mk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 b = at(0, c8 + 4 * h), w0 = at(1, c8 + 4 * h), w1 = at(2, c8 + 4 * h);
        float a0 = fmaf(v1, w1.x, fmaf(v0, w0.x, b.x)), a1 = fmaf(v1, w1.y, fmaf(v0, w0.y, b.y));
        float a2 = fmaf(v1, w1.z, fmaf(v0, w0.z, b.z)), a3 = fmaf(v1, w1.w, fmaf(v0, w0.w, b.w));
        if (planes == 3) {
            const float4 w2 = at(3, c8 + 4 * h);
            a0 = fmaf(v2, w2.x, a0); a1 = fmaf(v2, w2.y, a1); a2 = fmaf(v2, w2.z, a2); a3 = fmaf(v2, w2.w, a3);
        }
        out[4 * h] = a0; out[4 * h + 1] = a1; out[4 * h + 2] = a2; out[4 * h + 3] = a3;
    }
}
