// One tap of CorrBlock1D.__call__ (/root/reference/models/core/corr.py:74-94 with bilinear_sampler :8-21): linear interpolation of a
// pyramid row at (kk - 4) + xs / 2^lvl, zero padding, align_corners=True -- shared by the lookup kernels (corr.hip) and the fused
// correlation-encoder chain that looks its taps up itself (pwchain.hip).
#pragma once

__device__ __forceinline__ float lookup_tap(const float* __restrict__ L, int Wl, float xs, int lvl, int kk) {
    // same fp32 op sequence as the reference: normalise (corr.py:14,85) then grid_sample's un-normalise
    const float pos = (float)(kk - 4) + xs / (float)(1 << lvl);
    const float wm1 = (float)(Wl - 1);
    const float g = 2.0f * pos / wm1 - 1.0f;
    const float p = ((g + 1.0f) / 2.0f) * wm1;
    const float pf = floorf(p);
    const float a = p - pf;
    // far out-of-range positions (|p| beyond int range) contribute nothing
    if (!(pf >= -1.0f && pf <= (float)Wl)) return 0.0f;
    const int i0 = (int)pf, i1 = i0 + 1;
    const float v0 = (i0 >= 0 && i0 < Wl) ? L[i0] : 0.0f;
    const float v1 = (i1 >= 0 && i1 < Wl) ? L[i1] : 0.0f;
    return (1.0f - a) * v0 + a * v1;
}
