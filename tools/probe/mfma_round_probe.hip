// How does v_mfma_f32_32x32x16_bf16 round?  Each case is a 16-term dot product of bf16 values plus an fp32 addend; every row of A
// and every column of B hold the same vector, so every element of D is that dot product.  Prints the fp32 result beside the exactly
// rounded (RNE) and the truncated (toward zero) fp64 sum.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_round_probe.hip -o /tmp/mfma_round_probe && /tmp/mfma_round_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Case {
    float a[16], b[16], c;
};

__global__ void probe(const Case* cs, float* out, int n, int chain) {
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < n; ++i) {
        bf16x8 av, bv;
        for (int j = 0; j < 8; ++j) {
            av[j] = (bf16_t)cs[i].a[8 * (lane >> 5) + j];
            bv[j] = (bf16_t)cs[i].b[8 * (lane >> 5) + j];
        }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = cs[i].c;
        for (int s = 0; s < chain; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        if (lane == 0) out[i] = acc[0];
    }
}

static float trunc_f32(double x) {          // toward zero to fp32
    float f = (float)x;
    if (fabs((double)f) > fabs(x)) f = nextafterf(f, 0.0f);
    return f;
}

int main() {
    const float u = ldexpf(1.0f, -23);       // ulp(1.0)
    Case cs[16];
    memset(cs, 0, sizeof(cs));
    int n = 0;
    auto one = [&](float c, float x, int terms) {      // c + terms * x, x as 1 * x
        for (int k = 0; k < 16; ++k) {
            cs[n].a[k] = k < terms ? 1.0f : 0.0f;
            cs[n].b[k] = k < terms ? x : 0.0f;
        }
        cs[n].c = c;
        ++n;
    };
    one(1.0f, 0.75f * u, 1);        // RNE: 1 + u, truncation: 1
    one(1.0f, 0.25f * u, 1);        // RNE: 1
    one(-1.0f, -0.75f * u, 1);      // RNE: -(1 + u); toward zero: -1
    one(-1.0f, 0.75f * u, 1);       // exact -1 + 0.75u: RNE -(1 - 0.5 u) ... (below 1 the ulp halves)
    one(1.0f, 0.125f * u, 16);      // 16 terms of u/8 = 2u: exact 1 + 2u; per-term truncation gives 1
    one(1.0f, 0.0625f * u, 16);     // 16 terms of u/16 = 1u
    one(1.0f, 0.03125f * u, 16);    // 0.5 u: tie -> even = 1
    one(1.0f, 0.046875f * u, 16);   // 0.75 u
    one(0.0f, 1.0f + 0.0078125f, 2);  // 2 * 1.0078125 exact
    one(1024.0f, 0.75f * u, 16);    // small terms far below the addend
    one(1.0f, 0.75f * u, 1);        // (used with chain > 1 below)
    Case* d;
    float* o;
    hipMalloc(&d, sizeof(cs));
    hipMalloc(&o, 64 * sizeof(float));
    hipMemcpy(d, cs, sizeof(cs), hipMemcpyHostToDevice);
    for (int chain : {1, 8}) {
        probe<<<1, 64>>>(d, o, n, chain);
        float h[64];
        hipMemcpy(h, o, 64 * sizeof(float), hipMemcpyDeviceToHost);
        printf("chain of %d MFMAs onto the same accumulator:\n", chain);
        for (int i = 0; i < n; ++i) {
            double ex = cs[i].c;
            for (int s = 0; s < chain; ++s)
                for (int k = 0; k < 16; ++k) ex += (double)cs[i].a[k] * (double)cs[i].b[k];
            const float rne = (float)ex, tz = trunc_f32(ex);
            printf("  case %2d: got %.9g (%a)   exact %.12g   RNE(once) %a   toward-zero(once) %a   %s\n", i, h[i], h[i], ex, rne, tz,
                   h[i] == rne ? "== RNE of the exact sum" : (h[i] == tz ? "== truncation of the exact sum" : "neither"));
        }
    }
    return 0;
}
