// Fused chain of per-pixel (1x1) layers of PCBlock4_Deep_nopool_res (/root/reference/models/core/ppmtereo_update.py:1024-1030):
//   chain A:  x1 = gelu(x + ffn1.2(gelu(ffn1.0 x)));  x2 = gelu(x1 + dw1x1(x1))            (ffn1 + first conv_list entry)
//   chain B:  x4 = gelu(x3 + pw x3);  cor = gelu(ffn2.2(gelu(ffn2.0 x4)))                      (pw + ffn2 + the encoder's outer gelu)
// Every layer has <= 64 input channels, so a 128-pixel tile keeps its activations in LDS (split bf16 rows, the MFMA B
// operand format) from layer to layer: one launch replaces three implicit-GEMM launches plus the depthwise 1x1 kernel.
// Same arithmetic as the unfused path (bf16x3 split MFMA, fp32 accumulate, erf GELU).
#include "common.h"

namespace {

constexpr int TP = 128;                      // pixels per workgroup
constexpr int ROWB = 64;                     // bytes per LDS row: 32 channels x bf16
constexpr int ACT_PLANE = TP * ROWB;         // one 32-channel half (k-step) of one plane: 8 KiB
constexpr int ACT_BUF = 4 * ACT_PLANE;       // [kstep 2][plane 2][128 px][64 B] = 32 KiB
constexpr int W_BLK = 2 * 2 * 64 * 64;       // one 64-cout block x K = 64: [kstep 2][plane 2][64][64 B] = 16 KiB

__device__ __forceinline__ int swzp(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

struct Layer {
    const void* w;          // packed [kstep 2][M/64][plane 2][64][32] (pack_conv2 of a 1x1 conv with 64 padded inputs)
    const float* bias;      // [M]
    const float* post_s;    // optional per-channel affine applied after the activation: y = gelu(y + y*s + t)   (dw 1x1)
    const float* post_t;
    int M;                  // padded couts (64 or 256)
    int n_valid;            // real couts
    int resid;              // 1: add the layer-chain INPUT x (same channel) before the activation
};

struct ChainParams {
    ppms_sp in, out;
    Layer layer[3];
    int nlayers;
    int64_t P;
};

// (The parameter block stays behind a device pointer: passed by value in the kernel arguments -- as the conv descriptors are -- the layer
// loop's dynamic index c.layer[l] sends the struct to scratch memory: chain B at the 1/4 scale 43 -> 54 us, nothing gained at 1/16.)
__global__ __launch_bounds__(256) void pwchain_kernel(const ChainParams* __restrict__ cp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ChainParams& c = *cp;
    // Two activation buffers, used alternately, + one weight block: 80 KiB, so that TWO workgroups share a CU (three buffers: one
    // workgroup per CU and, with 400 tiles on 256 CUs, a half-empty second round).  The residual operand is the chain INPUT, buffer 0:
    // a residual layer is either the first layer (buffer 0 is its source) or the second one (buffer 0 is its DESTINATION: every lane
    // reads the residual at exactly the address it then writes, and nothing else reads buffer 0 in that layer); the host side
    // (engine.py PwChain) refuses chains with a residual further down, where buffer 0 no longer holds the input.
    char* bufX = smem;                        // chain input = buffer 0
    char* bufA = smem + ACT_BUF;              // buffer 1
    char* wsm = smem + 2 * ACT_BUF;           // one 64-cout weight block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * TP;

    // ---- stage the input tile: [kstep][plane][px][64 B], 16-B chunks swizzled ------------------------------------
    for (int q = tid; q < TP * 8; q += 256) {                  // 8 chunks (64 channels) per pixel and plane
        const int px = q >> 3, ch = q & 7;
        const int64_t pix = p0 + px;
        u32x4 vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
        if (pix < c.P) {
            vh = gload16((const bf16_t*)c.in.hi + pix * c.in.ld + ch * 8);
            vl = gload16((const bf16_t*)c.in.lo + pix * c.in.ld + ch * 8);
        }
        const int off = (ch >> 2) * 2 * ACT_PLANE + swzp(px, ch & 3);
        *(u32x4*)(bufX + off) = vh;
        *(u32x4*)(bufX + ACT_PLANE + off) = vl;
    }
    // weight blocks are requested one phase ahead (registers): phase = (layer, 64-cout block)
    u32x4 wreg[4];
    auto load_w = [&](int l, int mblk) __attribute__((always_inline)) {
        const Layer& Lw = c.layer[l];
        const int mbs = Lw.M / 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * 256;
            const int ks = q >> 9, rem = q & 511;                 // 512 chunks (8 KiB) per k-step
            wreg[i] = gload16((const char*)Lw.w + ((int64_t)(ks * mbs + mblk) * 512 + rem) * 16);
        }
    };
    load_w(0, 0);
    const char* src = bufX;
    for (int l = 0; l < c.nlayers; ++l) {
        const Layer& L = c.layer[l];
        const bool last = (l == c.nlayers - 1);
        char* dst = (src == bufX) ? bufA : bufX;
        const int mblocks = L.M / 64;
        for (int mblk = 0; mblk < mblocks; ++mblk) {
            __syncthreads();                                      // previous users of wsm / producers of src are done
            // weight block: k-step ks of block mblk lives at ((ks * mblocks + mblk) * 8 KiB) in the packed tensor
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(wsm + (tid + i * 256) * 16) = wreg[i];
            if (mblk + 1 < mblocks)
                load_w(l, mblk + 1);
            else if (l + 1 < c.nlayers)
                load_w(l + 1, 0);
            __syncthreads();
            // wave w: 64 couts x pixels [32 w, 32 w + 32)
            f32x16 acc[2];
            acc[0] = (f32x16){0};
            acc[1] = (f32x16){0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int k16 = 0; k16 < 2; ++k16) {
                    const int boff = ks * 2 * ACT_PLANE + swzp(wave * 32 + r, 2 * k16 + h);
                    const bf16x8 bh = *(const bf16x8*)(src + boff);
                    const bf16x8 bl = *(const bf16x8*)(src + ACT_PLANE + boff);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int aoff = ks * 8192 + swzp(mb * 32 + r, 2 * k16 + h);
                        const bf16x8 ah = *(const bf16x8*)(wsm + aoff);
                        const bf16x8 al = *(const bf16x8*)(wsm + 4096 + aoff);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mb], 0, 0, 0);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mb], 0, 0, 0);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mb], 0, 0, 0);
                    }
                }
            }
            // ---- epilogue of this 64-cout block: lane = pixel wave*32 + r, couts cl = mb*32 + 8 g + 4 h + j -------------
            const int px = wave * 32 + r;
            const int64_t pix = p0 + px;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = mb * 32 + 8 * g + 4 * h;          // within the block
                    const int cg = mblk * 64 + cl;                    // global cout
                    const f32x4 b4 = gld<f32x4>(L.bias + cg);
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = acc[mb][4 * g + j] + b4[j];
                    if (L.resid) {                                     // chain input x at the same channels (M == 64 here)
                        const int xo = (cl >> 5) * 2 * ACT_PLANE + swzp(px, (cl & 31) >> 3) + (cl & 7) * 2;
                        const bf16x4 xh = *(const bf16x4*)(bufX + xo), xl = *(const bf16x4*)(bufX + ACT_PLANE + xo);
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] += join_bf16(xh[j], xl[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = gelu_erf(y[j]);
                    if (L.post_s != nullptr) {
                        const f32x4 s4 = gld<f32x4>(L.post_s + cg), t4 = gld<f32x4>(L.post_t + cg);
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = gelu_erf(y[j] + (y[j] * s4[j] + t4[j]));
                    }
                    bf16x4 oh, ol;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16_t hh = (bf16_t)0.0f, ll = (bf16_t)0.0f;
                        if (cg + j < L.n_valid) split_bf16(y[j], hh, ll);   // padded couts stay exactly zero
                        oh[j] = hh;
                        ol[j] = ll;
                    }
                    {
                        const int xo = (cl >> 5) * 2 * ACT_PLANE + swzp(px, (cl & 31) >> 3) + (cl & 7) * 2;
                        *(bf16x4*)(dst + xo) = oh;
                        *(bf16x4*)(dst + ACT_PLANE + xo) = ol;
                    }
                }
            if (last) {
                // the 64-cout block sits in LDS like an intermediate layer; copy it out with whole 128-byte runs per pixel
                // and plane (in the accumulator layout a wave's store touched 32 cache lines with 16 useful bytes each)
                __syncthreads();
                for (int qd = tid; qd < TP * 16; qd += 256) {               // 8 chunks x 2 planes per pixel
                    const int opx = qd >> 4, plane = (qd >> 3) & 1, ch = qd & 7;
                    const int64_t opix = p0 + opx;
                    const int ocg = mblk * 64 + ch * 8;
                    if (opix < c.P && ocg < L.n_valid) {
                        const u32x4 v = *(const u32x4*)(dst + (ch >> 2) * 2 * ACT_PLANE + plane * ACT_PLANE + swzp(opx, ch & 3));
                        bf16_t* op = (bf16_t*)(plane ? c.out.lo : c.out.hi) + opix * c.out.ld + ocg;
                        if (ocg + 8 <= L.n_valid) {
                            gstore16(op, v);
                        } else {                                                // ragged tail of the valid couts
                            const bf16_t* e = (const bf16_t*)&v;
                            for (int j = 0; j < 8; ++j)
                                if (ocg + j < L.n_valid) gst<bf16_t>(op + j, e[j]);
                        }
                    }
                }
            }
        }
        src = dst;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Small maps (<= 16 384 pixels: the 1/8 and 1/16 scales).  The kernel above gives them 25-100 workgroups that walk 3-6 PHASES one after
// the other (a phase = one 64-cout block of one layer: weights to LDS, barrier, 12 MFMAs per wave, epilogue): 20-33 us for a few MFLOP.
// Here a workgroup owns 32 pixels (4x the workgroups) and a LAYER is one phase: all its weight blocks sit in LDS at once (<= 64 KiB) and
// the waves split the 32-cout row blocks between them (a 256-cout layer: two row blocks per wave; a 64-cout layer: waves 0 and 1).
// Same packs, same arithmetic per output element (k order, split products, epilogue), so the results are the big kernel's bit for bit.
constexpr int TP32 = 32;
constexpr int ACT32_PLANE = TP32 * ROWB;     // 2 KiB
constexpr int ACT32_BUF = 4 * ACT32_PLANE;   // [kstep 2][plane 2][32 px][64 B] = 8 KiB
constexpr int W32_MAX = 4 * W_BLK;           // weights of a 256-cout layer: 64 KiB (reused as the output staging patch of the last layer)

__global__ __launch_bounds__(256) void pwchain32_kernel(const ChainParams* __restrict__ cp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ChainParams& c = *cp;
    char* bufX = smem;
    char* bufA = smem + ACT32_BUF;
    char* wsm = smem + 2 * ACT32_BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * TP32;
    {   // input tile: 32 px x 8 chunks (64 channels) per plane = one 16-byte piece of each plane per thread
        const int px = tid >> 3, ch = tid & 7;
        const int64_t pix = p0 + px;
        u32x4 vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
        if (pix < c.P) {
            vh = gload16((const bf16_t*)c.in.hi + pix * c.in.ld + ch * 8);
            vl = gload16((const bf16_t*)c.in.lo + pix * c.in.ld + ch * 8);
        }
        const int off = (ch >> 2) * 2 * ACT32_PLANE + swzp(px, ch & 3);
        *(u32x4*)(bufX + off) = vh;
        *(u32x4*)(bufX + ACT32_PLANE + off) = vl;
    }
    u32x4 wreg[16];                          // the next layer's weights (<= 4 blocks x 4 pieces per thread), requested one layer ahead
    auto load_w = [&](int l) __attribute__((always_inline)) {
        const Layer& Lw = c.layer[l];
        const int mbs = Lw.M / 64;
#pragma unroll
        for (int mblk = 0; mblk < 4; ++mblk)
            if (mblk < mbs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = tid + i * 256;
                    const int ks = q >> 9, rem = q & 511;
                    wreg[mblk * 4 + i] = gload16((const char*)Lw.w + ((int64_t)(ks * mbs + mblk) * 512 + rem) * 16);
                }
            }
    };
    load_w(0);
    const char* src = bufX;
    for (int l = 0; l < c.nlayers; ++l) {
        const Layer& L = c.layer[l];
        const bool last = (l == c.nlayers - 1);
        char* dst = (src == bufX) ? bufA : bufX;
        const int mbs = L.M / 64, nrb = L.M / 32;
        __syncthreads();                                          // the previous layer's readers of wsm and writers of src are done
#pragma unroll
        for (int mblk = 0; mblk < 4; ++mblk)
            if (mblk < mbs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(u32x4*)(wsm + mblk * W_BLK + (tid + i * 256) * 16) = wreg[mblk * 4 + i];
            }
        if (l + 1 < c.nlayers) load_w(l + 1);
        __syncthreads();
        bf16x4 keep_h[2][4], keep_l[2][4];                        // last layer: the wave's (<= 2) row blocks wait in registers for the staging patch
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int rb = wave + 4 * it;                         // 32-cout row block of the layer
            if (rb >= nrb) continue;
            const int mblk = rb >> 1, mb = rb & 1;
            f32x16 acc = (f32x16){0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int k16 = 0; k16 < 2; ++k16) {
                    const int boff = ks * 2 * ACT32_PLANE + swzp(r, 2 * k16 + h);
                    const bf16x8 bh = *(const bf16x8*)(src + boff), bl = *(const bf16x8*)(src + ACT32_PLANE + boff);
                    const int aoff = mblk * W_BLK + ks * 8192 + swzp(mb * 32 + r, 2 * k16 + h);
                    const bf16x8 ah = *(const bf16x8*)(wsm + aoff), al = *(const bf16x8*)(wsm + 4096 + aoff);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = mb * 32 + 8 * g + 4 * h;             // within the 64-cout block
                const int cg = mblk * 64 + cl;                       // global cout
                const f32x4 b4 = gld<f32x4>(L.bias + cg);
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = acc[4 * g + j] + b4[j];
                const int xo = (cl >> 5) * 2 * ACT32_PLANE + swzp(r, (cl & 31) >> 3) + (cl & 7) * 2;
                if (L.resid) {
                    const bf16x4 xh = *(const bf16x4*)(bufX + xo), xl = *(const bf16x4*)(bufX + ACT32_PLANE + xo);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] += join_bf16(xh[j], xl[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = gelu_erf(y[j]);
                if (L.post_s != nullptr) {
                    const f32x4 s4 = gld<f32x4>(L.post_s + cg), t4 = gld<f32x4>(L.post_t + cg);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = gelu_erf(y[j] + (y[j] * s4[j] + t4[j]));
                }
                bf16x4 oh, ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16_t hh = (bf16_t)0.0f, ll = (bf16_t)0.0f;
                    if (cg + j < L.n_valid) split_bf16(y[j], hh, ll);
                    oh[j] = hh;
                    ol[j] = ll;
                }
                if (last) {
                    keep_h[it][g] = oh;
                    keep_l[it][g] = ol;
                } else {
                    *(bf16x4*)(dst + xo) = oh;
                    *(bf16x4*)(dst + ACT32_PLANE + xo) = ol;
                }
            }
        }
        if (last) {
            // staging patch over the weight area: [plane][32 px][M couts] bf16, then whole 16-byte pieces per pixel and plane to memory
            __syncthreads();                                      // every wave is done reading the weights
            const int rowb = L.M * 2 + 16;                        // bytes per pixel row of a plane (+16: rows start in different banks)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int rb = wave + 4 * it;
                if (rb >= nrb) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cg = rb * 32 + 8 * g + 4 * h;
                    *(bf16x4*)(wsm + r * rowb + cg * 2) = keep_h[it][g];
                    *(bf16x4*)(wsm + TP32 * rowb + r * rowb + cg * 2) = keep_l[it][g];
                }
            }
            __syncthreads();
            const int cpr = L.M / 8;                              // 16-byte pieces per pixel row
            for (int qd = tid; qd < TP32 * 2 * cpr; qd += 256) {
                const int plane = qd / (TP32 * cpr), rem = qd - plane * TP32 * cpr;
                const int opx = rem / cpr, ch = rem - opx * cpr;
                const int64_t opix = p0 + opx;
                const int ocg = ch * 8;
                if (opix < c.P && ocg < L.n_valid) {
                    const u32x4 v = *(const u32x4*)(wsm + plane * TP32 * rowb + opx * rowb + ocg * 2);
                    bf16_t* op = (bf16_t*)(plane ? c.out.lo : c.out.hi) + opix * c.out.ld + ocg;
                    if (ocg + 8 <= L.n_valid) {
                        gstore16(op, v);
                    } else {
                        const bf16_t* e = (const bf16_t*)&v;
                        for (int j = 0; j < 8; ++j)
                            if (ocg + j < L.n_valid) gst<bf16_t>(op + j, e[j]);
                    }
                }
            }
        }
        src = dst;
    }
}

}  // namespace

static int pwchain_launch(const void* dev_params, int64_t pixels, void* stream) {
    constexpr size_t lds = 2 * ACT_BUF + W_BLK;                // 80 KiB: two workgroups per CU
    constexpr size_t lds32 = 2 * ACT32_BUF + W32_MAX;          // 80 KiB
    static ppms_device_once once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)pwchain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * ACT_BUF + W_BLK));
        (void)hipFuncSetAttribute((const void*)pwchain32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * ACT32_BUF + W32_MAX));
    });
    if (pixels <= 16384) {                                       // small maps: 32-pixel tiles, one phase per layer
        hipLaunchKernelGGL(pwchain32_kernel, dim3(ceil_div(pixels, TP32)), dim3(256), lds32, (hipStream_t)stream, (const ChainParams*)dev_params);
        return ppms_check_launch("pwchain");
    }
    hipLaunchKernelGGL(pwchain_kernel, dim3(ceil_div(pixels, TP)), dim3(256), lds, (hipStream_t)stream, (const ChainParams*)dev_params);
    return ppms_check_launch("pwchain");
}

extern "C" int ppms_pwchain(const void* dev_params, int64_t pixels, void* stream) {
    PPMS_REQUIRE(dev_params != nullptr && pixels > 0, "pwchain: bad arguments");
    return pwchain_launch(dev_params, pixels, stream);
}

extern "C" int ppms_pwchain_param_bytes(void) { return (int)sizeof(ChainParams); }
