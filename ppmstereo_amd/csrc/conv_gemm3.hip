// Implicit-GEMM convolution, large-map variant: 128 couts x 256 pixels per workgroup, 64 x 128 per wave.
//
// Same math, descriptor and epilogues as conv_gemm2.hip; tuned for maps with >= ~200 tiles (the 1/4 scale):
//   * wave tile 64 couts x 128 pixels (2 x 4 MFMA 32x32x16 tiles, 128 accumulator registers): 24 ds_read_b128 feed
//     48 MFMAs per k-step instead of 16 per 24 -- measured ceiling of that mix on MI355X: 1.93 PF vs 1.54 PF (bf16,
//     one barrier per step, tools/probe/mfma_probe.hip);
//   * both operands reach LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  The weight tile is
//     a linear 16 KiB copy of the pre-swizzled packed image; the activation window is gathered with per-lane source
//     addresses (the XOR swizzle is applied on the SOURCE chunk index, the LDS destination stays lane-linear) and
//     out-of-image rows read a zero page;
//   * the activation window is a halo'd patch swept by the taps from LDS: along x (kh == 1), along y (kw == 1) or over
//     all kh x kw taps (2-D halo): a window switch stalls the workgroup for a global -> LDS round trip, so the more
//     taps share a window the better (3x3: one switch per 9 k-steps instead of per 3).
// Workgroup = 4 waves (wm, wn); LDS = 2 x 16 KiB weight stages + one window (<= 40 KiB): two workgroups per CU.
// Packed weights: pack_conv2 order, k-step = (rowstep * nchunk + chunk) * nsweep + s: y-sweep convs are packed with
// their kh / kw axes swapped, 2-D sweep convs with (ky, kx) flattened into the x axis (weight viewed as
// (cout, cin, kt, 1, kh*kw)), see ppmstereo_amd/engine.py.
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int A_BLK = 64 * BK * 2 * 2;        // 8 KiB: one 64-cout block, hi + lo
constexpr int WM = 2, NT = 256;
constexpr int A_STAGE = WM * A_BLK;           // 16 KiB
constexpr int MAXS = 6;                       // window 16-B chunks per thread and plane (window <= 384 rows)

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];      // zero-initialised: source of padded rows

struct Geo3 {
    int C, R, logC;          // patch R x C, R*C = 256
    int tiles_x, tiles_y;
    int WRL;                 // window row length (pixels)
    int Wr;                  // window rows
    int hxw, hyw;            // halo of the window in x / y
    int swx_n, row_jump;     // sweep: the LDS row advances by 1 per tap and by row_jump more after every swx_n taps
                             //   x sweep: (nsweep, 0);  y sweep: (1, WRL - 1);  2-D sweep (kh x kw taps): (kw, WRL - kw)
    int nsweep;              // taps swept inside one window (kw, kh or kh*kw)
    int nrow;                // row-steps per chunk set: taps NOT swept (kt*kh for x sweep, kt otherwise)
    int rdy;                 // 1: row-step index carries a dy (x sweep), 0: only dt
    int nchunk, n0;
    int mgroups;
};

__device__ __forceinline__ int swz3(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const PPMS_GLOBAL void*)(uintptr_t)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv3_kernel(const ppms_conv pv, const Geo3 g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ppms_conv& p = pv;                       // by value in the kernel arguments (see conv_gemm2.hip)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int mgrp = blockIdx.x % g.mgroups;
    int tile = blockIdx.x / g.mgroups;
    const int tx = tile % g.tiles_x;
    tile /= g.tiles_x;
    const int ty = tile % g.tiles_y;
    const int tf = tile / g.tiles_y;
    const int x0 = tx * g.C, y0 = ty * g.R;
    const int H = p.H, W = p.W, T = p.T;
    const int HW = H * W;
    const int ht = p.kt >> 1, hy = p.kh >> 1;

    char* sA = smem;                               // 2 x 16 KiB
    char* sB = smem + 2 * A_STAGE;                 // 2 planes x Wr x 64 B
    const int bplane = g.Wr * 64;

    // ---- window slots: LDS chunk q = tid + i*256 (lane-linear destination); source chunk = swizzled column ----------
    int sl_off[MAXS];          // pixel offset at (dt, dy) = 0, -1: column outside the image / slot unused
    int sl_y[MAXS];
    const int nq = g.Wr * 4;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const int q = tid + i * NT;
        sl_off[i] = -1;
        sl_y[i] = 0;
        if (q < nq) {
            const int wrow = q >> 2;
            const int wy = wrow / g.WRL, wx = wrow - wy * g.WRL;
            const int x = x0 + wx - g.hxw, y = y0 + wy - g.hyw;
            sl_y[i] = y;
            if ((unsigned)x < (unsigned)W) sl_off[i] = (tf * H + y) * W + x;
        }
    }
    // source chunk of this thread's slots: LDS position (q & 3) holds chunk (q & 3) ^ ((row >> 2) & 3); row = q >> 2 and
    // q = tid + 256 i  =>  row = (tid >> 2) + 64 i, so ((row >> 2) & 3) = ((tid >> 4) & 3) for every i
    const int src_chunk = (tid & 3) ^ ((tid >> 4) & 3);

    const char* wbase = (const char*)p.w + (int64_t)mgrp * A_STAGE + tid * 16;
    const int64_t wstep = (int64_t)(p.M / 64) * A_BLK;
    auto dma_a = [&](int ks, int stage) {
        const char* wp = wbase + (int64_t)ks * wstep;
        char* s = sA + stage * A_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(wp + i * NT * 16, s + i * NT * 16);
    };
    auto dma_b = [&](int rowstep, int chunk) {
        int dy = 0, dt;
        if (g.rdy) {
            const int ky = rowstep % p.kh;
            dy = ky - hy;
            dt = rowstep / p.kh - ht;
        } else {
            dt = rowstep - ht;
        }
        const int sg = (chunk >= g.n0) ? 1 : 0;
        const int c0 = (chunk - (sg ? g.n0 : 0)) * BK + src_chunk * 8;
        const bf16_t* sh = (const bf16_t*)p.seg[sg].hi;
        const bf16_t* sl = (const bf16_t*)p.seg[sg].lo;
        const int ld = p.seg[sg].ld;
        const bool tok = (unsigned)(tf + dt + p.t_halo) < (unsigned)(T + 2 * p.t_halo);
        const int shift = (dt * H + dy) * W;
        char* d = sB + wave * 1024;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) {
            if (i * NT < nq) {                                   // wave-uniform: whole 1 KiB pieces only (nq % 64 == 0 by construction)
                if (tid + i * NT < nq) {
                    const bool ok = tok && sl_off[i] >= 0 && (unsigned)(sl_y[i] + dy) < (unsigned)H;
                    const int64_t off = (int64_t)(sl_off[i] + shift) * ld + c0;
                    const void* ph = ok ? (const void*)(sh + off) : (const void*)g_zero_page;
                    const void* pl = ok ? (const void*)(sl + off) : (const void*)g_zero_page;
                    dma16(ph, d + i * NT * 16);
                    dma16(pl, d + bplane + i * NT * 16);
                }
            }
        }
    };

    // ---- B-operand rows of this lane's four pixel blocks (at sweep tap 0) ------------------------------------------
    int brow[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int pid = wn * 128 + nb * 32 + r;
        brow[nb] = (pid >> g.logC) * g.WRL + (pid & (g.C - 1));
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x16){0};

    // temporal taps outside [0, T) contribute zeros: skip them (contiguous kz range)
    const int kz0 = (ht - tf - p.t_halo) > 0 ? (ht - tf - p.t_halo) : 0;
    const int kz1 = (ht + T + p.t_halo - 1 - tf) < (p.kt - 1) ? (ht + T + p.t_halo - 1 - tf) : (p.kt - 1);
    const int rows_per_kz = g.rdy ? p.kh : 1;
    int rs = kz0 * rows_per_kz * g.nchunk;                       // row-step index = rowstep * nchunk + chunk
    const int rs_end = (kz1 + 1) * rows_per_kz * g.nchunk;
    const int nsteps = (rs_end - rs) * g.nsweep;

    dma_a(rs * g.nsweep, 0);
    dma_b(rs / g.nchunk, rs % g.nchunk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int sw = 0, swx = 0, trow = 0;
    for (int j = 0; j < nsteps; ++j) {
        const bool more = j + 1 < nsteps;
        const bool need_b = more && (sw + 1 == g.nsweep);
        if (more) dma_a(need_b ? (rs + 1) * g.nsweep : rs * g.nsweep + sw + 1, (j + 1) & 1);
        const char* a_s = sA + (j & 1) * A_STAGE + wm * A_BLK;
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int off = swz3(mb * 32 + r, 2 * k16 + h);
                ah[mb] = *(const bf16x8*)(a_s + off);
                al[mb] = *(const bf16x8*)(a_s + 4096 + off);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int off = swz3(brow[nb] + trow, 2 * k16 + h);
                bh[nb] = *(const bf16x8*)(sB + off);
                bl[nb] = *(const bf16x8*)(sB + bplane + off);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                }
        }
        if (need_b) {
            // single window: every wave must be done sweeping it before the DMA overwrites it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int nrs = rs + 1;
            dma_b(nrs / g.nchunk, nrs % g.nchunk);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (need_b) {
            sw = swx = trow = 0;
            ++rs;
        } else {
            ++sw;
            ++trow;
            if (++swx == g.swx_n) {
                swx = 0;
                trow += g.row_jump;
            }
        }
    }

    // ---- epilogue: accumulators -> wave-private LDS patch [32 px][64 couts] -> 8 couts of one pixel per lane -------
    // (see conv_epilogue.h; the loop's last barrier freed the operand stages, each wave only touches its own patch)
    const int cblock = (mgrp * WM + wm) * 64;
    const int half = (cblock >= p.m_split) ? 1 : 0;
    const ppms_epilogue e = p.epi[half];          // BY VALUE (SGPRs): through a reference every field is re-read from memory behind every
                                                  // store of the row loop (the stores might alias the descriptor), one scalar-load round trip each
    const int cbase = cblock - (half ? p.m_split : 0);
    float* stg = (float*)(smem + wave * STG_WAVE);
    const int q = lane & 7;
    float b8[8];
    {
        const f32x4 b0 = gld<f32x4>(p.bias + cblock + q * 8), b1 = gld<f32x4>(p.bias + cblock + q * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b8[j] = b0[j];
            b8[4 + j] = b1[j];
        }
        // the bias must have LANDED before the row loop: vmcnt counts loads and stores in one order, so a wait for this load placed
        // inside the loop (where its first use is) is a wait for every store of the previous 8-row step too -- one HBM write round
        // trip (~1 us) per step, which is what the epilogues cost before this line
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(b8[j]));
    }
    // the row loop exists twice: once for plain STORE epilogues, whose body holds no load (so nothing in it ever waits for the previous
    // step's stores: conv_epilogue.h), once for everything else
    auto rows = [&](auto ld_tag) {
        constexpr bool LD = decltype(ld_tag)::value;
    #pragma unroll 1
        for (int nb = 0; nb < 4; ++nb) {
    #pragma unroll
            for (int mb = 0; mb < 2; ++mb)
    #pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 a4;
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[mb][0][4 * gq + j];
                        x = (nb == 1) ? acc[mb][1][4 * gq + j] : x;
                        x = (nb == 2) ? acc[mb][2][4 * gq + j] : x;
                        x = (nb == 3) ? acc[mb][3][4 * gq + j] : x;
                        a4[j] = x;
                    }
                    if (LD && e.out_vt != nullptr) {                           // pixel-major V^T straight from the accumulator layout
                        const int pid = wn * 128 + nb * 32 + r;
                        const int px = x0 + (pid & (g.C - 1)), py = y0 + (pid >> g.logC);
                        const int c4 = mb * 32 + 8 * gq + 4 * h;
                        const f32x4 bb = gld<f32x4>(p.bias + cblock + c4);
                        float v4[4];
    #pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = a4[j] + bb[j];
                        if (px < W && py < H) epilogue_vt4(e, v4, tf, py * W + px, cbase + c4, HW);
                    }
                    stage_write32(stg, r, h, mb, gq, a4);
                }
            __builtin_amdgcn_wave_barrier();
    #pragma unroll 1
            for (int it = 0; it < 4; ++it) {
                const int prow = it * 8 + (lane >> 3);
                float v[8];
                stage_read8(stg, prow, q, v);
                const int pid = wn * 128 + nb * 32 + prow;
                const int px = x0 + (pid & (g.C - 1)), py = y0 + (pid >> g.logC);
                if (px < W && py < H) {
                    const int64_t pix = (int64_t)(tf * H + py) * W + px;
    #pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += b8[j];
                    epilogue_row8<LD>(e, v, pix, cbase + q * 8, HW);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (epilogue_is_plain(e)) rows(std::false_type{});
    else rows(std::true_type{});
}

}  // namespace

// returns 1 when the large-map kernel applies to this convolution (the caller then packs y-sweep convs with swapped axes)
extern "C" int ppms_conv_gemm3_applicable(const ppms_conv* d) {
    if (d == nullptr || d->M % 128 != 0 || d->m_split % 64 != 0) return 0;
    if (d->kw == 1 && d->kh == 1) return 0;
    if (d->kw > 1 && d->kh > 1) {                            // 2-D window: some 256-pixel patch with its halo must fit
        bool fits = false;
        for (int C = 8; C <= 256; C *= 2) fits = fits || (256 / C + d->kh - 1) * (C + d->kw - 1) <= 64 * MAXS;
        if (!fits) return 0;
    }
    const int64_t P = (int64_t)d->T * d->H * d->W;
    if (P >= (1ll << 31) / 512) return 0;                    // 32-bit byte offsets inside the kernel (ppms_conv_gemm3 checks the same)
    if (P / 256 * (d->M / 128) < 384) return 0;              // fewer than ~1.5 workgroups per CU: conv_gemm2's smaller tiles fill the chip better
    return 1;
}

extern "C" int ppms_conv_gemm3(const ppms_conv* d, const ppms_conv* dev_desc, void* stream) {
    PPMS_REQUIRE(d != nullptr && dev_desc != nullptr, "conv_gemm3: null descriptor");
    PPMS_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv_gemm3: nseg=%d", d->nseg);
    PPMS_REQUIRE(d->M > 0 && d->M % 128 == 0 && d->m_split % 64 == 0, "conv_gemm3: M=%d must be a multiple of 128", d->M);
    PPMS_REQUIRE((d->kt & 1) && (d->kh & 1) && (d->kw & 1) && d->kw <= 15 && d->kh <= 15, "conv_gemm3: odd kernel extents <= 15");
    PPMS_REQUIRE(d->kw > 1 || d->kh > 1, "conv_gemm3: needs a spatial sweep axis (kw > 1 or kh > 1)");
    PPMS_REQUIRE(d->w != nullptr && d->bias != nullptr, "conv_gemm3: weights/bias missing");
    PPMS_REQUIRE(d->t_halo >= 0 && d->t_halo <= 8, "conv_gemm3: t_halo=%d", d->t_halo);
    PPMS_REQUIRE((int64_t)d->T * d->H * d->W < (1ll << 31) / 512, "conv_gemm3: volume too large for 32-bit pixel offsets");
    int nchunk = 0;
    for (int s = 0; s < d->nseg; ++s) {
        PPMS_REQUIRE(d->seg[s].hi && d->seg[s].lo && d->seg[s].c > 0 && d->seg[s].c % BK == 0 && d->seg[s].ld % 8 == 0,
                     "conv_gemm3: segment %d needs hi/lo planes, c %% 32 == 0 and ld %% 8 == 0", s);
        PPMS_REQUIRE(((uintptr_t)d->seg[s].hi & 15) == 0 && ((uintptr_t)d->seg[s].lo & 15) == 0, "conv_gemm3: segment %d not 16-B aligned", s);
        nchunk += d->seg[s].c / BK;
    }
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && d->m_split >= d->M) break;
        PPMS_REQUIRE(e.n_valid > 0, "conv_gemm3: epilogue %d has n_valid=%d", hlf, e.n_valid);
        PPMS_REQUIRE(e.pre_f32 == nullptr || (e.n_valid % 4 == 0 && e.pre_f32_ld % 4 == 0), "conv_gemm3: pre_f32 needs n_valid and pre_f32_ld to be multiples of 4");
        {
            const char* why = epilogue_row8_check(e);
            PPMS_REQUIRE(why == nullptr, "conv_gemm3: epilogue %d: %s", hlf, why ? why : "");
        }
        if (e.out_sp.hi) PPMS_REQUIRE(e.out_sp.lo && e.out_sp.ld % 4 == 0, "conv_gemm3: epilogue %d SP output misaligned", hlf);
        if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU)
            PPMS_REQUIRE(e.aux_sp.hi && e.aux_sp.lo && e.aux_sp.ld % 4 == 0, "conv_gemm3: epilogue %d needs aux_sp", hlf);
        if (e.kind == PPMS_EPI_GRU) PPMS_REQUIRE(e.aux_f32 != nullptr, "conv_gemm3: GRU epilogue needs z");
    }
    const int mode = (d->kw > 1 && d->kh > 1) ? 2 : (d->kw > 1 ? 0 : 1);      // 0: x sweep, 1: y sweep, 2: 2-D sweep
    const int hx = mode != 1 ? d->kw - 1 : 0, hy = mode != 0 ? d->kh - 1 : 0;   // window halo (total) in x / y
    Geo3 g;
    // patch shape: x sweep wants wide patches (halo = kw-1 columns per row), y sweep tall ones (halo = kh-1 rows);
    // among the shapes whose window fits, take the one wasting the fewest pixels, then the smallest window
    int bestC = -1;
    double bestw = 1e30;
    int bestWr = 1 << 30;
    for (int C = 8; C <= 256; C *= 2) {
        const int R = 256 / C;
        const int Wr = (R + hy) * (C + hx);
        if (Wr > 64 * MAXS || C < 8) continue;
        const double waste = (double)((d->W + C - 1) / C * C) * ((d->H + R - 1) / R * R) / ((double)d->W * d->H);
        if (waste < bestw - 1e-9 || (waste < bestw + 1e-9 && Wr < bestWr)) {
            bestw = waste;
            bestC = C;
            bestWr = Wr;
        }
    }
    PPMS_REQUIRE(bestC > 0, "conv_gemm3: no patch shape fits the LDS window");
    g.C = bestC;
    g.R = 256 / bestC;
    g.logC = 0;
    while ((1 << g.logC) < g.C) ++g.logC;
    g.tiles_x = (d->W + g.C - 1) / g.C;
    g.tiles_y = (d->H + g.R - 1) / g.R;
    g.WRL = g.C + hx;
    g.Wr = (g.R + hy) * g.WRL;
    g.hxw = hx >> 1;
    g.hyw = hy >> 1;
    if (mode == 0) {
        g.swx_n = d->kw;
        g.row_jump = 0;
        g.nsweep = d->kw;
        g.nrow = d->kt * d->kh;
        g.rdy = 1;
    } else if (mode == 1) {
        g.swx_n = 1;
        g.row_jump = g.WRL - 1;
        g.nsweep = d->kh;
        g.nrow = d->kt;
        g.rdy = 0;
    } else {
        g.swx_n = d->kw;
        g.row_jump = g.WRL - d->kw;
        g.nsweep = d->kh * d->kw;
        g.nrow = d->kt;
        g.rdy = 0;
    }
    g.Wr = (g.Wr + 15) / 16 * 16;                 // whole 1 KiB DMA pieces (16 rows x 64 B); the extra rows are never read
    g.nchunk = nchunk;
    g.n0 = d->seg[0].c / BK;
    g.mgroups = d->M / 128;
    const int ntiles = g.tiles_x * g.tiles_y * d->T;
    size_t lds = (size_t)2 * A_STAGE + (size_t)2 * g.Wr * 64;
    if (lds < (size_t)4 * STG_WAVE) lds = (size_t)4 * STG_WAVE;          // the epilogue's transposition patches reuse the stages
    PPMS_REQUIRE(g.Wr <= 64 * MAXS && lds <= 80 * 1024, "conv_gemm3: window of %d rows does not fit", g.Wr);
    static ppms_device_once once;
    once.run([] { (void)hipFuncSetAttribute((const void*)conv3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); });
    hipLaunchKernelGGL(conv3_kernel, dim3(ntiles * g.mgroups), dim3(NT), lds, (hipStream_t)stream, *d, g);
    return ppms_check_launch("conv_gemm3");
}
