// Implicit-GEMM convolution, one 8-wave workgroup per CU with SIMD-balanced 7- or 8-block pixel tiles (gfx950).
//
// Same math, descriptor and epilogues as conv_gemm2/3/4 (bf16x3 split MFMA: hi*hi + hi*lo + lo*hi, fp32 accumulate) and the
// barrier-light k-loop of conv_gemm4 (weights from L2 straight to registers in MFMA-fragment order, 16-channel activation
// windows double buffered by LDS-DMA, B fragments prefetched inside the step, one barrier per window).  What it adds is the
// TILE: at BASELINE config 2 the 1/4-scale map has 51 200 pixels = 200 per CU, and power-of-two tiles (256 pixels x 128 couts,
// two workgroups per CU) leave 22 % of the CU-slots empty while the others hold two workgroups -- 1.28x the balanced time.
// Here a workgroup owns ALL couts of a tile of NBT = 7 (or 8) blocks of 32 pixels and a CU holds exactly one workgroup:
// 51 200 / 224 -> 240 workgroups on 256 CUs (89 % vs 78 %).  A 7-block tile splits 4 + 3 between the two pixel halves (wn) of
// the wave grid; the eight waves are laid out so that the two waves sharing a SIMD (w and w + 4) take 4 + 3 blocks:
//   M = 256: wave w -> cout block wm = w & 3 (4 x 64 couts), wn = w >> 2;
//   M = 128: two K-groups of four waves (kg = w >> 2) each own every other activation window of the K loop and their own pair
//            of window buffers; (wm, wn) = ((w & 3) >> 1, (w & 1) ^ kg), partial tiles summed through LDS in fixed order.
// Weights: pack_conv4 layout (ppmstereo_amd/packing.py).  LDS: <= 4 window buffers of <= 24 KiB, reused by the K-group
// reduction (128 KiB) and the epilogue's transposition patches.
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

constexpr int NT5 = 512;
constexpr int MAXS5 = 8;                      // window 16-B pieces per thread of a K-group
constexpr int DEPTH5 = 2;                     // A-fragment stages in registers: the next step's are loaded during this step's first MFMA group

__device__ __attribute__((aligned(256))) unsigned int g_zero_page5[64];     // zero-initialised: source of padded rows

struct Geo5 {
    int C, R, logC;          // patch R x C pixels, R*C = 32 * NBT
    int NBT;                 // 32-pixel blocks per tile: 7 (4 + 3) or 8 (4 + 4)
    int tiles_x, tiles_y;
    int WRL, Wr;             // window row length (pixels) and rows (padded to 16)
    int hxw, hyw;            // halo of the window in x / y
    int swx_n, row_jump;     // sweep: the LDS row advances by 1 per tap and by row_jump more after every swx_n taps
    int nsweep;              // taps swept inside one window (kw, kh or kh*kw)
    int rdy;                 // 1: the row-step index carries a dy (x sweep), 0: only dt
    int nchunk, n0;          // 16-channel chunks per tap (all segments), chunks of segment 0
    int lz0;                 // windows whose chunk index is >= lz0 hold bf16-exact activations (all-zero lo plane, ppms_conv.lo_zero_from): their
                             // a_hi x b_lo products are skipped; nchunk: none
    int kgroups;             // 1: M = 256 (4 cout blocks x 2 pixel halves), 2: M = 128 (2 x 2 x two K-groups)
    int npieces;             // DMA pieces per thread of a K-group and window
    int RW, logRW;           // 16-channel chunks per window row: 1 (a window = 16 channels, swept by the spatial taps) or, for convs
                             // without a spatial sweep (kh = kw = 1: GEMM mode), 4 / 2 (M = 256 / 128): the "sweep" then steps through
                             // the chunks of the 256- / 128-byte rows; nchunk / n0 count WINDOWS (16 RW channels) in that mode
    int trow_inc;            // LDS row advance per step inside a window: 1 (sweep), 0 (GEMM mode)
    int nslice;              // grid-level K slices (gridDim.y): slice s takes the s-th share of every K-group's windows (small maps)
    float* part;             // nslice > 1: fp32 partial sums [slice][pixel][M] (no bias), finished by the slice-reduce kernel
    int64_t P;               // pixels = T*H*W
#ifdef PPMS_CONV5_TIMING
    long long* dbg;          // debug build only: [workgroup][8] wall-clock stamps (100 MHz) of wave 0: start, loop start, loop end, reduced, end
#endif
};

#ifdef PPMS_CONV5_TIMING
static long long* g_conv5_dbg = nullptr;
#define CONV5_STAMP(K)                                                                    \
    if (g.dbg != nullptr && (__builtin_amdgcn_readfirstlane(threadIdx.x) & 255) == 0) /* whole waves 0 and 4: a wave-uniform branch */        \
        g.dbg[(int64_t)blockIdx.x * 16 + (__builtin_amdgcn_readfirstlane(threadIdx.x) >> 8) * 8 + (K)] = wall_clock64();
#else
#define CONV5_STAMP(K)
#endif

__device__ __forceinline__ void dma16_5(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const PPMS_GLOBAL void*)(uintptr_t)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void vm_wait5(int n) {
#define PPMS_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {      // n = DMA pieces per thread and window (1..3 for the 512-thread gather of M = 256, 2..8 for M = 128 / GEMM-mode windows)
        PPMS_VMW(1) PPMS_VMW(2) PPMS_VMW(3) PPMS_VMW(4) PPMS_VMW(5) PPMS_VMW(6) PPMS_VMW(7) PPMS_VMW(8) PPMS_VMW(9) PPMS_VMW(10) PPMS_VMW(11)
        PPMS_VMW(12) PPMS_VMW(13) PPMS_VMW(14)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // 0, and any other count: the safe full drain
    }
#undef PPMS_VMW
}

#ifndef CONV5_NOSYNC
#define CONV5_NOSYNC 0       // ablation builds: 1 drops wait + barrier + DMA at the window switches, 2 the DMA, 3 the barrier: wrong results, timing only
#endif
#ifndef CONV5_PRIO
#define CONV5_PRIO 0         // experiment builds: static issue priority for one of the two waves of a SIMD during the K loop (1: waves 0-3, 2: waves 4-7, 3: the 4-block waves)
#endif
#ifndef CONV5_ABL_A
#define CONV5_ABL_A 0        // ablation builds (-DCONV5_ABL_A=1): every k-step loads the FIRST step's weights (L1 hits): wrong results, timing only
#endif
#include "conv5_asm.h"

__global__ __launch_bounds__(512, 2) void conv5_kernel(const ppms_conv pv, const Geo5 g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CONV5_STAMP(0)
    const ppms_conv& p = pv;                       // by value in the kernel arguments (see conv_gemm2.hip)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the role branches below must be scalar
    const int r = lane & 31, h = lane >> 5;
    // wave roles (the two waves of a SIMD, w and w + 4, get 4 + (NBT - 4) pixel blocks)
    const int kg = (g.kgroups == 2) ? (wave >> 2) : 0;
    int wm = (g.kgroups == 2) ? ((wave & 3) >> 1) : (wave & 3);
    const int wn = (g.kgroups == 2) ? ((wave & 1) ^ kg) : (wave >> 2);
    int nbw = wn == 0 ? 4 : g.NBT - 4;             // 32-pixel blocks of this wave
    int blk0 = wn == 0 ? 0 : 4;                    // its first block in the tile
    if (p.M == 192) {
        // THREE 64-cout blocks (round 4: the 190- / 192-cout convs of the motion encoder used to run padded to 256 rows, a quarter of
        // their MFMAs on zeros).  3 x NBT (cout block, pixel block) units go to the 8 waves so that every SIMD pair (w, w + 4) carries 6
        // (NBT = 8) or 6 / 5 / 5 / 5 (NBT = 7) of them instead of 7 or 8: cout block 2 is split 4 + (NBT - 4) over waves 0 and 1,
        // cout blocks 0 and 1 each 3 + (NBT - 5) + 2 over waves (2, 6, 4) and (3, 7, 5).
        const int n5 = g.NBT - 5;
        wm = wave < 2 ? 2 : (wave & 1);                                   // waves 2, 4, 6 -> block 0; 3, 5, 7 -> block 1
        nbw = wave == 0 ? 4 : wave == 1 ? g.NBT - 4 : wave < 4 ? 3 : wave < 6 ? 2 : n5;
        blk0 = wave == 0 ? 0 : wave == 1 ? 4 : wave < 4 ? 0 : wave < 6 ? g.NBT - 2 : 3;
    }
    const int GT = NT5 / g.kgroups;                // threads of a K-group (they gather the group's windows)
    const int gt = tid - kg * GT;
    const int gwave = gt >> 6;
    int tile = blockIdx.x;
    const int tx = tile % g.tiles_x;
    tile /= g.tiles_x;
    const int ty = tile % g.tiles_y;
    const int tf = tile / g.tiles_y;
    const int x0 = tx * g.C, y0 = ty * g.R;
    const int H = p.H, W = p.W, T = p.T;
    const int HW = H * W;
    const int ht = p.kt >> 1, hy = p.kh >> 1;
    const int wbytes = g.npieces * (GT * 16);      // one window buffer: whole DMA pieces of the group (>= Wr * 64)
    char* const wbase = smem + kg * 2 * wbytes;    // this K-group's pair of window buffers

    // ---- window slots: LDS piece q = gt + i*GT (lane-linear destination); row = q >> 2, position q & 3 ------------------------
    // a 64-B row holds [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] of one pixel's 16-channel chunk, chunk c stored at position
    // c ^ ((row >> 2) & 3); GT is a multiple of 256, so ((row >> 2) & 3) = (gt >> 4) & 3 for every piece of a thread
    // per slot ONE register: pixel offset at (dt, dy) = 0 in the low 22 bits (the volume has < 2^22 pixels), y + 512 in the high 10
    // (the halo puts y in [-7, H + 7]); -1: column outside the image / slot unused
    int sl[MAXS5];
    const int nq = (g.Wr * 4) << g.logRW;
#pragma unroll
    for (int i = 0; i < MAXS5; ++i) {
        const int q = gt + i * GT;
        sl[i] = -1;
        if (q < nq) {
            const int wrow = q >> (2 + g.logRW);
            const int wy = wrow / g.WRL, wx = wrow - wy * g.WRL;
            const int x = x0 + wx - g.hxw, y = y0 + wy - g.hyw;
            if ((unsigned)x < (unsigned)W && (tf * H + y) * W + x >= 0) sl[i] = ((tf * H + y) * W + x) | ((y + 512) << 22);
        }
    }
    // GEMM mode (RW > 1): a row is RW 64-B blocks; block b of row `row` holds channel chunk (b - row) mod RW (rows that are read
    // together by consecutive lanes then start in different bank quarters), inside a block the 16-B parts are swizzled as above.
    // GT >> (2 + logRW) is a multiple of 16, so row & 15 is the same for every piece of a thread.
    const int row0 = gt >> (2 + g.logRW);
    const int src_sub = (((gt >> 2) & (g.RW - 1)) - row0) & (g.RW - 1);
    const int src_chunk = (gt & 3) ^ ((row0 >> 2) & 3);          // 0,1: hi k0-7 / k8-15;  2,3: lo k0-7 / k8-15
    const int src_plane = src_chunk >> 1, src_k8 = (src_chunk & 1) * 8;

    // every descriptor field the loop needs, fetched once (a descriptor load inside the loop would make the compiler drain
    // vmcnt, i.e. the A-fragment prefetch, at every window switch)
    const bf16_t* const sp0 = (const bf16_t*)(src_plane ? p.seg[0].lo : p.seg[0].hi);
    const bf16_t* const sp1 = (const bf16_t*)(src_plane ? p.seg[p.nseg - 1].lo : p.seg[p.nseg - 1].hi);
    const int ld0 = p.seg[0].ld, ld1 = p.seg[p.nseg - 1].ld;
    const int kh_ = p.kh;
    const char* zpage = (const char*)g_zero_page5;
    asm volatile("" : "+s"(zpage));                               // keep the address in registers (else: one GOT load per DMA piece)
    auto dma_b = [&](int win, int buf) {                          // win = rowstep * nchunk + chunk
        const int rowstep = win / g.nchunk, chunk = win - rowstep * g.nchunk;
        int dy = 0, dt;
        if (g.rdy) {
            const int ky = rowstep % kh_;
            dy = ky - hy;
            dt = rowstep / kh_ - ht;
        } else {
            dt = rowstep - ht;
        }
        const int sg = (chunk >= g.n0) ? 1 : 0;
        const int c0 = (((chunk - (sg ? g.n0 : 0)) << g.logRW) + src_sub) * 16 + src_k8;
        const bf16_t* sp = sg ? sp1 : sp0;
        const int ld = sg ? ld1 : ld0;
        const bool tok = (unsigned)(tf + dt + p.t_halo) < (unsigned)(T + 2 * p.t_halo);
        const int shift = (dt * H + dy) * W;
        char* d = wbase + buf * wbytes + gwave * 1024;
#pragma unroll
        for (int i = 0; i < MAXS5; ++i) {
            if (i < g.npieces) {                                  // uniform: every wave issues every piece (lanes past the window
                                                                  // read the zero page into the buffer's padding), so all waves count alike
                const int so = sl[i] & 0x3fffff, sy = ((unsigned)sl[i] >> 22) - 512;
                const bool ok = tok && sl[i] != -1 && (unsigned)(sy + dy) < (unsigned)H;   // (a valid slot has y + 512 < 1023)
                const void* ps = ok ? (const void*)(sp + (int64_t)(so + shift) * ld + c0) : (const void*)zpage;
                dma16_5(ps, d + i * GT * 16);
            }
        }
    };

    // ---- A fragments: packed [k16-step][M/64][4 frags: mb0 hi, mb0 lo, mb1 hi, mb1 lo][64 lanes][16 B] -----------------------
    const int mblocks = p.M >> 6;
    const char* abase = (const char*)p.w;                                       // wave-uniform part kept scalar
    const unsigned avoff = (unsigned)(wm * 4096 + lane * 16);
    const int64_t astep = (int64_t)mblocks * 4096;
    u32x4 areg[DEPTH5][4];   // (read-write asm operands below: the first 'read' is of an undefined value, on purpose -- see conv5_asm.h)
    auto load_a = [&](u32x4 (&st)[4], int ks) {
        const char* sb = abase + (int64_t)ks * astep;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(st[0]) : "v"(avoff), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "+v"(st[1]) : "v"(avoff), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "+v"(st[2]) : "v"(avoff), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "+v"(st[3]) : "v"(avoff), "s"(sb) : "memory");
    };

    // ---- B-operand rows of this lane's pixel blocks (at sweep tap 0) ---------------------------------------------------------
    int brow[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int pid = (blk0 + (nb < nbw ? nb : 0)) * 32 + r;   // (a 3-block wave never reads its 4th entry)
        brow[nb] = (pid >> g.logC) * g.WRL + (pid & (g.C - 1));
    }

    f32x16 acc[2][4];                              // (zeroed per path below: a 3-block wave must not carry two dead zero tuples)

    // temporal taps outside the readable frames contribute zeros: skip them (contiguous kz range)
    const int kz0 = (ht - tf - p.t_halo) > 0 ? (ht - tf - p.t_halo) : 0;
    const int kz1 = (ht + T + p.t_halo - 1 - tf) < (p.kt - 1) ? (ht + T + p.t_halo - 1 - tf) : (p.kt - 1);
    const int rows_per_kz = g.rdy ? p.kh : 1;
    int win0 = kz0 * rows_per_kz * g.nchunk + kg;                       // this K-group's first window; it takes every kgroups-th one
    int nwin = (kz1 + 1 - kz0) * rows_per_kz * g.nchunk / g.kgroups;
    if (g.nslice > 1) {                                                 // K-sliced launch: this workgroup's share of those windows
        const int i0 = (int)blockIdx.y * nwin / g.nslice, i1 = ((int)blockIdx.y + 1) * nwin / g.nslice;
        win0 += g.kgroups * i0;
        nwin = i1 - i0;                                                 // (>= 1: the host keeps nslice <= the smallest window count)
    }
    const int wstride = g.kgroups;
    const int nsteps = nwin * g.nsweep;

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)wbase;
    // LDS byte address of the hi fragment (chunk h) of pixel block nb at window-row offset TROW of buffer BUF; the lo fragment
    // (chunk 2 + h) of the same row is that address ^ 32
    // (GEMM mode: SUB = the chunk of the row this step multiplies, stored in block (SUB + row) mod RW; sweep modes: RW = 1, block 0)
#define CONV5_ADDR(DST, NBW, BUF, TROW, SUB)                                                                                   \
    _Pragma("unroll") for (int nb = 0; nb < (NBW); ++nb) {                                                                      \
        const int row = brow[nb] + (TROW);                                                                                     \
        DST[nb] = lds0 + (BUF) * wbytes + ((row * 64) << g.logRW) + ((((SUB) + row) & (g.RW - 1)) << 6) + ((h ^ ((row >> 2) & 3)) << 4); \
    }
    // One k16-step with static A stage U and static block count NBW.  Pipeline:
    //   top:     wait for A(jj) (requested during the previous step's group 1) and for the hi fragments (previous step's group 3)
    //   group 1: a_lo x b_hi MFMAs || A loads of step jj + 1 (into the other register stage)
    //   group 2: a_hi x b_hi MFMAs || requests for this step's lo fragments
    //   tap state -> next step; at the end of a window: wait for the next window's pieces, barrier, request the window after next
    //   group 3: requests for the next step's hi fragments, wait for the lo fragments, a_hi x b_lo MFMAs
    // vmcnt: at the top nothing but a window's pieces can be younger than A(jj) -- when the previous step switched windows (tap
    // index 0); at a switch only A(jj + 1) is younger than the awaited window's pieces.
#define CONV5_STEP(U, NBW, JJ, MORE)                                                                                               \
    {                                                                                                                          \
        const int jj = (JJ);                                                                                                   \
        const int ahead = nsteps - 1 - jj;                                                                                     \
        const int lz = (int)((unsigned)(g.lz0 - 1 - wchunk) >> 31); /* 1: this step's window (chunk >= lz0) has an all-zero lo plane; shift, not ?: -- a select lands in a VGPR */\
        vm_wait5((sw == 0 && w > 0 && w + 1 < nwin) ? g.npieces : ((jj == 0 && nwin > 1) ? g.npieces : 0));                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        /* (every MFMA group is unconditional straight-line code: an if / else around asm groups that redefine the eight      */ \
        /*  accumulator tuples makes the register allocator copy and spill them; the last step simply re-requests data)        */ \
        mfma_group1<NBW, MORE>(acc, areg[U][1], areg[U][3], bh, areg[(U) ^ 1], avoff, abase + (int64_t)(CONV5_ABL_A ? 0 : la_ks) * astep);         \
        if (ahead >= 2) {                                                                                                      \
            ++la_ks;                                                                                                           \
            if (++la_s == g.nsweep) la_s = 0, la_ks += (wstride - 1) * g.nsweep; /* next window of this K-group */             \
        }                                                                                                                      \
        {                                                                                                                      \
            unsigned adr_lo[4];                                                                                                \
            _Pragma("unroll") for (int nb = 0; nb < (NBW); ++nb) adr_lo[nb] = adr_hi[nb] ^ 32u;                                \
            mfma_group2<NBW>(acc, areg[U][0], areg[U][2], bh, areg[U][1], areg[U][3], bx, adr_lo);                                                   \
        }                                                                                                                      \
        if (++sw == g.nsweep) {                                                                                                \
            sw = swx = trow = 0;                                                                                               \
            wchunk += wstride;                                                                                                 \
            if (wchunk >= g.nchunk) wchunk -= g.nchunk;                                                                        \
            if (ahead > 0) {                                                                                                   \
                if (CONV5_NOSYNC != 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                        \
                if (CONV5_NOSYNC != 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
                if (CONV5_NOSYNC != 1 && CONV5_NOSYNC != 3) __builtin_amdgcn_s_barrier();                                      \
                if (CONV5_NOSYNC != 1 && CONV5_NOSYNC != 2) if (w + 2 < nwin) dma_b(win0 + wstride * (w + 2), w & 1);          \
            }                                                                                                                  \
            ++w;                                                                                                               \
        } else {                                                                                                               \
            trow += g.trow_inc;                                                                                                \
            if (++swx == g.swx_n) {                                                                                            \
                swx = 0;                                                                                                       \
                trow += g.row_jump;                                                                                            \
            }                                                                                                                  \
        }                                                                                                                      \
        if (ahead > 0) { CONV5_ADDR(adr_hi, NBW, w & 1, trow, sw) }                                                              \
        mfma_group3<NBW, MORE>(acc, areg[U][0], areg[U][2], bh, areg[U][1], areg[U][3], bx, adr_hi, lz);                                                      \
    }
    // the whole K loop for a static block count
#define CONV5_LOOP(NBW)                                                                                                        \
    {                                                                                                                          \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < (NBW); ++b) {                      \
            acc[a][b] = (f32x16){0};                                                                                           \
            /* materialise the zeros HERE: the MFMAs of the loop are inline asm, so the compiler pads no VALU-write ->       */ \
            /* MFMA-read wait states in front of them; left alone it sinks these moves right in front of the first MFMA      */ \
            asm volatile("" : "+v"(acc[a][b]));                                                                                \
        }                                                                                                                      \
        dma_b(win0, 0);                                                                                                        \
        load_a(areg[0], win0 * g.nsweep);                                                                                      \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); /* window 0 landed (everything but the 4 A loads) */                  \
        __builtin_amdgcn_s_barrier();                                                                                          \
        if (nwin > 1) dma_b(win0 + wstride, 1);                                                                                \
        int sw = 0, swx = 0, trow = 0, w = 0;                                                                                  \
        int wchunk = win0 % g.nchunk; /* chunk index of the current window (it advances by wstride, modulo nchunk) */           \
        int la_s = 1, la_ks = win0 * g.nsweep + 1; /* tap / packed k16-step of the next A load (nsweep >= 2); stops at the last step */ \
        bf16x8 bh[4], bx[2];                                                                                                  \
        unsigned adr_hi[4];                                                                                                    \
        CONV5_ADDR(adr_hi, NBW, 0, 0, 0)                                                                                       \
        _Pragma("unroll") for (int nb = 0; nb < (NBW); ++nb)                                                                    \
            asm volatile("ds_read_b128 %0, %1" : "+v"(bh[nb]) : "v"(adr_hi[nb]) : "memory");                                   \
        int j = 0;                                                                                                             \
        for (; j + 1 < nsteps; j += DEPTH5) {                                                                                  \
            CONV5_STEP(0, NBW, j, true)                                                                                        \
            CONV5_STEP(1, NBW, j + 1, true)                                                                                     \
        }                                                                                                                      \
        if (j < nsteps) CONV5_STEP(0, NBW, j, false) /* odd step count */                                                      \
        /* the last step's re-requests land in dead registers: drain them before the registers are reused; the last MFMAs' */  \
        /* results need 12+ wait states before any non-MFMA reader */                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");                                    \
        /* every register a load of the loop targets stays allocated up to here: a load whose result is dead would have its   */ \
        /* destination handed to the next value while the data is still in flight                                            */ \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(areg[0][k]), "v"(areg[1][k]), "v"(bh[k]));         \
    }
    static_assert(DEPTH5 == 2, "two A stages: U and U ^ 1; the step loop is unrolled twice");
    CONV5_STAMP(1)
#if CONV5_PRIO == 1
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
#elif CONV5_PRIO == 2
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#elif CONV5_PRIO == 3
    if (nbw == 4) __builtin_amdgcn_s_setprio(1);
#endif
    if (nbw == 4) CONV5_LOOP(4) else if (nbw == 3) CONV5_LOOP(3) else CONV5_LOOP(2)
#if CONV5_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    CONV5_STAMP(2)
#undef CONV5_LOOP
#undef CONV5_STEP
#undef CONV5_ADDR
    __syncthreads();                               // the window buffers become the reduction / epilogue staging areas

    // ---- M = 128: sum the two K-groups' partial tiles through LDS; BOTH groups then finish the tile: the two waves that hold partials of
    // the same (64 couts x nbw blocks) tile swap halves -- group 1 hands over its blocks 0-1, group 0 its blocks 2.. -- and each runs the
    // epilogue on the blocks it received (8 instead of 4 waves in the epilogue; a + b is the same fp32 value in either order) -------------
    int nb_lo = 0, nb_hi = nbw;
    if (g.kgroups == 2) {
        float* red = (float*)smem + (wm * 2 + wn) * (128 * 64);              // one 32 KiB slot per (wm, wn): 128 registers x 64 lanes
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                if ((nb < 2) == (kg == 1)) {                                 // (uniform: nb is static, kg is per wave) the blocks the OTHER group finishes
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v4 = {acc[mb][nb][4 * q4], acc[mb][nb][4 * q4 + 1], acc[mb][nb][4 * q4 + 2], acc[mb][nb][4 * q4 + 3]};
                        *(f32x4*)(red + (((mb * 4 + nb) * 4 + q4) * 64 + lane) * 4) = v4;
                    }
                }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                if ((nb < 2) == (kg == 0)) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v4 = *(const f32x4*)(red + (((mb * 4 + nb) * 4 + q4) * 64 + lane) * 4);
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) acc[mb][nb][4 * q4 + e2] += v4[e2];
                    }
                }
        __syncthreads();                           // the exchange area is reused as the epilogue's staging patches
        nb_lo = kg ? 2 : 0;
        nb_hi = kg ? nbw : 2;
    }

    CONV5_STAMP(3)
    // ---- epilogue: accumulators -> wave-private LDS patch [32 px][64 couts] -> 8 couts of one pixel per lane -------
    const int cblock = wm * 64;
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
    // the row loop is instantiated per epilogue class (conv_epilogue.h): each instance holds exactly the loads its class needs -- the plain
    // one none at all -- and fetches the extra operands of a GROUP of G 8-row steps before it finishes (stores) them, so the wait for
    // them (which on this one in-order counter is also a wait for the previous group's stores) comes once per group, not per step
    auto rows = [&](auto cls_tag, auto grp_tag) {
        constexpr int CLS = decltype(cls_tag)::value, G = decltype(grp_tag)::value;
    #pragma unroll 1
        for (int nb = nb_lo; nb < nb_hi; ++nb) {       // (not unrolled: code size; the selects keep every accumulator index static)
    #pragma unroll
            for (int mb = 0; mb < 2; ++mb)
    #pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 a4;
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[mb][0][4 * gq + j];
    #pragma unroll
                        for (int k = 1; k < 4; ++k) x = (nb == k) ? acc[mb][k][4 * gq + j] : x;
                        a4[j] = x;
                    }
                    if (CLS == EPI_CLS_ANY && e.out_vt != nullptr) {                     // pixel-major V^T straight from the accumulator layout
                        const int pid = (blk0 + nb) * 32 + r;
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
            if (nb == nb_lo) { CONV5_STAMP(5) }
    #pragma unroll 1
            for (int it0 = 0; it0 < 4; it0 += G) {
                if (nb == nb_lo && it0 == G) { CONV5_STAMP(6) }
                row8_aux aux[G];
                int64_t pixg[G];
                bool okg[G];
    #pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const int pid = (blk0 + nb) * 32 + (it0 + gi) * 8 + (lane >> 3);
                    const int px = x0 + (pid & (g.C - 1)), py = y0 + (pid >> g.logC);
                    okg[gi] = px < W && py < H;
                    pixg[gi] = (int64_t)(tf * H + py) * W + px;
                    if (okg[gi] && g.nslice == 1) row8_fetch<CLS>(e, pixg[gi], cbase + q * 8, aux[gi]);
                }
    #pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    float v[8];
                    stage_read8(stg, (it0 + gi) * 8 + (lane >> 3), q, v);
                    if (okg[gi]) {
                        if (g.nslice > 1) {                              // raw partial sums; bias and the fused epilogue run in the reduce kernel
                            float* pp = g.part + ((int64_t)blockIdx.y * g.P + pixg[gi]) * p.M + cblock + q * 8;
                            gst<f32x4>(pp, (f32x4){v[0], v[1], v[2], v[3]});
                            gst<f32x4>(pp + 4, (f32x4){v[4], v[5], v[6], v[7]});
                        } else {
    #pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += b8[j];
                            row8_finish<CLS>(e, v, pixg[gi], cbase + q * 8, HW, aux[gi]);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (nb == nb_lo) { CONV5_STAMP(7) }
        }
    };
    using I0 = std::integral_constant<int, EPI_CLS_PLAIN>;
    using I1 = std::integral_constant<int, EPI_CLS_PRE>;
    using I2 = std::integral_constant<int, EPI_CLS_AUX>;
    using I3 = std::integral_constant<int, EPI_CLS_GRU>;
    using I4 = std::integral_constant<int, EPI_CLS_ANY>;
    using I5 = std::integral_constant<int, EPI_CLS_AUXPRE>;
    using G2 = std::integral_constant<int, 2>;
    using G1 = std::integral_constant<int, 1>;
    using G4 = std::integral_constant<int, 4>;
    const int cls = g.nslice > 1 ? (int)EPI_CLS_PLAIN : epilogue_class(e);
    if (cls == EPI_CLS_PLAIN) rows(I0{}, G1{});
    else if (cls == EPI_CLS_PRE) rows(I1{}, G4{});
    else if (cls == EPI_CLS_AUX) rows(I2{}, G4{});
    else if (cls == EPI_CLS_GRU) rows(I3{}, G1{});        // (groups of two rows were measured: 24 more live registers spill, 30.7 -> 38 us)
    else if (cls == EPI_CLS_AUXPRE) rows(I5{}, G2{});
    else rows(I4{}, G1{});
    CONV5_STAMP(4)
}

// tile shape / window geometry for a descriptor; picks (NBT, C) with the best chip fill; false when nothing fits
// compute units of the current device (the planner deals one workgroup per CU: 256 on MI355X); queried once per process and device
static bool plan5(const ppms_conv* d, Geo5& g, int force_nbt = 0, bool sliced = false) {
    const bool gemm = d->kw == 1 && d->kh == 1;                              // no spatial sweep: windows of 16 RW channels, no halo
    const int mode = gemm ? 3 : (d->kw > 1 && d->kh > 1) ? 2 : (d->kw > 1 ? 0 : 1);      // 0: x sweep, 1: y sweep, 2: 2-D sweep, 3: GEMM
    const int hx = (mode == 0 || mode == 2) ? d->kw - 1 : 0, hy = (mode == 1 || mode == 2) ? d->kh - 1 : 0;   // window halo (total) in x / y
    const int kgroups = d->M == 128 ? 2 : 1;
    const int GT = NT5 / kgroups;
    const int RW = gemm ? (kgroups == 1 ? 4 : 2) : 1;
    int nchunk = 0;
    for (int s = 0; s < d->nseg; ++s) {
        if (d->seg[s].c % (16 * RW)) return false;
        nchunk += d->seg[s].c / (16 * RW);
    }
    double best = -1.0;
    int bestC = -1, bestN = 0, bestWr = 0;
    for (int nbt = 7; nbt <= 8; ++nbt) {
        if (force_nbt && nbt != force_nbt) continue;
        for (int C = 16; C <= 128; C *= 2) {
            if ((32 * nbt) % C) continue;
            const int R = 32 * nbt / C;
            const int Wr = ((R + hy) * (C + hx) + 15) / 16 * 16;
            if (Wr * RW > 1024) continue;                                     // one window buffer <= 64 KiB (sweep modes: Wr <= 384 rows of 64 B fit anyway)
            if (!gemm && Wr > 384) continue;
            const int64_t tiles = (int64_t)((d->W + C - 1) / C) * ((d->H + R - 1) / R) * d->T;
            // useful pixels per CU-slot-round: rounds of 256 workgroups (one per CU), each costing nbt blocks
            // (K-sliced launches of small maps fill the chip through the slices: there only the ragged tile edges count)
            const int64_t cus = ppms_num_cus();
            const double eff = sliced ? (double)d->T * d->H * d->W / ((double)tiles * 32 * nbt)
                                      : (double)d->T * d->H * d->W / ((double)((tiles + cus - 1) / cus) * cus * 32 * nbt);
            if (eff > best + 1e-9 || (eff > best - 1e-9 && Wr < bestWr)) best = eff, bestC = C, bestN = nbt, bestWr = Wr;
        }
    }
    if (bestC < 0) return false;
    g.NBT = bestN;
    g.C = bestC;
    g.R = 32 * bestN / bestC;
    g.logC = 0;
    while ((1 << g.logC) < g.C) ++g.logC;
    g.tiles_x = (d->W + g.C - 1) / g.C;
    g.tiles_y = (d->H + g.R - 1) / g.R;
    g.WRL = g.C + hx;
    g.Wr = bestWr;
    g.hxw = hx >> 1;
    g.hyw = hy >> 1;
    g.RW = RW;
    g.logRW = RW == 4 ? 2 : (RW == 2 ? 1 : 0);
    g.trow_inc = gemm ? 0 : 1;
    if (mode == 3) {
        g.swx_n = 1 << 30, g.row_jump = 0, g.nsweep = RW, g.rdy = 0;
    } else if (mode == 0) {
        g.swx_n = d->kw, g.row_jump = 0, g.nsweep = d->kw, g.rdy = 1;
    } else if (mode == 1) {
        g.swx_n = 1, g.row_jump = g.WRL - 1, g.nsweep = d->kh, g.rdy = 0;
    } else {
        g.swx_n = d->kw, g.row_jump = g.WRL - d->kw, g.nsweep = d->kh * d->kw, g.rdy = 0;
    }
    g.nchunk = nchunk;
    g.n0 = d->seg[0].c / (16 * RW);
    g.lz0 = (d->lo_zero_from > 0 && d->lo_zero_from % (16 * RW) == 0) ? d->lo_zero_from / (16 * RW) : nchunk;
    g.kgroups = kgroups;
    g.npieces = (g.Wr * 4 * RW + GT - 1) / GT;
    g.nslice = 1;
    g.part = nullptr;
    g.P = (int64_t)d->T * d->H * d->W;
    return g.npieces <= MAXS5 && nchunk % kgroups == 0;
}

}  // namespace

// returns 1 when this kernel serves the convolution: M == 256 or 128 (all couts in one workgroup), a spatial sweep of >= 3 taps,
// 16-channel-aligned segments, a halo'd window that fits, and at least ~a workgroup per CU
// the packed window slots hold a pixel offset in 22 bits and a row index in 10 (conv5_launch checks the same)
static bool conv5_volume_fits(const ppms_conv* d) { return (int64_t)d->T * d->H * d->W < (1ll << 22) && d->H < 480; }

extern "C" int ppms_conv_gemm5_applicable(const ppms_conv* d) {
    if (d == nullptr || (d->M != 256 && d->M != 192 && d->M != 128) || d->m_split % 64 != 0 || d->nseg < 1 || d->nseg > 2) return 0;
    if (!conv5_volume_fits(d)) return 0;
    for (int s = 0; s < d->nseg; ++s)
        if (d->seg[s].c <= 0 || d->seg[s].c % 16 != 0) return 0;
    Geo5 g;
    if (!plan5(d, g)) return 0;                     // (kh = kw = 1: GEMM mode, segments in multiples of 64 / 32 channels)
    if (d->M == 192 && (d->epi[0].out_vt != nullptr || (d->m_split < d->M && d->epi[1].out_vt != nullptr))) return 0;
    return (int64_t)g.tiles_x * g.tiles_y * d->T >= ppms_num_cus() * 25 / 32 ? 1 : 0;   // (200 of 256) fewer workgroups than CUs: conv_gemm2's K slicing fills the chip better
}

// smallest number of windows any (tile, K-group) pair of the launch sweeps: frames at the ends of the volume skip the temporal taps
// that fall outside it
static int min_windows5(const ppms_conv* d, const Geo5& g) {
    const int ht = d->kt >> 1;
    int kz_min = d->kt;
    for (int tf = 0; tf < d->T; ++tf) {
        const int kz0 = (ht - tf - d->t_halo) > 0 ? (ht - tf - d->t_halo) : 0;
        const int kz1 = (ht + d->T + d->t_halo - 1 - tf) < (d->kt - 1) ? (ht + d->T + d->t_halo - 1 - tf) : (d->kt - 1);
        if (kz1 + 1 - kz0 < kz_min) kz_min = kz1 + 1 - kz0;
    }
    return kz_min * (g.rdy ? d->kh : 1) * g.nchunk / g.kgroups;
}

// Small maps (fewer tiles than CUs): how many grid-level K slices let this kernel fill the chip.  0: not applicable / not worth it.
extern "C" int ppms_conv_gemm5_slices(const ppms_conv* d) {
    if (d == nullptr || (d->M != 256 && d->M != 128) || d->m_split % 64 != 0 || d->nseg < 1 || d->nseg > 2) return 0;
    if (d->kw == 1 && d->kh == 1) return 0;
    if (!conv5_volume_fits(d)) return 0;
    if (d->epi[0].out_vt != nullptr || (d->m_split < d->M && d->epi[1].out_vt != nullptr)) return 0;   // V^T is written from the accumulators
    for (int s = 0; s < d->nseg; ++s)
        if (d->seg[s].c <= 0 || d->seg[s].c % 16 != 0) return 0;
    Geo5 g;
    if (!plan5(d, g, 0, true) || g.nsweep < 3) return 0;
    const int64_t tiles = (int64_t)g.tiles_x * g.tiles_y * d->T;
    const int cus = ppms_num_cus();
    if (tiles >= cus * 25 / 32 || tiles < 8) return 0;
    int ns = (int)((cus + tiles - 1) / tiles);                        // one workgroup per CU, one round
    if (tiles * ns > cus * 9 / 8) --ns;
    const int wmin = min_windows5(d, g);
    if (ns > wmin) ns = wmin;
    if (ns > 16) ns = 16;
    return ns >= 2 ? ns : 0;
}

static int conv5_launch(const ppms_conv* d, const ppms_conv* dev_desc, int nbt, int nslice, float* part, void* stream);
#ifdef PPMS_CONV5_TIMING
extern "C" void ppms_debug_conv5_timing(long long* p) { g_conv5_dbg = p; }      // debug builds only (tools/conv5_phase_probe.py)
#endif

extern "C" int ppms_conv_gemm5(const ppms_conv* d, const ppms_conv* dev_desc, int nbt, void* stream) {
    return conv5_launch(d, dev_desc, nbt, 1, nullptr, stream);
}

// K-sliced form for small maps: nslice workgroups share each output tile (each sweeps its share of the windows and writes fp32 partial
// sums to `workspace`, ppms_conv_gemm2_slice_workspace_bytes), then the slice-reduce kernel of conv_gemm2.hip sums them in slice order
// (bit-reproducible) and runs the fused epilogue.
extern "C" int ppms_conv_gemm5_sliced(const ppms_conv* d, const ppms_conv* dev_desc, int nbt, int nslice, void* workspace, void* stream) {
    PPMS_REQUIRE(nslice >= 1 && nslice <= 16, "conv_gemm5_sliced: nslice=%d", nslice);
    if (nslice == 1) return conv5_launch(d, dev_desc, nbt, 1, nullptr, stream);
    PPMS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "conv_gemm5_sliced: workspace missing or not 16-B aligned");
    const int rc = conv5_launch(d, dev_desc, nbt, nslice, (float*)workspace, stream);
    if (rc != 0) return rc;
    return ppms_launch_slice_reduce(d, dev_desc, (const float*)workspace, nslice, stream);
}

static int conv5_launch(const ppms_conv* d, const ppms_conv* dev_desc, int nbt, int nslice, float* part, void* stream) {
    PPMS_REQUIRE(d != nullptr && dev_desc != nullptr, "conv_gemm5: null descriptor");
    PPMS_REQUIRE(nbt == 0 || nbt == 7 || nbt == 8, "conv_gemm5: nbt must be 0 (choose), 7 or 8");
    PPMS_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv_gemm5: nseg=%d", d->nseg);
    PPMS_REQUIRE(d->groups <= 1, "conv_gemm5: a grouped convolution (groups=%d) is served by ppms_conv_gemm6 only", d->groups);
    PPMS_REQUIRE(d->T > 0 && d->H > 0 && d->W > 0, "conv_gemm5: bad volume %dx%dx%d", d->T, d->H, d->W);
    PPMS_REQUIRE((d->M == 256 || d->M == 192 || d->M == 128) && d->m_split % 64 == 0, "conv_gemm5: M=%d must be 128, 192 or 256", d->M);
    PPMS_REQUIRE(d->M != 192 || nslice == 1, "conv_gemm5: the 192-cout layout has no K-sliced form");
    PPMS_REQUIRE((d->kt & 1) && (d->kh & 1) && (d->kw & 1) && d->kw <= 15 && d->kh <= 15, "conv_gemm5: odd kernel extents <= 15");
    PPMS_REQUIRE(d->w != nullptr && d->bias != nullptr, "conv_gemm5: weights/bias missing");
    PPMS_REQUIRE(d->t_halo >= 0 && d->t_halo <= 8, "conv_gemm5: t_halo=%d", d->t_halo);
    PPMS_REQUIRE(conv5_volume_fits(d), "conv_gemm5: volume too large for the packed window slots (< 2^22 pixels, H < 480)");
    for (int s = 0; s < d->nseg; ++s) {
        PPMS_REQUIRE(d->seg[s].hi && d->seg[s].lo && d->seg[s].c > 0 && d->seg[s].c % 16 == 0 && d->seg[s].ld % 8 == 0,
                     "conv_gemm5: segment %d needs hi/lo planes, c %% 16 == 0 and ld %% 8 == 0", s);
        PPMS_REQUIRE(((uintptr_t)d->seg[s].hi & 15) == 0 && ((uintptr_t)d->seg[s].lo & 15) == 0, "conv_gemm5: segment %d not 16-B aligned", s);
    }
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && d->m_split >= d->M) break;
        PPMS_REQUIRE(e.n_valid > 0, "conv_gemm5: epilogue %d has n_valid=%d", hlf, e.n_valid);
        PPMS_REQUIRE(e.pre_f32 == nullptr || (e.n_valid % 4 == 0 && e.pre_f32_ld % 4 == 0), "conv_gemm5: pre_f32 needs n_valid and pre_f32_ld to be multiples of 4");
        {
            const char* why = epilogue_row8_check(e);
            PPMS_REQUIRE(why == nullptr, "conv_gemm5: epilogue %d: %s", hlf, why ? why : "");
        }
        if (e.out_sp.hi) PPMS_REQUIRE(e.out_sp.lo && e.out_sp.ld % 4 == 0, "conv_gemm5: epilogue %d SP output misaligned", hlf);
        if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU)
            PPMS_REQUIRE(e.aux_sp.hi && e.aux_sp.lo && e.aux_sp.ld % 4 == 0, "conv_gemm5: epilogue %d needs aux_sp", hlf);
        if (e.kind == PPMS_EPI_GRU) PPMS_REQUIRE(e.aux_f32 != nullptr, "conv_gemm5: GRU epilogue needs z");
    }
    Geo5 g;
    PPMS_REQUIRE(plan5(d, g, nbt, nslice > 1), "conv_gemm5: no tile shape fits the LDS window");
    PPMS_REQUIRE(g.nsweep >= 2, "conv_gemm5: a window must serve at least 2 k-steps");
    if (nslice > 1) {
        PPMS_REQUIRE(d->epi[0].out_vt == nullptr && (d->m_split >= d->M || d->epi[1].out_vt == nullptr), "conv_gemm5: sliced launch cannot write out_vt");
        PPMS_REQUIRE(nslice <= min_windows5(d, g), "conv_gemm5: %d slices for %d windows", nslice, min_windows5(d, g));
        g.nslice = nslice;
        g.part = part;
    }
    PPMS_REQUIRE(g.npieces <= MAXS5 && g.npieces >= 1, "conv_gemm5: window of %d rows needs too many DMA pieces", g.Wr);
    const int ntiles = g.tiles_x * g.tiles_y * d->T;
    size_t lds = (size_t)2 * g.kgroups * g.npieces * (NT5 / g.kgroups) * 16;        // the K-groups' pairs of window buffers (GEMM mode: 2 x 64 KiB)
    if (g.kgroups == 2 && lds < (size_t)4 * 128 * 64 * 4) lds = (size_t)4 * 128 * 64 * 4;   // K-group exchange: 4 x 32 KiB
    if (lds < (size_t)8 * STG_WAVE) lds = (size_t)8 * STG_WAVE;                  // the epilogue's transposition patches
    PPMS_REQUIRE(lds <= 160 * 1024, "conv_gemm5: LDS budget exceeded (%zu B)", lds);
    static ppms_device_once once;
    once.run([] { (void)hipFuncSetAttribute((const void*)conv5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
#ifdef PPMS_CONV5_TIMING
    g.dbg = g_conv5_dbg;
#endif
    hipLaunchKernelGGL(conv5_kernel, dim3(ntiles, g.nslice), dim3(NT5), lds, (hipStream_t)stream, *d, g);
    return ppms_check_launch("conv_gemm5");
}
