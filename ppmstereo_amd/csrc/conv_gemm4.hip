// Implicit-GEMM convolution, barrier-free k-loop variant (gfx950).
//
// Same math, descriptor and epilogues as conv_gemm2/3 (bf16x3 split MFMA: hi*hi + hi*lo + lo*hi, fp32 accumulate).  What
// bounded conv_gemm3 was its k-step structure: every 32-channel step ended in `s_waitcnt vmcnt(0)` + `s_barrier` (the weight
// tile went through LDS), each wave paid the issue cost of four LDS-DMA pieces per step, and a window switch drained the
// whole workgroup for a global -> LDS round trip.  Here
//   * the WEIGHTS never touch LDS: they are packed on the host in MFMA-fragment order ([k16-step][64-cout block][frag][lane]
//     16 B), so a wave fetches its own A fragments with four coalesced 1 KiB global loads per k16-step straight into
//     registers (L2 / L1 served: the two waves that share a cout block read the same lines), four steps ahead, waited for
//     with counted `s_waitcnt vmcnt` -- no barrier, no LDS write / read, no DMA issue cost for this operand;
//   * the ACTIVATION window holds 16 channels (hi and lo interleaved in one 64-B row) and is double buffered: the window of
//     the next (row-step, 16-channel chunk) is gathered by LDS-DMA while the taps sweep the current one;
//   * so the loop synchronises once per WINDOW (one `s_barrier` per 3 - 15 k16-steps), not once per step.
// Workgroup = 4 waves (wm, wn): 128 couts x (64 NB) pixels, wave tile 64 couts x 32 NB pixels (2 x NB MFMA 32x32x16 tiles).
// NB = 4 (256-pixel tile) for the 256-cout convs, NB = 2 (128-pixel tile) for 128-cout convs (keeps >= 400 workgroups).
// Weights: ppmstereo_amd/packing.py pack_conv4 (k16-step order = (row-step, 16-channel chunk, sweep tap)).
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

constexpr int NT = 256;
constexpr int MAXS4 = 6;                      // window 16-B pieces per thread (window <= 384 rows of 64 B)
constexpr int DEPTH = 3;                      // A-fragment stages in registers (2 steps of prefetch; 4 stages spill at 256 registers)

__device__ __attribute__((aligned(256))) unsigned int g_zero_page4[64];     // zero-initialised: source of padded rows

struct Geo4 {
    int C, R, logC;          // patch R x C pixels, R*C = 64*NB
    int tiles_x, tiles_y;
    int WRL, Wr;             // window row length (pixels) and rows (padded to 16)
    int hxw, hyw;            // halo of the window in x / y
    int swx_n, row_jump;     // sweep: the LDS row advances by 1 per tap and by row_jump more after every swx_n taps
    int nsweep;              // taps swept inside one window (kw, kh or kh*kw)
    int rdy;                 // 1: the row-step index carries a dy (x sweep), 0: only dt
    int nchunk, n0;          // 16-channel chunks per tap (all segments), chunks of segment 0
    int mgroups;             // M / 128
    int npieces;             // DMA pieces per thread and window = ceil(Wr * 4 / 256)
};

__device__ __forceinline__ void dma16_4(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const PPMS_GLOBAL void*)(uintptr_t)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void vm_wait(int n) {
#define PPMS_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    if (n == 8) {                                  // the steady state
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        return;
    }
    switch (n) {
        PPMS_VMW(4) PPMS_VMW(9) PPMS_VMW(10) PPMS_VMW(11) PPMS_VMW(12) PPMS_VMW(13) PPMS_VMW(14)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // 0, and any other count: the safe full drain
    }
#undef PPMS_VMW
}

// PRIO: 0 = no priority changes; 1 = priority 1 from a step's first MFMA to its last; 2 = priority 1 around each MFMA group
#ifndef PPMS_CONV4_PRIO
#define PPMS_CONV4_PRIO 1
#endif
constexpr int PRIO = PPMS_CONV4_PRIO;
// timing ablations (diagnostic builds only: wrong results): skip the A loads / the B fragment reads / the window DMA / the epilogue
#ifndef PPMS_ABL
#define PPMS_ABL 0
#endif
constexpr bool ABL_A = PPMS_ABL & 1, ABL_B = PPMS_ABL & 2, ABL_D = PPMS_ABL & 4, ABL_E = PPMS_ABL & 8;

template <int NB>
__global__ __launch_bounds__(256, 2) void conv4_kernel(const ppms_conv pv, const Geo4 g) {
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
    const int wbytes = g.npieces * (NT * 16);      // one window buffer: whole 4 KiB DMA pieces (>= Wr * 64)

    // ---- window slots: LDS piece q = tid + i*256 (lane-linear destination); row = q >> 2, position q & 3 ---------------------
    // a 64-B row holds [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] of one pixel's 16-channel chunk, chunk c stored at position
    // c ^ ((row >> 2) & 3); q = tid + 256 i  =>  row = (tid >> 2) + 64 i, so ((row >> 2) & 3) = (tid >> 4) & 3 for every i
    int sl_off[MAXS4];          // pixel offset at (dt, dy) = 0, -1: column outside the image / slot unused
    int sl_y[MAXS4];
    const int nq = g.Wr * 4;
#pragma unroll
    for (int i = 0; i < MAXS4; ++i) {
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
    const int src_chunk = (tid & 3) ^ ((tid >> 4) & 3);          // 0,1: hi k0-7 / k8-15;  2,3: lo k0-7 / k8-15
    const int src_plane = src_chunk >> 1, src_k8 = (src_chunk & 1) * 8;

    // every descriptor field the loop needs, fetched once (a descriptor load inside the loop would make the compiler drain
    // vmcnt, i.e. the A-fragment prefetch, at every window top)
    const bf16_t* const sp0 = (const bf16_t*)(src_plane ? p.seg[0].lo : p.seg[0].hi);
    const bf16_t* const sp1 = (const bf16_t*)(src_plane ? p.seg[p.nseg - 1].lo : p.seg[p.nseg - 1].hi);
    const int ld0 = p.seg[0].ld, ld1 = p.seg[p.nseg - 1].ld;
    const int kh_ = p.kh;
    const char* zpage = (const char*)g_zero_page4;
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
        const int c0 = (chunk - (sg ? g.n0 : 0)) * 16 + src_k8;
        const bf16_t* sp = sg ? sp1 : sp0;
        const int ld = sg ? ld1 : ld0;
        const bool tok = (unsigned)(tf + dt + p.t_halo) < (unsigned)(T + 2 * p.t_halo);
        const int shift = (dt * H + dy) * W;
        char* d = smem + buf * wbytes + wave * 1024;
#pragma unroll
        for (int i = 0; i < MAXS4; ++i) {
            if (i < g.npieces) {                                  // uniform: every wave issues every piece (lanes past the window
                                                                  // read the zero page into the buffer's padding), so all waves count alike
                const bool ok = tok && sl_off[i] >= 0 && (unsigned)(sl_y[i] + dy) < (unsigned)H;
                const void* ps = ok ? (const void*)(sp + (int64_t)(sl_off[i] + shift) * ld + c0) : (const void*)zpage;
                dma16_4(ps, d + i * NT * 16);
            }
        }
    };

    // ---- A fragments: packed [k16-step][M/64][4 frags: mb0 hi, mb0 lo, mb1 hi, mb1 lo][64 lanes][16 B] -----------------------
    const int mblocks = p.M >> 6;
    const char* abase = (const char*)p.w + (int64_t)(mgrp * 2) * 4096;          // wave-uniform part kept scalar
    const unsigned avoff = (unsigned)(wm * 4096 + lane * 16);
    const int64_t astep = (int64_t)mblocks * 4096;
    u32x4 areg[DEPTH][4];
    auto load_a = [&](u32x4 (&st)[4], int ks) {
        const char* sb = abase + (int64_t)ks * astep;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(st[0]) : "v"(avoff), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(st[1]) : "v"(avoff), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(st[2]) : "v"(avoff), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(st[3]) : "v"(avoff), "s"(sb) : "memory");
    };

    // ---- B-operand rows of this lane's pixel blocks (at sweep tap 0) ---------------------------------------------------------
    int brow[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pid = wn * (32 * NB) + nb * 32 + r;
        brow[nb] = (pid >> g.logC) * g.WRL + (pid & (g.C - 1));
    }

    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = (f32x16){0};

    // temporal taps outside [0, T) contribute zeros: skip them (contiguous kz range)
    const int kz0 = (ht - tf - p.t_halo) > 0 ? (ht - tf - p.t_halo) : 0;
    const int kz1 = (ht + T + p.t_halo - 1 - tf) < (p.kt - 1) ? (ht + T + p.t_halo - 1 - tf) : (p.kt - 1);
    const int rows_per_kz = g.rdy ? p.kh : 1;
    const int win0 = kz0 * rows_per_kz * g.nchunk;
    const int nwin = (kz1 + 1 - kz0) * rows_per_kz * g.nchunk;
    const int nsteps = nwin * g.nsweep;
    const int ks0 = win0 * g.nsweep;               // packed k16-steps are in (window, tap) order

    // B fragments of one tap from a window buffer: cpos 0 = hi chunks, 2 = lo chunks.  The reads are inline asm so that the
    // LDS counter is managed by hand below (reads return in order: with the NB youngest reads still in flight the older group has
    // arrived) -- the compiler's own bookkeeping across the loop back-edge degenerates to lgkmcnt(0) in front of every MFMA group.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    auto read_b = [&](bf16x8 (&dst)[NB], int buf, int trow_, int cpos) {
        const unsigned bs = lds0 + buf * wbytes;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int row = brow[nb] + trow_;
            const unsigned addr = bs + row * 64 + (((cpos + h) ^ ((row >> 2) & 3)) << 4);
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst[nb]) : "v"(addr) : "memory");
        }
    };
    auto lgkm_wait_older = [&](bool younger_group_in_flight) {     // the older of the (at most) two fragment groups in flight has arrived
        if (younger_group_in_flight) {
            if (NB == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // ---- prologue: window 0, the first two A stages, then window 1 in flight; hi fragments of step 0 ------------------------
    dma_b(win0, 0);
    load_a(areg[0], ks0);
    load_a(areg[1], ks0 + 1);                      // (nsteps >= 3: nsweep >= 3)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // window 0 landed (everything but the 8 A loads)
    __builtin_amdgcn_s_barrier();
    if (nwin > 1) dma_b(win0 + 1, 1);
    int sw = 0, swx = 0, trow = 0, w = 0;
    bf16x8 bh[NB], bl[NB];
    read_b(bh, 0, 0, 0);
    // One k16-step with static A stage U (a macro: the body must be inlined with U a literal so that the register arrays are
    // indexed statically).  Software pipeline inside the step, so that the matrix pipe never waits for LDS:
    //     lo fragments of THIS step are requested at its start and used by its last 8 MFMAs,
    //     hi fragments of the NEXT step are requested after the 16th MFMA (the hi registers are dead by then).
    // When the next step opens a new window, the window switch happens at that point too: wait for this wave's pieces of the
    // next window (older than the 8 youngest A loads), barrier (all waves' pieces landed, all waves done reading the current
    // window: their lo reads are retired by the lgkmcnt), then request the window after next into the buffer just left.
    // vmcnt arithmetic: A(jj) was issued two steps ago; younger ops are A(jj+1), A(jj+2) (4 loads each, when they exist) and --
    // when the last switch lies less than two steps back (tap index < 2) and issued a window -- that window's pieces.
#define CONV4_STEP(U, JJ)                                                                                                      \
    {                                                                                                                          \
        const int jj = (JJ);                                                                                                   \
        const int ahead = nsteps - 1 - jj; /* steps after this one */                                                          \
        if (ahead >= DEPTH - 1 && !ABL_A) load_a(areg[((U) + DEPTH - 1) % DEPTH], ks0 + jj + DEPTH - 1);                       \
        if (!ABL_B) read_b(bl, w & 1, trow, 2);                                                                                \
        if (ABL_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
        else vm_wait(4 * (ahead < DEPTH - 1 ? ahead : DEPTH - 1) + ((sw < DEPTH - 1 && w + 1 < nwin && !ABL_D) ? g.npieces : 0)); \
        lgkm_wait_older(true); /* hi fragments (requested in the previous step) are in; the lo requests may still fly */      \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                               \
        const bf16x8 ah0 = __builtin_bit_cast(bf16x8, areg[U][0]), al0 = __builtin_bit_cast(bf16x8, areg[U][1]);               \
        const bf16x8 ah1 = __builtin_bit_cast(bf16x8, areg[U][2]), al1 = __builtin_bit_cast(bf16x8, areg[U][3]);               \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh[nb], acc[0][nb], 0, 0, 0); \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh[nb], acc[1][nb], 0, 0, 0); \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh[nb], acc[0][nb], 0, 0, 0); \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh[nb], acc[1][nb], 0, 0, 0); \
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        /* advance the tap state to the next step */                                                                          \
        if (++sw == g.nsweep) {                                                                                                \
            sw = swx = trow = 0;                                                                                               \
            if (ahead > 0) {                                                                                                   \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                               \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
                __builtin_amdgcn_s_barrier();                                                                                  \
                if (w + 2 < nwin && !ABL_D) dma_b(win0 + w + 2, w & 1);                                                        \
            }                                                                                                                  \
            ++w;                                                                                                               \
        } else {                                                                                                               \
            ++trow;                                                                                                            \
            if (++swx == g.swx_n) {                                                                                            \
                swx = 0;                                                                                                       \
                trow += g.row_jump;                                                                                            \
            }                                                                                                                  \
        }                                                                                                                      \
        if (ahead > 0 && !ABL_B) read_b(bh, w & 1, trow, 0);                                                                   \
        lgkm_wait_older(ahead > 0); /* lo fragments are in */                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);                                                                          \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl[nb], acc[0][nb], 0, 0, 0); \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl[nb], acc[1][nb], 0, 0, 0); \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                               \
    }
    static_assert(DEPTH == 3, "the switch waits are vmcnt(4 * (DEPTH - 1)); the step loop is unrolled DEPTH times");
    for (int j = 0; j < nsteps; j += DEPTH) {
        CONV4_STEP(0, j)
        if (j + 1 < nsteps) CONV4_STEP(1, j + 1)
        if (j + 2 < nsteps) CONV4_STEP(2, j + 2)
    }
#undef CONV4_STEP
    __syncthreads();                               // the window buffers become the epilogue's staging patches

    // ---- epilogue: accumulators -> wave-private LDS patch [32 px][64 couts] -> 8 couts of one pixel per lane -------
    const int cblock = (mgrp * 2 + wm) * 64;
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
    if (ABL_E) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)                               // keep every accumulator live (no dead-code elimination of MFMAs)
#pragma unroll
                for (int i = 0; i < 16; ++i) ((volatile float*)smem)[tid] = acc[a][b][i];
        return;
    }
    // the row loop exists twice: once for plain STORE epilogues, whose body holds no load (so nothing in it ever waits for the previous
    // step's stores: conv_epilogue.h), once for everything else
    auto rows = [&](auto ld_tag) {
        constexpr bool LD = decltype(ld_tag)::value;
    #pragma unroll 1
        for (int nb = 0; nb < NB; ++nb) {              // (not unrolled: code size; the selects keep every accumulator index static)
    #pragma unroll
            for (int mb = 0; mb < 2; ++mb)
    #pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 a4;
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[mb][0][4 * gq + j];
    #pragma unroll
                        for (int k = 1; k < NB; ++k) x = (nb == k) ? acc[mb][k][4 * gq + j] : x;
                        a4[j] = x;
                    }
                    if (LD && e.out_vt != nullptr) {                           // pixel-major V^T straight from the accumulator layout
                        const int pid = wn * (32 * NB) + nb * 32 + r;
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
                const int pid = wn * (32 * NB) + nb * 32 + prow;
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

// patch shape / window geometry for a descriptor and a pixel-tile size; false when no patch shape fits
static bool plan4(const ppms_conv* d, int npx, Geo4& g) {
    const int mode = (d->kw > 1 && d->kh > 1) ? 2 : (d->kw > 1 ? 0 : 1);      // 0: x sweep, 1: y sweep, 2: 2-D sweep
    const int hx = mode != 1 ? d->kw - 1 : 0, hy = mode != 0 ? d->kh - 1 : 0;   // window halo (total) in x / y
    int bestC = -1, bestWr = 1 << 30;
    double bestw = 1e30;
    for (int C = 8; C <= npx; C *= 2) {
        const int R = npx / C;
        const int Wr = (R + hy) * (C + hx);
        if (Wr > 64 * MAXS4) continue;
        const double waste = (double)((d->W + C - 1) / C * C) * ((d->H + R - 1) / R * R) / ((double)d->W * d->H);
        if (waste < bestw - 1e-9 || (waste < bestw + 1e-9 && Wr < bestWr)) {
            bestw = waste;
            bestC = C;
            bestWr = Wr;
        }
    }
    if (bestC < 0) return false;
    g.C = bestC;
    g.R = npx / bestC;
    g.logC = 0;
    while ((1 << g.logC) < g.C) ++g.logC;
    g.tiles_x = (d->W + g.C - 1) / g.C;
    g.tiles_y = (d->H + g.R - 1) / g.R;
    g.WRL = g.C + hx;
    g.Wr = ((g.R + hy) * g.WRL + 15) / 16 * 16;       // whole 1 KiB DMA pieces (16 rows x 64 B); the extra rows are never read
    g.hxw = hx >> 1;
    g.hyw = hy >> 1;
    if (mode == 0) {
        g.swx_n = d->kw, g.row_jump = 0, g.nsweep = d->kw, g.rdy = 1;
    } else if (mode == 1) {
        g.swx_n = 1, g.row_jump = g.WRL - 1, g.nsweep = d->kh, g.rdy = 0;
    } else {
        g.swx_n = d->kw, g.row_jump = g.WRL - d->kw, g.nsweep = d->kh * d->kw, g.rdy = 0;
    }
    int nchunk = 0;
    for (int s = 0; s < d->nseg; ++s) nchunk += d->seg[s].c / 16;
    g.nchunk = nchunk;
    g.n0 = d->seg[0].c / 16;
    g.mgroups = d->M / 128;
    g.npieces = (g.Wr * 4 + NT - 1) / NT;
    return g.Wr <= 64 * MAXS4 && g.npieces <= MAXS4;
}

// 256-pixel tiles when that still gives >= ~1.5 workgroups per CU, else 128-pixel tiles
static int tile_px4(const ppms_conv* d) {
    const int64_t P = (int64_t)d->T * d->H * d->W;
    return (P / 256) * (d->M / 128) >= 384 ? 256 : 128;
}

}  // namespace

// returns 1 when the barrier-free kernel serves this convolution: M % 128 == 0, a spatial sweep of >= 3 taps (kw > 1 or kh > 1),
// 16-channel-aligned segments, a halo'd window that fits, and enough workgroups to fill the chip
extern "C" int ppms_conv_gemm4_applicable(const ppms_conv* d) {
    if (d == nullptr || d->M <= 0 || d->M % 128 != 0 || d->m_split % 64 != 0 || d->nseg < 1 || d->nseg > 2) return 0;
    if (d->kw == 1 && d->kh == 1) return 0;
    for (int s = 0; s < d->nseg; ++s)
        if (d->seg[s].c <= 0 || d->seg[s].c % 16 != 0) return 0;
    const int npx = tile_px4(d);
    const int64_t P = (int64_t)d->T * d->H * d->W;
    if ((P / npx) * (d->M / 128) < 256) return 0;            // fewer workgroups than CUs: conv_gemm2's K slicing fills the chip better
    Geo4 g;
    return plan4(d, npx, g) ? 1 : 0;
}

extern "C" int ppms_conv_gemm4(const ppms_conv* d, const ppms_conv* dev_desc, int tile_px, void* stream) {
    PPMS_REQUIRE(tile_px == 0 || tile_px == 128 || tile_px == 256, "conv_gemm4: tile_px must be 0 (choose), 128 or 256");
    PPMS_REQUIRE(d != nullptr && dev_desc != nullptr, "conv_gemm4: null descriptor");
    PPMS_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv_gemm4: nseg=%d", d->nseg);
    PPMS_REQUIRE(d->T > 0 && d->H > 0 && d->W > 0, "conv_gemm4: bad volume %dx%dx%d", d->T, d->H, d->W);
    PPMS_REQUIRE(d->M > 0 && d->M % 128 == 0 && d->m_split % 64 == 0, "conv_gemm4: M=%d must be a multiple of 128", d->M);
    PPMS_REQUIRE((d->kt & 1) && (d->kh & 1) && (d->kw & 1) && d->kw <= 15 && d->kh <= 15, "conv_gemm4: odd kernel extents <= 15");
    PPMS_REQUIRE(d->kw > 1 || d->kh > 1, "conv_gemm4: needs a spatial sweep axis (kw > 1 or kh > 1)");
    PPMS_REQUIRE(d->w != nullptr && d->bias != nullptr, "conv_gemm4: weights/bias missing");
    PPMS_REQUIRE(d->t_halo >= 0 && d->t_halo <= 8, "conv_gemm4: t_halo=%d", d->t_halo);
    PPMS_REQUIRE((int64_t)d->T * d->H * d->W < (1ll << 31) / 512, "conv_gemm4: volume too large for 32-bit pixel offsets");
    for (int s = 0; s < d->nseg; ++s) {
        PPMS_REQUIRE(d->seg[s].hi && d->seg[s].lo && d->seg[s].c > 0 && d->seg[s].c % 16 == 0 && d->seg[s].ld % 8 == 0,
                     "conv_gemm4: segment %d needs hi/lo planes, c %% 16 == 0 and ld %% 8 == 0", s);
        PPMS_REQUIRE(((uintptr_t)d->seg[s].hi & 15) == 0 && ((uintptr_t)d->seg[s].lo & 15) == 0, "conv_gemm4: segment %d not 16-B aligned", s);
    }
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && d->m_split >= d->M) break;
        PPMS_REQUIRE(e.n_valid > 0, "conv_gemm4: epilogue %d has n_valid=%d", hlf, e.n_valid);
        PPMS_REQUIRE(e.pre_f32 == nullptr || (e.n_valid % 4 == 0 && e.pre_f32_ld % 4 == 0), "conv_gemm4: pre_f32 needs n_valid and pre_f32_ld to be multiples of 4");
        {
            const char* why = epilogue_row8_check(e);
            PPMS_REQUIRE(why == nullptr, "conv_gemm4: epilogue %d: %s", hlf, why ? why : "");
        }
        if (e.out_sp.hi) PPMS_REQUIRE(e.out_sp.lo && e.out_sp.ld % 4 == 0, "conv_gemm4: epilogue %d SP output misaligned", hlf);
        if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU)
            PPMS_REQUIRE(e.aux_sp.hi && e.aux_sp.lo && e.aux_sp.ld % 4 == 0, "conv_gemm4: epilogue %d needs aux_sp", hlf);
        if (e.kind == PPMS_EPI_GRU) PPMS_REQUIRE(e.aux_f32 != nullptr, "conv_gemm4: GRU epilogue needs z");
    }
    const int npx = tile_px ? tile_px : tile_px4(d);
    Geo4 g;
    PPMS_REQUIRE(plan4(d, npx, g), "conv_gemm4: no patch shape fits the LDS window");
    PPMS_REQUIRE(g.nsweep >= 3, "conv_gemm4: the sweep must have at least 3 taps");
    PPMS_REQUIRE(4 * (DEPTH - 1) + g.npieces <= 18 && g.npieces >= 1, "conv_gemm4: window of %d rows needs too many DMA pieces", g.Wr);
    const int ntiles = g.tiles_x * g.tiles_y * d->T;
    size_t lds = (size_t)2 * g.npieces * NT * 16;     // two window buffers of whole 4 KiB DMA pieces
    if (lds < (size_t)4 * STG_WAVE) lds = (size_t)4 * STG_WAVE;          // the epilogue's transposition patches reuse the windows
    PPMS_REQUIRE(lds <= 80 * 1024, "conv_gemm4: window of %d rows does not fit", g.Wr);
    static ppms_device_once once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)conv4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    });
    if (npx == 256)
        hipLaunchKernelGGL(conv4_kernel<4>, dim3(ntiles * g.mgroups), dim3(NT), lds, (hipStream_t)stream, *d, g);
    else
        hipLaunchKernelGGL(conv4_kernel<2>, dim3(ntiles * g.mgroups), dim3(NT), lds, (hipStream_t)stream, *d, g);
    return ppms_check_launch("conv_gemm4");
}
