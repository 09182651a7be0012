// Implicit-GEMM convolution for large maps on v_mfma_f32_16x16x32_bf16: ONE wave per SIMD, 16-pixel block granularity (gfx950).
//
// Same math, descriptor and epilogues as conv_gemm2/3/5 (bf16x3 split: hi*hi + lo*hi + hi*lo, fp32 accumulate).  What changes against
// conv_gemm5 (8 waves, 32x32x16 MFMA, 64 x 128 wave tiles, 7 x 32-pixel tiles):
//   * the MFMA shape: under dense MFMA load on real operands the part holds a ~12 % higher clock on 16x16x32 than on 32x32x16 at equal
//     cycles per FLOP (tools/probe/mfma_shape_probe.hip, MI355X_MICROARCH.md "DVFS give-back" item 7);
//   * the tile: 16 rows x 13 columns = 208 pixels (13 blocks of 16: one block = one column of the patch) -> 51 200 pixels = 250 tiles on
//     256 CUs, 4 % of tile quantisation instead of 10.7 % (240 tiles of 224);
//   * the wave tile: a 4-wave workgroup, one wave per SIMD, each wave 64 (48) couts x all 13 pixel blocks of the tile (M = 256 / 192), or
//     64 couts x 7 / 6 blocks (M = 128): 12 MFMAs per pair of LDS fragment reads, 156 per 8 weight-fragment loads -- a third of conv_gemm5's
//     operand traffic per MFMA -- with the accumulators (208 registers) in the AGPR half of the 512-register file;
//   * 32-channel activation windows (one k32-step per tap): LDS rows of 128 B [hi k0-31 | lo k0-31], 16-B chunk c stored at position
//     c ^ (wy & 6) (wy = row inside the window column): the fragment reads are conflict-free for every tap offset.
// Window = halo'd patch stored COLUMN-major (row = wx * WH + wy), gathered by LDS-DMA with per-lane source addresses (zero page for padding),
// double buffered, one barrier per window; the taps sweep the window from LDS.  Without a spatial sweep (kh = kw = 1: the temporal GRU pass,
// 1x1 layers) nothing re-uses a window: the STREAM form of the kernel (template parameter) gives every k32-step its own window (the tile's 208
// pixels x 32 channels of one temporal tap: 7 LDS-DMA pieces per thread) in a ring of THREE buffers, issued two steps ahead.  (A first form with
// two 32-channel chunks per window in the double-buffered loop waited for its whole window at every switch and showed rare wrong pixels under
// back-to-back launches; the STREAM form is clean over hundreds of launches: tools/conv6_stress.py, tests/test_gpu_concurrency.py.)
// Weights: pack_conv6 (ppmstereo_amd/packing.py): [k32-step][M/16][hi, lo][lane = 16 kg + r][8] = the MFMA A-operand images.
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

constexpr int NT6 = 256;
constexpr int MAXP6 = 14;                     // LDS-DMA pieces (16 B) per thread and window
// Pieces per thread and window as a COMPILE-TIME constant of the kernel variant (CR = rows of a window column): the loop's vmcnt waits count
// them, and s_waitcnt takes an immediate (a run-time count through a switch cost a 25-branch decision tree per k-step).  16-row columns
// (x sweeps up to 15 taps: 27 x 16 rows; two 13-column chunks without spatial taps): 14; 18- / 20-row columns (3x3, (1,5,1), (1,3,1)): 9.
// A window with fewer rows issues the surplus pieces as reads of the zero page into the buffer's padding.
__host__ __device__ constexpr int conv6_np(int cr) { return cr == 16 ? 14 : 9; }
constexpr int NBT6 = 13;                      // 16-pixel blocks (columns) per tile
constexpr int STG6_ROWS = 7 * 16;             // pixels a wave stages per epilogue pass


struct Geo6 {
    int tiles_x, tiles_y;
    int WH, WC;              // window column height (16 + y halo; even) and columns (13 + x halo)
    int hxl, hyl;            // halo on the left / above
    int mode;                // 0: x sweep, 1: y sweep, 2: 2-D sweep, 4: no spatial taps (the STREAM form: one k32-step per window)
    int nsweep;              // k32-steps per window
    int swx_n, inc, jump;    // the LDS row offset of the tap advances by inc per step and by jump more after every swx_n steps
    int yinc, ywrap;         // the tap's y offset (the swizzle phase of the fragment reads): += yinc per step, += 1 and back to ... see kernel
    int nchunk, n0;          // windows per temporal tap (all segments), windows of segment 0
    int cpw;                 // input channels per window: 32
    int lz0;                 // windows whose index inside the tap is >= lz0 hold bf16-exact activations (all-zero lo plane): hi x lo products skipped
    int npieces;             // DMA pieces per thread and window that carry rows (<= conv6_np(WH), which is what every window issues)
    int wbytes;              // bytes of one window buffer (conv6_np(WH) * 4 KiB)
    int ks;                  // 1: the K-split form of M = 128 (below); nchunk / lz0 then count ONE K group's windows per tap, *_full all of them
    int nchunk_full, lz0_full;
    int64_t P;
#ifdef PPMS_CONV6_TIMING
    long long* dbg;
#endif
};

#ifdef PPMS_CONV6_TIMING
static long long* g_conv6_dbg = nullptr;
// debug build only (tools/conv6_phase_probe.py): wave 0 of every workgroup leaves [workgroup][K] = wall clock (100 MHz) and [workgroup][8 + K] =
// shader cycle counter at: 0 entry, 1 loop start, 2 loop end, 3 exit (the clock the part holds in the loop = cycles / wall time)
#define CONV6_STAMP(K)                                                                    \
    if (g.dbg != nullptr && (__builtin_amdgcn_readfirstlane(threadIdx.x) & 255) == 0) {   \
        g.dbg[(int64_t)blockIdx.x * 16 + (K)] = wall_clock64();                           \
        g.dbg[(int64_t)blockIdx.x * 16 + 8 + (K)] = (long long)__builtin_amdgcn_s_memtime(); \
    }
#else
#define CONV6_STAMP(K)
#endif

// (M0: compiler-reserved on this target, written in the SAME statement that reads it; an "m0" clobber only draws "clobber list contains reserved
// registers".)
// LDS-DMA of 16 B per lane through a buffer resource: lane l's bytes land at LDS address m0 + 16 l; the source is srd.base + off (32-bit, per lane), and
// a lane whose offset lies beyond srd.num_records reads ZEROS (tools/probe/buf_lds_oob_probe.hip) -- the padding of a window costs no zero page and no
// per-lane select.  Inline asm on purpose: for the compiler's own LDS-DMA (__builtin_amdgcn_global_load_lds) the waitcnt pass puts a wait for that
// transfer in front of EVERY later inline-asm statement (each has a memory clobber, so each "may read LDS") -- one exposed memory round trip per
// piece.  The loop orders these transfers itself (counted vmcnt + barrier at the window switch).
__device__ __forceinline__ void dma16_6(unsigned off, const u32x4& srd, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(srd), "s"(lds_dst) : "memory");
}
constexpr unsigned CONV6_NUM_RECORDS = 0xFFFFFF00u;     // range of a window's buffer resource; CONV6_OOB is beyond it
constexpr unsigned CONV6_OOB = 0xFFFFFFF0u;

#ifndef CONV6_EPI_G
#define CONV6_EPI_G 2
#endif
#ifndef CONV6_ABL_NOWAIT_A
#define CONV6_ABL_NOWAIT_A 0
#endif
#ifndef CONV6_XCD_ORDER
#define CONV6_XCD_ORDER 0      // measured neutral (+-1 % on every 1/4-scale conv, the (3,3,3) flow-head conv -3 %): off; -DCONV6_XCD_ORDER=1 for A/B
#endif
#ifndef CONV6_SLACK
#define CONV6_SLACK 0
#endif
#if CONV6_SLACK == 1
#define CONV6_SWITCH_SLACK asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#define CONV6_PRE_BARRIER
#elif CONV6_SLACK == 2
#define CONV6_SWITCH_SLACK asm volatile("s_sleep 20" ::: "memory");
#define CONV6_PRE_BARRIER
#elif CONV6_SLACK == 3
#define CONV6_SWITCH_SLACK
#define CONV6_PRE_BARRIER asm volatile("s_sleep 20" ::: "memory");
#else
#define CONV6_SWITCH_SLACK
#define CONV6_PRE_BARRIER
#endif
#ifndef CONV6_KSPLIT
#define CONV6_KSPLIT 1         // -DCONV6_KSPLIT=0: M = 128 convolutions always in the two-pixel-halves layout (A/B builds)
#endif
#include "conv6_asm.h"

// MB: 16-cout blocks per wave -- 4 (M = 256, 128) or 3 (M = 192); CR: rows of a window column = 16 + y halo (16, 18, 20)
// STREAM: convolutions WITHOUT spatial taps ((kt,1,1) and 1x1): a window = the tile's 208 pixels x 32 channels of one temporal tap, used by ONE k32-step
// (nothing re-uses it), three window buffers, every step issues the window two steps ahead -- see CONV6_STEP1
// GRP: a GROUPED convolution of two groups (ppms_conv.groups == 2; M = 256, x sweep of <= 5 taps): input segment s feeds the couts of epilogue half s only
// (the two 128 -> 128 (1,1,5) tails of convz1 / convr1, ppmtereo_update.py:254-312, as ONE launch in the M = 256 wave layout instead of two M = 128
// launches with half the work per fixed cost).  Waves 0-1 own group 0's couts, waves 2-3 group 1's; every window exists twice -- the same patch of
// segment 0 and of segment 1, 9 DMA pieces each -- and a wave reads its group's copy.
// KS: the K-SPLIT form of an M = 128 convolution (round 6).  In the plain M = 128 layout a wave owns 64 couts x 7 (6) pixel blocks: 84 MFMAs per k32-step, a
// step of 0.7 us -- shorter than the L2 round trip of the next step's weight fragments, with a step's fixed costs (waits, address upkeep, window switch)
// spread over half the work of an M = 256 step.  Here the two wave PAIRS split K instead of the pixels: waves 0-1 take one half of the windows of every tap,
// waves 2-3 the other half (each half: the same share of windows with and without a lo plane, so the two-phase loop stays in lockstep), every wave 64 couts x
// all 13 pixel blocks -- the M = 256 step body on half as many steps.  Both halves' windows are resident like GRP's two images.  Behind the loop the pairs
// exchange the partial sums of the pixel blocks the other one finishes (through LDS, one barrier) and the epilogue runs in the two-pixel-halves layout.
// The sum of a cout is (first half of K) + (second half of K): a fixed order, but not the plain layout's -- results agree to fp32 rounding, not bit for bit.
template <int MB, int CR, bool STREAM = false, bool GRP = false, bool KS = false>
__global__ __launch_bounds__(NT6, 1) void conv6_kernel(const ppms_conv pv, const Geo6 g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ppms_conv& p = pv;                       // by value in the kernel arguments (see conv_gemm2.hip)
    CONV6_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lkg = lane >> 4;                    // fragment roles: pixel (cout) of the block, k-group of 8 channels
    // wave roles: M = 256 / 192: wave w owns cout blocks [w MB, w MB + MB) and all 13 pixel blocks; M = 128: (wm, wn) = (w & 1, w >> 1), 64 couts x
    // pixel blocks [0, 7) or [7, 13)
    constexpr bool DUAL = GRP || KS;               // two window images per window buffer, one per wave pair
    const bool split = p.M == 128 && !KS;
    const int wm = (split || KS) ? (wave & 1) : wave;
    const int wn = split ? (wave >> 1) : 0;
    const int kg = KS ? (wave >> 1) : 0;           // KS: this wave's K group
    const int nbw = split ? (wn ? 6 : 7) : NBT6;
    const int blk0 = wn ? 7 : 0;
    // KS: K group kg's c-th window of a tap is window gchunk(kg, c) of the tap in the descriptor's order (phase 0: the windows with a lo plane, phase 1: the others)
    auto gchunk = [&](int grp, int c) { return c < g.lz0 ? grp * g.lz0 + c : g.lz0_full + grp * (g.nchunk - g.lz0) + (c - g.lz0); };
    auto wchunk = [&](int c) { return KS ? gchunk(kg, c) : c; };          // window index the WEIGHTS of this wave's step are stored under
    const int ncf = KS ? g.nchunk_full : g.nchunk;
    // workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one; each XCD has its own L2): XCD k gets the k-th eighth of the tile
    // list, so that the x / y neighbours whose windows overlap in their halo columns fetch them through ONE L2 (placement is a speed hint only)
    int tile = blockIdx.x;
#if CONV6_XCD_ORDER
    {
        const int nt = (int)gridDim.x, per = (nt + 7) >> 3, full = nt - 8 * (per - 1);      // XCDs 0 .. full-1 hold `per` tiles, the others per - 1
        const int xcd = tile & 7, slot = tile >> 3;
        tile = xcd < full ? xcd * per + slot : full * per + (xcd - full) * (per - 1) + slot;
    }
#endif
    const int tx = tile % g.tiles_x;
    tile /= g.tiles_x;
    const int ty = tile % g.tiles_y;
    const int tf = tile / g.tiles_y;
    const int x0 = tx * NBT6, y0 = ty * 16;
    const int H = p.H, W = p.W, T = p.T;
    const int HW = H * W;
    const int ht = p.kt >> 1;

    // prefetch for window w + 1 goes out in the first step of window w (the other buffer was released by the barrier that ended window w - 1)
    // ---- window slots: LDS piece q = tid + i * 256 (lane-linear destination); row = q >> 3, position q & 7 --------------------------------
    // per piece and input segment ONE register: the byte offset of the lane's 16 B from the window's base (= the segment's hi plane at the window's
    // frame shift and first channel): pixel * bytes per pixel + 16-B unit inside the window's channels (+ the distance of the lo plane); padding:
    // an offset beyond the buffer resource's range, which reads zeros.  (The host checks that every offset fits: conv6_offsets_fit.)
    constexpr int NPG = DUAL ? 9 : conv6_np(CR);       // pieces per thread of ONE window image
    constexpr int NP = DUAL ? 2 * NPG : NPG;           // pieces per thread and window (DUAL: group 0's image, then group 1's)
    constexpr unsigned GBYTES = NPG * NT6 * 16;        // DUAL: distance of group 1's window image
    const int ld0 = p.seg[0].ld * 2, ld1 = p.seg[p.nseg - 1].ld * 2;          // bytes between pixels
    const char* const sp0h = (const char*)p.seg[0].hi;
    const char* const sp1h = (const char*)p.seg[p.nseg - 1].hi;
    const unsigned pd0 = (unsigned)((const char*)p.seg[0].lo - sp0h), pd1 = (unsigned)((const char*)p.seg[p.nseg - 1].lo - sp1h);
    unsigned off0[NPG], off1[NPG], off[NP];          // off: the offsets of the segment the NEXT window lies in (copied by dma_setup; a piece's register is
                                                    // not written again before the next window's set-up, a whole k-step after the piece went out)
    {
        const int rows = g.WH * g.WC;
#pragma unroll
        for (int i = 0; i < NPG; ++i) {
            const int q = tid + i * NT6;
            const int row = q >> 3, pos = q & 7;
            off0[i] = off1[i] = CONV6_OOB;
            if (row < rows) {
                const int wx = row / g.WH, wy = row - wx * g.WH;
                const int x = x0 + wx - g.hxl, y = y0 + wy - g.hyl;
                const int c = pos ^ (wy & 6);                    // the 16-B chunk of the row this position holds: plane * 4 + k-group
                if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
                    const unsigned pix = (unsigned)((tf * H + y) * W + x), unit16 = (unsigned)((c & 3) * 16);
                    off0[i] = pix * (unsigned)ld0 + unit16 + ((c >> 2) ? pd0 : 0u);
                    off1[i] = pix * (unsigned)ld1 + unit16 + ((c >> 2) ? pd1 : 0u);
                }
            }
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    const unsigned wave_dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wave * 1024));
    // one DMA piece of window (temporal tap kz, chunk) into buffer `buf`: the window's buffer resource (base = the segment's hi plane + frame shift +
    // first channel; stride 0; range CONV6_NUM_RECORDS) + the piece's offset
    int d_buf = 0;
    u32x4 d_srd = {0u, 0u, CONV6_NUM_RECORDS, 0x00020000u};
    u32x4 d_srd1 = {0u, 0u, CONV6_NUM_RECORDS, 0x00020000u};      // GRP: group 1's window (segment 1)
    auto dma_setup = [&](int kz, int chunk, int buf) {
        const int dt = kz - ht;
        if constexpr (GRP) {                       // both groups' windows of (tap kz, chunk): the same channels of segment 0 and of segment 1
            const int c0 = chunk * g.cpw * 2;
            const uint64_t b0 = (uint64_t)(uintptr_t)(sp0h + (int64_t)dt * HW * ld0 + c0), b1 = (uint64_t)(uintptr_t)(sp1h + (int64_t)dt * HW * ld1 + c0);
            unsigned w0 = __builtin_amdgcn_readfirstlane((unsigned)b0), w1 = __builtin_amdgcn_readfirstlane((unsigned)(b0 >> 32) & 0xffffu);
            unsigned w2 = __builtin_amdgcn_readfirstlane((unsigned)b1), w3 = __builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32) & 0xffffu);
            asm volatile("s_nop 4" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3));      // (five wait states between a VALU-written word and its VMEM reader: below)
            d_srd[0] = w0, d_srd[1] = w1, d_srd1[0] = w2, d_srd1[1] = w3;
            d_buf = buf;
#pragma unroll
            for (int i = 0; i < NPG; ++i) off[i] = off0[i], off[NPG + i] = off1[i];
            return;
        }
        if constexpr (KS) {                        // the two K groups' windows of (tap kz, local window `chunk`)
            const int gc0 = gchunk(0, chunk), gc1 = gchunk(1, chunk);
            const int s0 = gc0 >= g.n0 ? 1 : 0, s1 = gc1 >= g.n0 ? 1 : 0;
            const uint64_t b0 = (uint64_t)(uintptr_t)((s0 ? sp1h : sp0h) + (int64_t)dt * HW * (s0 ? ld1 : ld0) + (gc0 - (s0 ? g.n0 : 0)) * g.cpw * 2);
            const uint64_t b1 = (uint64_t)(uintptr_t)((s1 ? sp1h : sp0h) + (int64_t)dt * HW * (s1 ? ld1 : ld0) + (gc1 - (s1 ? g.n0 : 0)) * g.cpw * 2);
            unsigned w0 = __builtin_amdgcn_readfirstlane((unsigned)b0), w1 = __builtin_amdgcn_readfirstlane((unsigned)(b0 >> 32) & 0xffffu);
            unsigned w2 = __builtin_amdgcn_readfirstlane((unsigned)b1), w3 = __builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32) & 0xffffu);
            asm volatile("s_nop 4" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3));
            d_srd[0] = w0, d_srd[1] = w1, d_srd1[0] = w2, d_srd1[1] = w3;
            d_buf = buf;
#pragma unroll
            for (int i = 0; i < NPG; ++i) off[i] = s0 ? off1[i] : off0[i], off[NPG + i] = s1 ? off1[i] : off0[i];
            return;
        }
        const int sg = (chunk >= g.n0) ? 1 : 0;
        const int c0 = (chunk - (sg ? g.n0 : 0)) * g.cpw * 2;                  // byte offset of the window's first channel
        const int64_t shift = (int64_t)dt * HW * (sg ? ld1 : ld0) + c0;
        const uint64_t base = (uint64_t)(uintptr_t)((sg ? sp1h : sp0h) + shift);
        // The resource words must be FIVE wait states old when a buffer_load reads them if a VALU instruction (v_readfirstlane) wrote them -- the
        // documented "VALU writes SGPR -> VMEM reads that SGPR" hazard.  The compiler inserts those wait states for its own memory instructions only: the
        // LDS-DMA pieces are inline asm, and tools/probe/lds_dma_hazard_probe.hip shows what the violation does on this part (0 wait states: the piece
        // reads through a garbage descriptor -- a memory fault, or silently the PREVIOUS window's base: whole stale pieces, the symptom round 5 fenced by
        // timing) and what is safe at any distance (M0, the offset VGPR and SALU-written resource words rewritten directly behind a piece: 3 x 10^8
        // wave-pieces at 3.5 TB/s, none wrong; profiles/r06_lds_dma_hazard_probe.txt).  Today the compiler computes these words on the SALU (the window
        // index is uniform) -- this statement makes the distance a property of the source instead of the register allocator's mood: the words pass through
        // an asm statement that holds 5 wait states, so whatever produced them is at least that far from every piece.
        unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)base), w_hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
        asm volatile("s_nop 4" : "+s"(w_lo), "+s"(w_hi));
        d_srd[0] = w_lo;
        d_srd[1] = w_hi;
        d_buf = buf;
#pragma unroll
        for (int i = 0; i < NP; ++i) off[i] = sg ? off1[i] : off0[i];
    };
    // (round 5 spaced the pieces >= 4 MFMAs apart on the belief that an LDS-DMA reads M0 / its offset register late; the round-6 probe refutes that --
    //  what bit was the VALU-written resource, see dma_setup -- but the spacing stays: a piece costs ~60 issue cycles, one per MFMA group hides it)
    auto dma_piece = [&](int i) { dma16_6(off[i], (DUAL && i >= NPG) ? d_srd1 : d_srd, wave_dst + (unsigned)(d_buf * g.wbytes + i * (NT6 * 16))); };
    // the pieces of a window leave in the first TWO k-steps of the window before it: PPS0 in the first, PPS1 in the second
    constexpr int PPS0 = (NP + 1) / 2, PPS1 = NP / 2;

    // ---- weights: [k32-step][M/16][plane][64 lanes][16 B]; this wave's 2 MB fragments are contiguous ----------------------------------------
    const char* abase = (const char*)p.w;
    const unsigned avoff0 = (unsigned)(wm * MB * 2048 + lane * 16), avoff1 = avoff0 + 4096;
    const int64_t astep = (int64_t)(p.M >> 4) * 2048;
    u32x4 areg[2][8];
    auto load_a = [&](u32x4 (&st)[8], int ks) {
        const char* sb = abase + (int64_t)ks * astep;
#define CONV6_LA(K, OFFV, IMM) if ((K) < 2 * MB) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #IMM : "+v"(st[K]) : "v"(OFFV), "s"(sb) : "memory");
        CONV6_LA(0, avoff0, 0) CONV6_LA(1, avoff0, 1024) CONV6_LA(2, avoff0, 2048) CONV6_LA(3, avoff0, 3072)
        CONV6_LA(4, avoff1, 0) CONV6_LA(5, avoff1, 1024) CONV6_LA(6, avoff1, 2048) CONV6_LA(7, avoff1, 3072)
#undef CONV6_LA
    };

    // ---- B-operand addressing: row = (blk0 + n) * CR + tap offset + li; chunk position (plane * 4 + lkg) ^ ((li + tap y) & 6); the n * CR * 128
    // part is an immediate of the read, hi and lo fragments differ in bit 6 ---------------------------------------------------------------
    const unsigned grp_off = DUAL ? (unsigned)(wave >> 1) * GBYTES : 0u;         // DUAL: this wave pair's window image
    auto lane_addr = [&](int tyo) { return lds0 + grp_off + (unsigned)(blk0 * CR * 128 + li * 128) + (unsigned)(((lkg ^ ((li + tyo) & 6)) & 7) << 4); };

    // temporal taps outside the readable frames contribute zeros: skip them (contiguous kz range)
    const int kz0 = (ht - tf - p.t_halo) > 0 ? (ht - tf - p.t_halo) : 0;
    const int kz1 = (ht + T + p.t_halo - 1 - tf) < (p.kt - 1) ? (ht + T + p.t_halo - 1 - tf) : (p.kt - 1);
    // Window order.  A window = (temporal tap kz, chunk of cpw input channels).  Windows whose chunk is >= lz0 hold bf16-exact activations (an
    // all-zero lo plane): their hi x lo products add exact zeros and are left out.  A branch per block cost more than it saved (2 340 scalar
    // branches per wave of the z/r conv) and an if / else around two step bodies makes the accumulators phi nodes inside the loop (391 spilled
    // registers), so the K loop runs in TWO PHASES, each with its own straight-line step body: phase 0 = the windows with chunk < lz0 of every
    // tap (full product), phase 1 = the others (no hi x lo MFMAs).  The summation order is a fixed function of the descriptor.
    const int nkz = kz1 + 1 - kz0;
    const int nw0 = nkz * g.lz0, nw1 = nkz * (g.nchunk - g.lz0);
    const int nwin = nw0 + nw1;

    f32x4 acc[4][13];
    u32x4 ring[4][2];
    // current window (ckz, cch) and its successor in the order above (nkz_ < 0: none)
    int ckz = kz0, cch = nw0 > 0 ? 0 : g.lz0;
    auto next_window = [&](int kz, int ch, int& okz, int& och) {
        const bool ph1 = ch >= g.lz0;
        okz = kz, och = ch + 1;
        if (och == (ph1 ? g.nchunk : g.lz0)) {
            och = ph1 ? g.lz0 : 0;
            if (++okz > kz1) {
                if (ph1 || nw1 == 0) okz = -1;
                else okz = kz0, och = g.lz0;
            }
        }
    };

#define CONV6_STEP(U, NBW, SKIP, LASTSTEP)                                                                                         \
    {                                                                                                                              \
        /* tap state of the NEXT step */                                                                                           \
        int n_sw = sw + 1, n_swx = swx + 1, n_off = off + g.inc, n_ty = tyo + g.yinc, n_w = w;                                      \
        if (n_swx == g.swx_n) n_swx = 0, n_off += g.jump, n_ty += g.ywrap;                                                          \
        const bool wend = n_sw == g.nsweep;                                                                                        \
        const bool last = (LASTSTEP) && wend && nxkz < 0; /* no step behind this one */                                            \
        if (wend) n_sw = 0, n_swx = 0, n_off = 0, n_ty = 0, n_w = w ^ 1;                                                            \
        if (last) n_off = off, n_ty = tyo, n_w = w;                                                                                \
        const unsigned bh = lane_addr(tyo) + (unsigned)(w * g.wbytes + off * 128), bhn = lane_addr(n_ty) + (unsigned)(n_w * g.wbytes + n_off * 128); \
        const int sw_cur = sw;                                                                                                     \
        const bool issue = sw < 2 && nxkz >= 0; /* the first two steps of a window carry the DMA of the next window (other buffer) */ \
        if (issue && sw == 0) dma_setup(nxkz, nxch, w ^ 1);                                                                        \
        const int ksn = last ? (ckz * ncf + wchunk(cch)) * g.nsweep + sw : (wend ? (nxkz * ncf + wchunk(nxch)) * g.nsweep : (ckz * ncf + wchunk(cch)) * g.nsweep + sw + 1); \
        const char* sbn = abase + (int64_t)ksn * astep;                                                                            \
        conv6_step<MB, NBW, CR, SKIP>(acc, areg[U], areg[(U) ^ 1], ring, bh, bh ^ 64u, bhn, bhn ^ 64u, avoff0, avoff1, sbn, [&](int K) { \
            constexpr int SL = conv6_shape<MB, NBW, SKIP>::SLOTS;                                                                    \
            static_assert(SL >= PPS0, "a step has a DMA slot for each of its pieces");                                             \
            if (issue) { /* at most one piece per slot, the pieces spread evenly over the step's slots */                          \
                if (sw_cur == 0) {                                                                                                 \
                    _Pragma("unroll") for (int j = 0; j < PPS0; ++j) if (j * SL / PPS0 == K) dma_piece(j);                          \
                } else {                                                                                                           \
                    _Pragma("unroll") for (int j = 0; j < PPS1; ++j) if (j * SL / PPS1 == K) dma_piece(PPS0 + j);                   \
                }                                                                                                                  \
            }                                                                                                                      \
        });                                                                                                                        \
        sw = n_sw, swx = n_swx, off = n_off, tyo = n_ty;                                                                            \
        if (wend && !last) {                                                                                                       \
            /* window switch: everything this wave issued has landed (the next window's pieces are older than this step's weight  */ \
            /* loads), every wave is done reading the old window                                                                  */ \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                            \
            CONV6_PRE_BARRIER                                                                                                      \
            __builtin_amdgcn_s_barrier();                                                                                          \
            CONV6_SWITCH_SLACK                                                                                                     \
            w = n_w;                                                                                                               \
            ckz = nxkz, cch = nxch;                                                                                                \
            next_window(ckz, cch, nxkz, nxch);                                                                                     \
            conv6_prime<NBW, CR>(ring, bhn, bhn ^ 64u);                                                                             \
        } else if (CONV6_ABL_NOWAIT_A) { /* timing experiment (wrong results): no wait for the next step's weight fragments */      \
        } else if (issue) { /* the next step's weight fragments: everything but the pieces this step issued behind them */           \
            if constexpr (NP == 18) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); /* GRP: PPS0 = PPS1 = 9 */                     \
            else if constexpr (NP == 14) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); /* PPS0 = PPS1 = 7 */                     \
            else if (sw_cur == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); /* NP = 9: PPS0 = 5 */                           \
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); /* PPS1 = 4 */                                                    \
        } else {                                                                                                                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                       \
        }                                                                                                                          \
    }
    // one phase: nst steps (phase 0 of a two-phase loop: an even number -- the host keeps lz0 * nsweep even --, so that phase 1 starts on weight
    // register stage 0 again)
#define CONV6_PHASE(NBW, SKIP, NST)                                                                                                \
    {                                                                                                                              \
        const int nst = (NST);                                                                                                     \
        int j = 0;                                                                                                                 \
        for (; j + 1 < nst; j += 2) {                                                                                              \
            CONV6_STEP(0, NBW, SKIP, false)                                                                                        \
            CONV6_STEP(1, NBW, SKIP, j + 2 >= nst)                                                                                 \
        }                                                                                                                          \
        if (j < nst) CONV6_STEP(0, NBW, SKIP, true)                                                                                \
    }
#define CONV6_LOOP(NBW)                                                                                                            \
    {                                                                                                                              \
        _Pragma("unroll") for (int a = 0; a < MB; ++a) _Pragma("unroll") for (int b = 0; b < (NBW); ++b) {                          \
            acc[a][b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};                                                                            \
            asm volatile("" : "+a"(acc[a][b]));                                                                                     \
        }                                                                                                                          \
        int nxkz, nxch;                                                                                                            \
        next_window(ckz, cch, nxkz, nxch);                                                                                         \
        dma_setup(ckz, cch, 0);                                                                                                    \
        asm volatile("s_nop 7" ::: "memory"); /* (the resource's SGPRs come from v_readfirstlane: wait states before a VMEM read) */ \
        _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                                            \
            dma_piece(i);                                                                                                          \
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); /* 64 cycles between two pieces */         \
        }                                                                                                                          \
        load_a(areg[0], (ckz * ncf + wchunk(cch)) * g.nsweep);                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                              \
        int sw = 0, swx = 0, off = 0, tyo = 0, w = 0;                                                                               \
        conv6_prime<NBW, CR>(ring, lane_addr(0), lane_addr(0) ^ 64u);                                                               \
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); /* the zeroed accumulators are AGPR writes: wait states in front of the first MFMA */ \
        CONV6_PHASE(NBW, false, nw0 * g.nsweep)                                                                                    \
        CONV6_PHASE(NBW, true, nw1 * g.nsweep)                                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");                                        \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(areg[0][k]), "v"(areg[1][k]));                          \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(ring[k][0]), "v"(ring[k][1]));                          \
    }
    // ---- STREAM form: one k32-step per window.  Step s multiplies window s (buffer w), while the LDS-DMA of window s + 1 (issued during step s - 1) is
    // landing and the pieces of window s + 2 go out (NPS pieces, one per hook slot, behind the step's weight loads).  At the end of the step:
    // vmcnt(NPS) = everything but those pieces has landed (the older window s + 1, the next step's weight fragments), barrier (every wave is done
    // reading buffer w: it is window s + 3's target), the fragment ring is primed from window s + 1.  Same window order as above (two phases).
    constexpr int NPS = 7;                            // pieces per thread and window: 208 rows x 8 positions / 256 threads = 6.5
#define CONV6_STEP1(U, NBW, SKIP, LASTSTEP)                                                                                        \
    {                                                                                                                              \
        const bool last = (LASTSTEP) && nxkz < 0; /* no step behind this one */                                                    \
        const bool issue = n2kz >= 0;             /* window s + 2 exists */                                                        \
        const int bn = (w == 2) ? 0 : w + 1, b2 = (bn == 2) ? 0 : bn + 1;                                                           \
        const unsigned bh = lane_addr(0) + (unsigned)(w * g.wbytes), bhn = lane_addr(0) + (unsigned)((last ? w : bn) * g.wbytes);    \
        if (issue) dma_setup(n2kz, n2ch, b2);                                                                                      \
        const int ksn = last ? (ckz * g.nchunk + cch) : (nxkz * g.nchunk + nxch);                                                   \
        const char* sbn = abase + (int64_t)ksn * astep;                                                                            \
        conv6_step<MB, NBW, CR, SKIP>(acc, areg[U], areg[(U) ^ 1], ring, bh, bh ^ 64u, bhn, bhn ^ 64u, avoff0, avoff1, sbn, [&](int K) { \
            constexpr int SL = conv6_shape<MB, NBW, SKIP>::SLOTS;                                                                    \
            static_assert(SL >= NPS, "a step has a DMA slot for each piece of a window");                                          \
            if (issue) {                                                                                                           \
                _Pragma("unroll") for (int j = 0; j < NPS; ++j) if (j * SL / NPS == K) dma_piece(j);                                \
            }                                                                                                                      \
        });                                                                                                                        \
        if (!last) {                                                                                                               \
            if (issue) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); /* NPS */                                       \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                       \
            __builtin_amdgcn_s_barrier();                                                                                          \
            w = bn;                                                                                                                \
            ckz = nxkz, cch = nxch, nxkz = n2kz, nxch = n2ch;                                                                       \
            if (nxkz >= 0) next_window(nxkz, nxch, n2kz, n2ch);                                                                     \
            conv6_prime<NBW, CR>(ring, bhn, bhn ^ 64u);                                                                             \
        }                                                                                                                          \
    }
#define CONV6_PHASE1(NBW, SKIP, NST)                                                                                               \
    {                                                                                                                              \
        const int nst = (NST);                                                                                                     \
        int j = 0;                                                                                                                 \
        for (; j + 1 < nst; j += 2) {                                                                                              \
            CONV6_STEP1(0, NBW, SKIP, false)                                                                                       \
            CONV6_STEP1(1, NBW, SKIP, j + 2 >= nst)                                                                                \
        }                                                                                                                          \
        if (j < nst) CONV6_STEP1(0, NBW, SKIP, true)                                                                               \
    }
#define CONV6_LOOP1(NBW)                                                                                                           \
    {                                                                                                                              \
        _Pragma("unroll") for (int a = 0; a < MB; ++a) _Pragma("unroll") for (int b = 0; b < (NBW); ++b) {                          \
            acc[a][b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};                                                                            \
            asm volatile("" : "+a"(acc[a][b]));                                                                                     \
        }                                                                                                                          \
        int nxkz, nxch, n2kz = -1, n2ch = 0;                                                                                       \
        next_window(ckz, cch, nxkz, nxch);                                                                                         \
        if (nxkz >= 0) next_window(nxkz, nxch, n2kz, n2ch);                                                                         \
        _Pragma("unroll 1") for (int pw = 0; pw < 2; ++pw) { /* windows 0 and 1 into buffers 0 and 1 */                             \
            if (pw == 1 && nxkz < 0) break;                                                                                        \
            dma_setup(pw ? nxkz : ckz, pw ? nxch : cch, pw);                                                                       \
            asm volatile("s_nop 7" ::: "memory");                                                                                  \
            _Pragma("unroll") for (int i = 0; i < NPS; ++i) {                                                                       \
                dma_piece(i);                                                                                                      \
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); /* 64 cycles between two pieces */     \
            }                                                                                                                      \
        }                                                                                                                          \
        load_a(areg[0], ckz * g.nchunk + cch);                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                              \
        int w = 0;                                                                                                                 \
        conv6_prime<NBW, CR>(ring, lane_addr(0), lane_addr(0) ^ 64u);                                                               \
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                                                                           \
        CONV6_PHASE1(NBW, false, nw0)                                                                                              \
        CONV6_PHASE1(NBW, true, nw1)                                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");                                        \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(areg[0][k]), "v"(areg[1][k]));                          \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(ring[k][0]), "v"(ring[k][1]));                          \
    }
    CONV6_STAMP(1)
    if constexpr (STREAM) {
        if (nbw == NBT6) CONV6_LOOP1(13)
        else if constexpr (MB == 4) {
            if (nbw == 7) CONV6_LOOP1(7) else CONV6_LOOP1(6)
        }
    } else {
        if (nbw == NBT6) CONV6_LOOP(13)
        else if constexpr (MB == 4 && !DUAL) {     // (M = 128: the two pixel halves of the tile)
            if (nbw == 7) CONV6_LOOP(7) else CONV6_LOOP(6)
        }
    }
    CONV6_STAMP(2)
#undef CONV6_LOOP
#undef CONV6_PHASE
#undef CONV6_STEP
#undef CONV6_LOOP1
#undef CONV6_PHASE1
#undef CONV6_STEP1
    __syncthreads();                               // the window buffers become the epilogue's staging areas
    int e_nbw = nbw, e_blk0 = blk0;                // the epilogue's layout: pixel blocks [e_blk0, e_blk0 + e_nbw) of the tile are this wave's to finish
    if constexpr (KS) {
        // K groups -> pixel halves.  Wave (wm, kg) holds the partial sums of ALL 13 pixel blocks over its half of K; it finishes blocks 0..6 (kg = 0) or
        // 7..12 (kg = 1) and hands the other blocks' partial sums to its partner (wm, kg ^ 1): 16 B per lane, [block][m][lane], conflict-free
        constexpr int XW = 7 * 4 * 64 * 4;                              // floats per wave: 7 blocks x 4 cout blocks x 64 lanes x 4 (28 KiB)
        float* xs = (float*)smem + wave * XW;
        const float* xr = (const float*)smem + (wave ^ 2) * XW;
        if (kg == 0) {
#pragma unroll
            for (int nl = 0; nl < 6; ++nl)
#pragma unroll
                for (int m = 0; m < 4; ++m) *(f32x4*)(xs + ((nl * 4 + m) * 64 + lane) * 4) = acc[m][7 + nl];
        } else {
#pragma unroll
            for (int nl = 0; nl < 7; ++nl)
#pragma unroll
                for (int m = 0; m < 4; ++m) *(f32x4*)(xs + ((nl * 4 + m) * 64 + lane) * 4) = acc[m][nl];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int nl = 0; nl < 7; ++nl)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][nl] += *(const f32x4*)(xr + ((nl * 4 + m) * 64 + lane) * 4);
        } else {
#pragma unroll
            for (int nl = 0; nl < 6; ++nl)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][7 + nl] += *(const f32x4*)(xr + ((nl * 4 + m) * 64 + lane) * 4);
        }
        __syncthreads();                           // (the staging patches below overlay the exchange area)
        e_nbw = kg ? 6 : 7;
        e_blk0 = kg ? 7 : 0;
    }

    // ---- epilogue: accumulators (lane: pixel li, couts 16 m + 4 lkg ..) -> wave-private LDS patch [pixel][16 MB couts] -> 8 couts of one
    // pixel per lane, the shared row epilogue (conv_epilogue.h).  Two passes of <= 7 pixel blocks.
    constexpr int LD6 = 16 * MB + 4;
    float* stg = (float*)smem + wave * (STG6_ROWS * LD6);
    const int c_lo = wm * 16 * MB;                                     // first cout of this wave
    const int q = lane & 7;                                            // cout group of 8 inside the wave's couts (MB = 3: groups 0..5)
    const bool qok = q < 2 * MB;
    const int cout = c_lo + q * 8;
    const int lane_half = (cout >= p.m_split) ? 1 : 0;
    float b8[8];
    {
        const float* bp = p.bias + (qok ? cout : c_lo);
        const f32x4 b0 = gld<f32x4>(bp), b1 = gld<f32x4>(bp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b8[j] = b0[j];
            b8[4 + j] = b1[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(b8[j]));       // landed before the row loops (vmcnt counts loads and stores in one order)
    }
    // AB: accumulator block of the wave's first block (0; KS, K group 1: 7 -- its accumulators are indexed by the tile's blocks, its blocks to finish are 7..12)
    auto do_pass = [&](auto pass_tag, auto ab_tag) {
        constexpr int pass = decltype(pass_tag)::value, AB = decltype(ab_tag)::value;
        constexpr int pb0 = pass * 7;                                   // first block (wave-local) of the pass
        const int npb = (e_nbw - pb0) < 7 ? (e_nbw - pb0) : 7;          // blocks in this pass (<= 0: none)
        if (npb > 0) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nl = 0; nl < 7; ++nl) {
            if (pb0 + nl < 13 && nl < npb) {
#pragma unroll
                for (int m = 0; m < MB; ++m) *(f32x4*)(stg + (nl * 16 + li) * LD6 + m * 16 + 4 * lkg) = acc[m][AB + pb0 + nl < 13 ? AB + pb0 + nl : 0];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int nit = npb * 2;                                        // 8 pixels per iteration
        for (int hlf = 0; hlf < 2; ++hlf) {
            // the wave's couts that fall into this half of the descriptor (M = 192: wave 2 straddles m_split = 128)
            const int h_lo = hlf ? p.m_split : 0, h_hi = hlf ? p.M : p.m_split;
            if (c_lo + 16 * MB <= h_lo || c_lo >= h_hi) continue;
            const ppms_epilogue e = p.epi[hlf];                         // BY VALUE (SGPRs)
            const int cl = cout - h_lo;
            const bool mine = qok && lane_half == hlf;
            auto rows = [&](auto cls_tag, auto grp_tag) {
                constexpr int CLS = decltype(cls_tag)::value, G = decltype(grp_tag)::value;
#pragma unroll 1
                for (int it0 = 0; it0 < nit; it0 += G) {
                    row8_aux aux[G];
                    int64_t pixg[G];
                    bool okg[G];
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        const int prow = (it0 + gi) * 8 + (lane >> 3);
                        const int px = x0 + e_blk0 + pb0 + (prow >> 4), py = y0 + (prow & 15);
                        okg[gi] = mine && it0 + gi < nit && px < W && py < H;
                        pixg[gi] = (int64_t)(tf * H + py) * W + px;
                        if (okg[gi]) row8_fetch<CLS>(e, pixg[gi], cl, aux[gi]);
                    }
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        float v[8];
                        const int prow = (it0 + gi) * 8 + (lane >> 3);
                        const float* sp = stg + (prow < STG6_ROWS ? prow : 0) * LD6 + (qok ? q : 0) * 8;
                        const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = a[j] + b8[j];
                            v[4 + j] = b[j] + b8[4 + j];
                        }
                        if (okg[gi]) row8_finish<CLS>(e, v, pixg[gi], cl, HW, aux[gi]);
                    }
                }
            };
            using I0 = std::integral_constant<int, EPI_CLS_PLAIN>;
            using I1 = std::integral_constant<int, EPI_CLS_PRE>;
            using I2 = std::integral_constant<int, EPI_CLS_AUX>;
            using I3 = std::integral_constant<int, EPI_CLS_GRU>;
            using I4 = std::integral_constant<int, EPI_CLS_ANY>;
            using I5 = std::integral_constant<int, EPI_CLS_AUXPRE>;
            using G1 = std::integral_constant<int, 1>;
            using G2 = std::integral_constant<int, 2>;
            using GL = std::integral_constant<int, CONV6_EPI_G>;
            const int cls = epilogue_class(e);
            // (groups of CONV6_EPI_G 8-pixel steps: the operands of a group are fetched before its rows are finished -- one wave per SIMD has nobody to
            //  hide a memory round trip behind, so the epilogue's time is its number of such round trips)
            if (cls == EPI_CLS_PLAIN) rows(I0{}, G2{});
            else if (cls == EPI_CLS_PRE) rows(I1{}, GL{});
            else if (cls == EPI_CLS_AUX) rows(I2{}, GL{});
            else if (cls == EPI_CLS_GRU) rows(I3{}, GL{});
            else if (cls == EPI_CLS_AUXPRE) rows(I5{}, GL{});
            else rows(I4{}, G1{});
        }
        }
    };
    using Z0 = std::integral_constant<int, 0>;
    if constexpr (KS) {
        if (kg == 0) do_pass(Z0{}, Z0{});
        else do_pass(Z0{}, std::integral_constant<int, 7>{});
    } else {
        do_pass(Z0{}, Z0{});
        do_pass(std::integral_constant<int, 1>{}, Z0{});
    }
    CONV6_STAMP(3)
}

// window geometry for a descriptor; false when this kernel does not serve it
static bool plan6(const ppms_conv* d, Geo6& g) {
    const bool grouped = d->groups == 2;
    if (d->groups > 2 || d->groups < 0) return false;
    // two groups: segment s -> the couts of epilogue half s; served in the M = 256 layout for an x sweep of <= 5 taps (a 16 x 17 window: 9 pieces per group)
    if (grouped && !(d->nseg == 2 && d->seg[0].c == d->seg[1].c && d->M == 256 && d->m_split == 128 && d->kh == 1 && d->kw >= 3 && d->kw <= 5 && d->lo_zero_from == 0))
        return false;
    const bool stream = d->kw == 1 && d->kh == 1;         // no spatial taps: the STREAM form (one k32-step per window, three buffers)
    g.mode = stream ? 4 : (d->kw > 1 && d->kh > 1) ? 2 : (d->kw > 1 ? 0 : 1);
    const int hx = (g.mode == 0 || g.mode == 2) ? d->kw - 1 : 0, hy = (g.mode == 1 || g.mode == 2) ? d->kh - 1 : 0;
    g.cpw = 32;
    int nchunk = 0;
    for (int s = 0; s < d->nseg; ++s) {
        if (d->seg[s].c <= 0 || d->seg[s].c % g.cpw) return false;
        nchunk += d->seg[s].c / g.cpw;
    }
    if (grouped) nchunk = d->seg[0].c / g.cpw;           // the groups advance in lockstep: a window = one chunk of EACH segment
    g.nchunk = nchunk;
    g.n0 = grouped ? nchunk : d->seg[0].c / g.cpw;
    g.lz0 = (d->lo_zero_from > 0 && d->lo_zero_from % g.cpw == 0) ? d->lo_zero_from / g.cpw : nchunk;
    g.tiles_x = (d->W + NBT6 - 1) / NBT6;
    g.tiles_y = (d->H + 15) / 16;
    g.WH = 16 + hy;
    g.WC = NBT6 + hx;
    g.hxl = hx >> 1;
    g.hyl = hy >> 1;
    g.yinc = 0, g.ywrap = 0;
    if (g.mode == 4) {
        g.nsweep = 1, g.swx_n = 1 << 30, g.inc = 0, g.jump = 0;
    } else if (g.mode == 0) {
        g.nsweep = d->kw, g.swx_n = 1 << 30, g.inc = g.WH, g.jump = 0;
    } else if (g.mode == 1) {
        g.nsweep = d->kh, g.swx_n = 1 << 30, g.inc = 1, g.jump = 0, g.yinc = 1;
    } else {
        g.nsweep = d->kh * d->kw, g.swx_n = d->kw, g.inc = g.WH, g.jump = 1 - d->kw * g.WH, g.ywrap = 1;
    }
    const int rows = g.WH * g.WC;
    if (g.lz0 < nchunk && ((g.lz0 * g.nsweep) & 1)) g.lz0 = nchunk;   // (phase 0 must hold an even number of steps: else every product is computed)
    g.npieces = (rows * 8 + NT6 - 1) / NT6;
    g.wbytes = (stream ? 7 : conv6_np(g.WH)) * NT6 * 16;
    g.P = (int64_t)d->T * d->H * d->W;
    g.ks = 0, g.nchunk_full = nchunk, g.lz0_full = g.lz0;
    // K-split form of M = 128 (the kernel's KS parameter): an even split of the windows with and of those without a lo plane over the two wave pairs, a
    // window image of <= 9 pieces (two images per buffer: 144 KiB), and -- as for the plain layout -- an even number of phase-0 steps per K group
    if (CONV6_KSPLIT && d->M == 128 && !grouped && !stream && g.npieces <= 9 && (g.WH == 16 || g.WH == 18 || g.WH == 20) && g.nsweep >= 2 && nchunk % 2 == 0 &&
        nchunk >= 4) {
        int lz = g.lz0;
        if (lz < nchunk && ((lz % 2) || ((nchunk - lz) % 2) || (((lz / 2) * g.nsweep) & 1))) lz = nchunk;      // uneven phases: every product is computed
        g.ks = 1;
        g.nchunk_full = nchunk, g.lz0_full = lz;
        g.nchunk = nchunk / 2, g.lz0 = lz / 2;
        g.wbytes = 2 * 9 * NT6 * 16;
        return true;
    }
    if (grouped) {
        g.wbytes = 2 * 9 * NT6 * 16;
        return g.WH == 16 && g.npieces <= 9 && g.nsweep >= 2;
    }
    if (stream) return g.npieces == 7;
    return (g.WH == 16 || g.WH == 18 || g.WH == 20) && g.npieces <= conv6_np(g.WH) && g.nsweep >= 2;
}

static bool conv6_volume_fits(const ppms_conv* d) { return (int64_t)d->T * d->H * d->W < (1ll << 22); }

// a window's LDS-DMA addresses a lane's bytes as (the segment's hi plane + frame shift) + a 32-bit offset that also spans the distance to the lo plane:
// both planes of a segment must lie inside one 4 GiB range above the hi plane (ppmstereo_amd's SPTensor keeps them in one allocation)
static bool conv6_offsets_fit(const ppms_conv* d) {
    const int64_t P = (int64_t)d->T * d->H * d->W;
    for (int s = 0; s < d->nseg; ++s) {
        const int64_t pd = (const char*)d->seg[s].lo - (const char*)d->seg[s].hi;
        if (pd <= 0 || pd + P * d->seg[s].ld * 2 + 4096 >= (int64_t)CONV6_NUM_RECORDS) return false;
    }
    return true;
}

static size_t conv6_lds(const ppms_conv* d, const Geo6& g) {
    size_t lds = (size_t)(g.mode == 4 ? 3 : 2) * g.wbytes;
    const size_t stg = (size_t)4 * STG6_ROWS * ((d->M == 192 ? 48 : 64) + 4) * 4;
    return lds < stg ? stg : lds;
}

}  // namespace

// 0: not served.  1: served, and the 208-pixel tiles fill the chip at least as well as conv_gemm5's 224- / 256-pixel tiles would (>= 85 % of the
// CU-slots of the launch's rounds carry pixels).  2: served, but the fill is poor (the caller keeps conv_gemm5 there).
extern "C" int ppms_conv_gemm6_applicable(const ppms_conv* d) {
    if (d == nullptr || (d->M != 256 && d->M != 192 && d->M != 128) || d->nseg < 1 || d->nseg > 2) return 0;
    if (d->m_split % 8 != 0 || !conv6_volume_fits(d)) return 0;
    for (int s = 0; s < d->nseg; ++s)
        if (d->seg[s].hi == nullptr || d->seg[s].lo == nullptr) return 0;
    if (!conv6_offsets_fit(d)) return 0;
    if (!(d->kt & 1) || !(d->kh & 1) || !(d->kw & 1)) return 0;
    if (d->kw > 1 && d->kh > 1 && d->kh > 5) return 0;
    if (d->kw > 15 || d->kh > 5) return 0;
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && d->m_split >= d->M) break;
        if (d->epi[h].out_vt != nullptr) return 0;                       // V^T is written from the accumulator layout: conv_gemm5 / gemm1
    }
    Geo6 g;
    if (!plan6(d, g)) return 0;
    if (conv6_lds(d, g) > 160 * 1024) return 0;
    const int64_t tiles = (int64_t)g.tiles_x * g.tiles_y * d->T;
    const int64_t cus = ppms_num_cus();
    if (tiles < cus * 25 / 32) return 0;                                 // fewer workgroups than CUs: the K-sliced small-map kernels fill the chip better
    const double fill = (double)g.P / ((double)((tiles + cus - 1) / cus) * cus * 16 * NBT6);
    return fill >= 0.85 ? 1 : 2;
}

#ifdef PPMS_CONV6_TIMING
extern "C" void ppms_debug_conv6_timing(long long* p) { g_conv6_dbg = p; }
#endif

extern "C" int ppms_conv_gemm6(const ppms_conv* d, const ppms_conv* dev_desc, void* stream) {
    PPMS_REQUIRE(d != nullptr && dev_desc != nullptr, "conv_gemm6: null descriptor");
    PPMS_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv_gemm6: nseg=%d", d->nseg);
    PPMS_REQUIRE(d->T > 0 && d->H > 0 && d->W > 0, "conv_gemm6: bad volume %dx%dx%d", d->T, d->H, d->W);
    PPMS_REQUIRE((d->M == 256 || d->M == 192 || d->M == 128) && d->m_split % 8 == 0, "conv_gemm6: M=%d must be 128, 192 or 256 (m_split a multiple of 8)", d->M);
    PPMS_REQUIRE((d->kt & 1) && (d->kh & 1) && (d->kw & 1) && d->kw <= 15 && d->kh <= 5, "conv_gemm6: odd kernel extents, kw <= 15, kh <= 5");
    PPMS_REQUIRE(d->w != nullptr && d->bias != nullptr, "conv_gemm6: weights/bias missing");
    PPMS_REQUIRE(d->t_halo >= 0 && d->t_halo <= 8, "conv_gemm6: t_halo=%d", d->t_halo);
    PPMS_REQUIRE(conv6_volume_fits(d), "conv_gemm6: volume too large for the packed window slots (< 2^22 pixels)");
    for (int s = 0; s < d->nseg; ++s) {
        PPMS_REQUIRE(d->seg[s].hi && d->seg[s].lo && d->seg[s].c > 0 && d->seg[s].ld % 8 == 0 && d->seg[s].ld <= 1024,
                     "conv_gemm6: segment %d needs hi/lo planes, ld %% 8 == 0 and ld <= 1024", s);
        PPMS_REQUIRE(((uintptr_t)d->seg[s].hi & 15) == 0 && ((uintptr_t)d->seg[s].lo & 15) == 0, "conv_gemm6: segment %d not 16-B aligned", s);
    }
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && d->m_split >= d->M) break;
        PPMS_REQUIRE(e.n_valid > 0, "conv_gemm6: epilogue %d has n_valid=%d", hlf, e.n_valid);
        PPMS_REQUIRE(e.out_vt == nullptr, "conv_gemm6: no out_vt epilogue (use ppms_conv_gemm5 / ppms_gemm1)");
        PPMS_REQUIRE(e.pre_f32 == nullptr || (e.n_valid % 4 == 0 && e.pre_f32_ld % 4 == 0), "conv_gemm6: pre_f32 needs n_valid and pre_f32_ld to be multiples of 4");
        {
            const char* why = epilogue_row8_check(e);
            PPMS_REQUIRE(why == nullptr, "conv_gemm6: epilogue %d: %s", hlf, why ? why : "");
        }
        if (e.out_sp.hi) PPMS_REQUIRE(e.out_sp.lo && e.out_sp.ld % 4 == 0, "conv_gemm6: epilogue %d SP output misaligned", hlf);
        if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU)
            PPMS_REQUIRE(e.aux_sp.hi && e.aux_sp.lo && e.aux_sp.ld % 4 == 0, "conv_gemm6: epilogue %d needs aux_sp", hlf);
        if (e.kind == PPMS_EPI_GRU) PPMS_REQUIRE(e.aux_f32 != nullptr, "conv_gemm6: GRU epilogue needs z");
    }
    PPMS_REQUIRE(conv6_offsets_fit(d), "conv_gemm6: the lo plane of a segment must follow its hi plane inside one 4 GiB range (32-bit window offsets)");
    Geo6 g;
    PPMS_REQUIRE(plan6(d, g), "conv_gemm6: not a convolution this kernel serves (segments in multiples of 32 channels, a halo'd 16 x 13 window of <= 14 DMA "
                              "pieces per thread)");
    const size_t lds = conv6_lds(d, g);
    PPMS_REQUIRE(lds <= 160 * 1024, "conv_gemm6: LDS budget exceeded (%zu B)", lds);
    const int ntiles = g.tiles_x * g.tiles_y * d->T;
    PPMS_REQUIRE(g.WH == 16 || g.WH == 18 || g.WH == 20, "conv_gemm6: window columns of %d rows", g.WH);
    static ppms_device_once once;
    once.run([] {
#define CONV6_ATTR(MBV, CRV) (void)hipFuncSetAttribute((const void*)conv6_kernel<MBV, CRV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        CONV6_ATTR(4, 16) CONV6_ATTR(4, 18) CONV6_ATTR(4, 20) CONV6_ATTR(3, 16) CONV6_ATTR(3, 18) CONV6_ATTR(3, 20)
#undef CONV6_ATTR
        (void)hipFuncSetAttribute((const void*)conv6_kernel<4, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv6_kernel<3, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv6_kernel<4, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv6_kernel<4, 16, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv6_kernel<4, 18, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv6_kernel<4, 20, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
#ifdef PPMS_CONV6_TIMING
    g.dbg = g_conv6_dbg;
#endif
#define CONV6_GO(MBV, CRV) hipLaunchKernelGGL((conv6_kernel<MBV, CRV>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g)
    if (d->groups == 2) {
        hipLaunchKernelGGL((conv6_kernel<4, 16, false, true>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g);
    } else if (g.ks) {
        if (g.WH == 16) hipLaunchKernelGGL((conv6_kernel<4, 16, false, false, true>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g);
        else if (g.WH == 18) hipLaunchKernelGGL((conv6_kernel<4, 18, false, false, true>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g);
        else hipLaunchKernelGGL((conv6_kernel<4, 20, false, false, true>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g);
    } else if (g.mode == 4) {
        if (d->M == 192) hipLaunchKernelGGL((conv6_kernel<3, 16, true>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g);
        else hipLaunchKernelGGL((conv6_kernel<4, 16, true>), dim3(ntiles), dim3(NT6), lds, (hipStream_t)stream, *d, g);
    } else if (d->M == 192) {
        if (g.WH == 16) CONV6_GO(3, 16); else if (g.WH == 18) CONV6_GO(3, 18); else CONV6_GO(3, 20);
    } else {
        if (g.WH == 16) CONV6_GO(4, 16); else if (g.WH == 18) CONV6_GO(4, 18); else CONV6_GO(4, 20);
    }
#undef CONV6_GO
    return ppms_check_launch("conv_gemm6");
}
