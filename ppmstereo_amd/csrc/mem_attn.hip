// Pick-and-Play memory read-out: for every clip i, hid_i = bf16( softmax(Q_i K'_i^T * scale) V_i ) over the
// pixels of the picked frames, fused with mfg_i = mf_i + beta * hid_i.
// Replaces the T sequential flash_attn_func calls of /root/reference/models/core/ppmstereo.py:517-552
// (1 head, d = 128, Nq = n, Nk = ksel * n; bf16 operands, fp32 softmax / accumulate, bf16 result).
//
// gfx950 structure: workgroup = 4 waves x 32 queries; KV tile = 64 keys; swapped QK^T (S^T = K Q^T, so a query's
// scores sit in one lane pair and softmax needs one cross-lane op); the S^T accumulator tile is re-used in place as
// the B operand of O^T += V^T P^T (no LDS round trip for P); V arrives already transposed ([d][key], written by the
// to_v conv epilogue) so both MFMA operands are plain 8/16-byte LDS reads.  K rows are XOR-swizzled on 16-B chunks,
// V^T rows on 8-B granules: all fragment reads are bank-conflict free.  Two LDS stages, register-staged prefetch.
#include "common.h"
#include <type_traits>

namespace {

constexpr int D = 128, KT = 64, QW = 32, NW = 4;
constexpr int K_TILE = KT * D * 2;       // 16 KiB
constexpr int V_TILE = D * KT * 2;       // 16 KiB
constexpr int ATT_STAGE = K_TILE + V_TILE;

__device__ __forceinline__ u32x4 load16_guard(const bf16_t* row, int e0, int limit, bool vec_ok) {
    // 8 bf16 at row[e0 .. e0+8), elements >= limit read as 0
    if (vec_ok && e0 + 8 <= limit) return gload16(row + e0);
    u32x4 v = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned int bits = 0;
        if (e0 + j < limit) bits = *(const PPMS_GLOBAL unsigned short*)(uintptr_t)(row + e0 + j);
        v[j >> 1] |= bits << (16 * (j & 1));
    }
    return v;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// TAIL = (n % 64 != 0): only that instantiation carries the per-key masking of a frame's last tile.
// P16: the unnormalised probabilities P~ enter O^T += V^T P~ as fp16 (11 significand bits) instead of bf16 (8), and V^T holds the fp16 image of
// the bf16-rounded values (ppms_epilogue.vt_f16: the same numbers -- the reference's cast of V to bf16, ppmstereo.py:550, is preserved -- in the
// operand format of v_mfma_f32_*_f16).  Under the running maximum P~ <= 1, so fp16's range is no constraint here; probabilities below 2^-14 of the
// row maximum keep an ABSOLUTE precision of 2^-25 (fp16 subnormals), which is 2^-25 of the denominator at most.
template <bool TAIL, bool P16>
__global__ __launch_bounds__(256, 2) void mem_attn_kernel(const bf16_t* __restrict__ qb, const bf16_t* __restrict__ kb,
                                                          const bf16_t* __restrict__ vt, const int32_t* __restrict__ sel, int ksel,
                                                          float scale_log2, const float* __restrict__ beta_p, ppms_sp mf, ppms_sp mfg,
                                                          bf16_t* __restrict__ out_bf16, int n, float* __restrict__ part_o,
                                                          float* __restrict__ part_ml, int32_t* __restrict__ redo, int redo_stride, int sps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int clip = blockIdx.y;
    if (redo != nullptr) {                       // fix-up pass behind mem_attn64_kernel: only flagged tiles are recomputed
        int32_t* f = redo + (int64_t)(clip * gridDim.z + blockIdx.z) * redo_stride + blockIdx.x;
        const int flagged = *f;
        if (flagged == 0) return;                // (uniform)
        __syncthreads();
    }
    const int q0 = blockIdx.x * (QW * NW) + wave * QW;
    const int qi = q0 + r;                       // this lane's query
    const int qc = qi < n ? qi : n - 1;          // clamped for loads
    const bool n_vec = (n & 7) == 0;

    // ---- Q fragments: B operand of S^T = K Q^T, lane (r,h) holds Q[q][16 s + 8 h + j] -----------------
    bf16x8 qf[8];
    {
        const bf16_t* qp = qb + ((int64_t)clip * n + qc) * D + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
    }

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f32x16){0};
    float m_run = -INFINITY, l_run = 0.0f;

    const int tpf = (n + KT - 1) / KT;           // tiles per frame
    // split mode (part_o != null): gridDim.z = ceil(ksel / sps) splits, this workgroup reads the `sps` picked frames of split blockIdx.z
    // and writes unnormalised partials; a combine kernel merges them (more, smaller work units, and the small scales get ksel x the
    // parallelism)
    const int nsplit = gridDim.z;
    const int split = (nsplit > 1) ? (int)blockIdx.z : 0;
    const int it0 = split * sps * tpf;
    const int ntile = (nsplit > 1 && it0 + sps * tpf < ksel * tpf) ? it0 + sps * tpf : ksel * tpf;

    u32x4 rk[4], rv[4];
    auto load_k = [&](int it) {
        const int slot = it / tpf;
        const int key0 = (it - slot * tpf) * KT;
        // K': [clip][slot][key][128], 256-B rows; thread -> rows (tid>>4) + 16 i, chunk tid&15
        const bf16_t* kbase = kb + ((int64_t)(clip * ksel + slot) * n) * D;
        if (!TAIL || key0 + KT <= n) {           // whole tile inside the frame (always, when n % 64 == 0): no per-row guards
#pragma unroll
            for (int i = 0; i < 4; ++i) rk[i] = gload16(kbase + (int64_t)(key0 + (tid >> 4) + 16 * i) * D + (tid & 15) * 8);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + (tid >> 4) + 16 * i;
                rk[i] = key < n ? gload16(kbase + (int64_t)key * D + (tid & 15) * 8) : (u32x4){0, 0, 0, 0};
            }
        }
    };
    auto load_v = [&](int it) {
        const int slot = it / tpf;
        const int key0 = (it - slot * tpf) * KT;
        const int frame = sel[clip * 5 + slot];
        // V^T: [frame][d][n], this tile = 64 keys (128 B) of every d row; thread -> d = (tid>>3) + 32 i, chunk tid&7
        const bf16_t* vbase = vt + (int64_t)frame * D * n;
        if (!TAIL || (n_vec && key0 + KT <= n)) {   // 16-byte aligned rows and a whole tile: plain vector loads
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = gload16(vbase + (int64_t)((tid >> 3) + 32 * i) * n + key0 + (tid & 7) * 8);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = (tid >> 3) + 32 * i;
                rv[i] = load16_guard(vbase + (int64_t)d * n, key0 + (tid & 7) * 8, n, n_vec);
            }
        }
    };
    auto store_k = [&](int stage) {
        char* ks = smem + stage * ATT_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            *(u32x4*)(ks + row * 256 + (((tid & 15) ^ (row & 15)) << 4)) = rk[i];
        }
    };
    auto store_v = [&](int stage) {
        char* vs = smem + stage * ATT_STAGE + K_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // V^T row d holds 64 keys as 4 groups of 16; inside a group the keys are stored [0-3, 8-11 | 4-7, 12-15] so
            // that lane half h of the PV MFMA reads its 8 keys (4h..4h+3, 8+4h..8+4h+3) as ONE 16-byte chunk.
            // This thread holds keys 8e..8e+7 of group G: the low 8 bytes go to chunk 2G, the high 8 bytes to chunk 2G+1,
            // both at byte 8e.  Chunks are XOR-swizzled by (d>>1)&7: conflict-free for ds_read_b128.
            const int d = (tid >> 3) + 32 * i;
            const int G = (tid & 7) >> 1, e = tid & 1;
            const int f = (d >> 1) & 7;
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const u32x4 v = rv[i];
            *(u32x2*)(vs + d * 128 + (((2 * G) ^ f) << 4) + 8 * e) = (u32x2){v[0], v[1]};
            *(u32x2*)(vs + d * 128 + (((2 * G + 1) ^ f) << 4) + 8 * e) = (u32x2){v[2], v[3]};
        }
    };
    // k-steps [s0, s1) of S^T[key][query] = K Q^T for one staged K tile
    auto s_steps = [&](const char* ks, f32x16 (&st)[2], int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk) {
                const int row = kblk * 32 + r;
                const bf16x8 kf = *(const bf16x8*)(ks + row * 256 + (((2 * s + h) ^ (row & 15)) << 4));
                st[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kblk], 0, 0, 0);
            }
        }
    };
    auto mask_tail = [&](f32x16 (&st)[2], int key0) {       // keys beyond the frame's last pixel (only when n % 64 != 0)
        if (TAIL && key0 + KT > n) {
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int key = key0 + kblk * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (key >= n) st[kblk][g] = -INFINITY;
                }
        }
    };

    // Software pipeline over the KV tiles.  The softmax VALU stream of a tile (~1400 issue cycles with its 32
    // quarter-rate v_exp) is longer than the tile's 32 MFMAs (1024 cycles), and two resident workgroups tend to fall into
    // lock step, so MFMA and VALU phases did not overlap (MFMA pipe 41 % busy).  Here one wave carries TWO score tiles:
    // while the VALU works through softmax(S_j), the matrix pipe already computes S_{j+1} = K_{j+1} Q^T; K therefore
    // runs one tile ahead of V in the two LDS stages.  Iteration j:
    //     global loads K_{j+2}, V_{j+1} -> registers
    //     S_{j+1} MFMAs  ||  max / exp / sum of S_j           (independent instruction streams, one basic block each side
    //     O += V_j P_j                                          of the rare rescale branch)
    //     registers -> LDS: K_{j+2} over K_j, V_{j+1} over V_{j-1};  one barrier
    auto body = [&](int j, int nt, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
        const int it = it0 + j;
        const bool more1 = j + 1 < nt, more2 = j + 2 < nt;
        if (more2) load_k(it + 2);
        if (more1) load_v(it + 1);
        const char* kn = smem + ((j + 1) & 1) * ATT_STAGE;              // K_{j+1} (stale data on the last tile: result unused)
        const char* vs = smem + (j & 1) * ATT_STAGE + K_TILE;           // V_j
        nxt[0] = (f32x16){0};
        nxt[1] = (f32x16){0};
        s_steps(kn, nxt, 0, 2);
        // ---- online softmax (fp32), one query per lane pair (r, r+32) --------------------------------
        float mx = cur[0][0];
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, cur[kblk][g]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * scale_log2);
        // rescale O and l only when some query of the wave raised its running max (wave-uniform branch); NaN scores
        // (T == 1) must still poison the output, hence the unordered compare
        if (__any(!(m_new <= m_run))) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] *= alpha;
            m_run = m_new;
        }
        s_steps(kn, nxt, 2, 8);
        const float neg_m = -m_run;
        float psum = 0.0f;
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                // raw v_exp_f32: arguments are <= 0, results in [0,1]; values below 2^-126 flush to 0 (irrelevant for P)
                const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[kblk][g], scale_log2, neg_m));
                cur[kblk][g] = pe;
                psum += pe;
            }
        l_run += psum;
        mask_tail(nxt, ((it + 1) % tpf) * KT);
        // ---- O^T[d][query] += V^T P^T ; P^T fragments come straight from the S^T accumulators -------------
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int chunk = (kblk * 2 + s2) * 2 + h;
                if constexpr (P16) {
                    f16x8 pf;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) pf[jj] = (_Float16)cur[kblk][8 * s2 + jj];
#pragma unroll
                    for (int dblk = 0; dblk < 4; ++dblk) {
                        const int d = dblk * 32 + r;
                        const f16x8 vf = *(const f16x8*)(vs + d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4));
                        o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dblk], 0, 0, 0);
                    }
                } else {
                    bf16x8 pf;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16_t)cur[kblk][8 * s2 + jj];
#pragma unroll
                    for (int dblk = 0; dblk < 4; ++dblk) {
                        const int d = dblk * 32 + r;
                        const bf16x8 vf = *(const bf16x8*)(vs + d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4));
                        o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dblk], 0, 0, 0);
                    }
                }
            }
        }
        if (more2) store_k(j & 1);
        if (more1) store_v((j + 1) & 1);
        __syncthreads();
    };

    const int nt = ntile - it0;
    f32x16 sa[2], sb[2];
    load_k(it0);
    load_v(it0);
    store_k(0);
    store_v(0);
    if (nt > 1) load_k(it0 + 1);
    __syncthreads();
    sa[0] = (f32x16){0};
    sa[1] = (f32x16){0};
    s_steps(smem, sa, 0, 8);
    mask_tail(sa, (it0 % tpf) * KT);
    if (nt > 1) store_k(1);
    __syncthreads();
    for (int j = 0; j < nt; j += 2) {
        body(j, nt, sa, sb);
        if (j + 1 < nt) body(j + 1, nt, sb, sa);
    }

    // ---- epilogue: hid = bf16(O / l); mfg = mf + beta * hid ---------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (qi >= n) return;
    if (part_o != nullptr) {
        const int64_t row = ((int64_t)clip * nsplit + split) * n + qi;
        float* po = part_o + row * D;
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v4 = {o[dblk][4 * g], o[dblk][4 * g + 1], o[dblk][4 * g + 2], o[dblk][4 * g + 3]};
                *(f32x4*)(po + dblk * 32 + 8 * g + 4 * h) = v4;
            }
        if (h == 0) {
            part_ml[row * 2] = m_run;
            part_ml[row * 2 + 1] = l_tot;
        }
        return;
    }
    const float inv_l = 1.0f / l_tot;
    const float beta = beta_p[0];
    const int64_t pix = (int64_t)clip * n + qi;
    const bf16_t* mh = (const bf16_t*)mf.hi + pix * mf.ld;
    const bf16_t* ml = (const bf16_t*)mf.lo + pix * mf.ld;
    bf16_t* gh = (bf16_t*)mfg.hi + pix * mfg.ld;
    bf16_t* gl = (bf16_t*)mfg.lo + pix * mfg.ld;
#pragma unroll
    for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = dblk * 32 + 8 * g + 4 * h;
            bf16x4 hid;
#pragma unroll
            for (int j = 0; j < 4; ++j) hid[j] = (bf16_t)(o[dblk][4 * g + j] * inv_l);
            if (out_bf16) *(bf16x4*)(out_bf16 + pix * D + d) = hid;
            bf16x4 oh = hid, ol = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};      // mf.hi == NULL: the view receives hid itself
            if (mf.hi != nullptr) {
                const bf16x4 a = *(const bf16x4*)(mh + d), b = *(const bf16x4*)(ml + d);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y = join_bf16(a[j], b[j]) + beta * (float)hid[j];
                    bf16_t hh, ll;
                    split_bf16(y, hh, ll);
                    oh[j] = hh;
                    ol[j] = ll;
                }
            }
            *(bf16x4*)(gh + d) = oh;
            *(bf16x4*)(gl + d) = ol;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// 64 queries per wave, LDS-DMA ring, 16x16x32 MFMA (n % 64 == 0, split workspace given).  The hot loop is attn64_asm.h, generated by
// tools/gen_attn_asm.py (schedule and register roles: its docstring).
//
// What bounds the 32-query kernel above is not the matrix pipe: (1) it issues one ds_read_b128 per MFMA (each K / V^T
// fragment feeds a single 32-query block); (2) its KV tile for iteration j+1 is requested at the start of iteration j and
// must have arrived by its end -- a ~2 us round trip through L2 / Infinity Cache under load, against ~1 us of MFMA work;
// (3) softmax VALU work and MFMAs of two co-resident waves do not overlap on this part (a wave streaming MFMAs starves
// the other wave's VALU; independent VALU work of the SAME wave does issue under its MFMAs: tools/probe/coissue_probe.hip).
// This kernel addresses all three:
//  * a wave owns FOUR 16-query blocks: every K / V^T fragment read feeds four MFMAs.  State = O^T (128 x 64: 8 x 4 blocks of 16 x 16,
//    128 accumulator registers) + two S^T sub-tiles (2 x 32) + Q (64): the unified 512-register file, one wave per SIMD;
//  * the MFMA shape is 16x16x32, not 32x32x16: at equal cycles per FLOP the part holds a ~12 % higher clock on it under dense MFMA
//    load on real operands (tools/probe/mfma_shape_probe.hip: 2065 vs 1854 TFLOP/s from registers, 1720 vs 1576 from LDS; equal on zeros),
//    and the 32 keys of a sub-tile are ONE k-step of O^T += V^T P, so the V^T fragments are read once per sub-tile.  K row m of block
//    row b is key 8 (m >> 2) + 4 b + (m & 3) of the sub-tile: lane (query c, g = lane >> 4) then holds keys 8g..8g+3 (block row 0) and
//    8g+4..8g+7 (block row 1), i.e. packed to bf16 exactly k-block g of the P operand -- no LDS round trip for P, no key permutation of V^T;
//  * KV tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a ring of FOUR 32 KiB
//    stages, requested three tiles ahead and waited for with counted s_waitcnt vmcnt: >= 2 iterations of latency hiding.
//    The DMA destination is lane-linear, so the bank-conflict swizzles are applied on the SOURCE chunk index: the K tile's 16-B chunks by
//    f(row) = bits {0, 1, 3, 4} of the row (the 16 rows a 16-lane group reads differ in exactly those bits), V^T's by (d >> 1) & 7.
//    The tile's 8 DMA instructions sit INSIDE the scheduled loop, one per ~16 MFMAs: issued as a burst at the top of the iteration
//    (behind the barrier, no MFMA in flight) they cost 6.5 % of the loop and doubled the barrier's cost (profiles/r03_attn_shape_ab.txt);
//  * the 64-key tile is consumed as two 32-key sub-tiles, software pipelined inside the wave: the matrix pipe computes S
//    of the next sub-tile while the VALU runs exp / bf16 pack of the current one, then O += V P.  One basic block;
//  * above 256 registers the MFMA accumulators live in the accumulator half of the register file, which the VALU only
//    reaches through copies, and the compiler puts such copies on the hot path as soon as ANY code multiplies O^T (the
//    usual online-softmax rescale).  So O^T is never rescaled here: the softmax reference of a query is fixed to the
//    maximum over the first 32 keys of the frame (an actual score: the largest P is >= 1, nothing underflows), later
//    scores may exceed it by up to 2^60 (P, l and O carry that factor in fp32 / bf16, exponent range 2^127), and
//    (O, m, l) go to the split workspace for attn_combine_kernel, which normalises.  A larger jump (or a NaN, T == 1)
//    makes the workgroup raise a flag instead: the 32-query kernel, launched right after as a fix-up pass, recomputes
//    exactly the flagged tiles with the classic online softmax and returns immediately everywhere else;
//  * the softmax denominator comes out of the matrix pipe too: l += 1 * P with an all-ones A operand, i.e. the sum of exactly the
//    bf16-rounded probabilities the numerator uses, accumulated the same way (docs/LOG_r01_r05.md section 4: the rounding of P then cancels
//    to first order between numerator and denominator).
//  * workgroups are renumbered so that the ~40 workgroups streaming one (clip, picked frame) K / V^T sit on ONE XCD (each XCD has
//    its own L2; dealt round-robin every XCD would read all of K / V^T: 8x the fetch traffic).
constexpr int ATT_NS = 4;                   // LDS ring stages (K 16 KiB + V^T 16 KiB each)

#ifdef PPMS_ATTN_TIMING
static __device__ long long* g_attn_dbg_dev = nullptr;          // debug builds only: [workgroup][4] wall-clock stamps (100 MHz) of wave 0
#define ATTN_STAMP(K)                                                                                                   \
    if (g_attn_dbg_dev != nullptr && __builtin_amdgcn_readfirstlane(threadIdx.x) == 0) /* all of wave 0: a uniform branch */ \
        g_attn_dbg_dev[(int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + (K)] = wall_clock64();
#else
#define ATTN_STAMP(K)
#endif
#include "attn64_asm.h"

__device__ __forceinline__ int att_kswz(int row) { return (row & 3) | ((row >> 1) & 12); }

template <bool P16>
__global__ __launch_bounds__(256, 1) void mem_attn64_kernel(const bf16_t* __restrict__ qb, const bf16_t* __restrict__ kb,
                                                             const bf16_t* __restrict__ vt_g, const int32_t* __restrict__ sel, int ksel,
                                                             float scale_log2, int n, float* __restrict__ part_o, float* __restrict__ part_ml,
                                                             int32_t* __restrict__ redo, int sps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ATTN_STAMP(0)
    constexpr int QB = 4;                         // 16-query blocks per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    // a workgroup streams the `sps` (1 or 2) consecutive picked frames of split gz = blockIdx.z (gridDim.z == ceil(ksel / sps)): two where the
    // one-frame grid would run several rounds on the chip anyway -- 40 % fewer partials to write, read and combine, one prologue / epilogue per
    // two frames.  K' of consecutive slots is contiguous ([clip][slot][key][128]): one stream of nfr * nt tiles; V^T switches frames at tile nt
    const int nsplit = gridDim.z;
    int qblk, clip, gz;
    {   // XCD-aware order.  Workgroups are dealt to the 8 XCDs round-robin by dispatch index (xcd = lin & 7; the k-th workgroup of an XCD is
        // lin >> 3) and each XCD has its own L2, so (1) the ~40 workgroups that stream the same (clip, split) K / V^T get consecutive k on
        // ONE XCD (dealt the plain way every XCD reads all of K / V^T: 8x the fetch traffic), and (2) with two-frame splits every XCD gets
        // its eighth of the two-frame ("heavy") workgroups AND its eighth of the one-frame ones, heavy first -- equal work per XCD
        // (in plain split order five XCDs hold only heavy workgroups: 1.33 instead of 1.08 ms per 1/4-scale call)
        const int gx = gridDim.x, gy = gridDim.y;
        const int nwg = gx * gy * nsplit;
        const int nfull = ksel / sps;                                   // splits with `sps` frames; one more, shorter split if ksel % sps
        const int nH = gx * gy * nfull;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int xcd = lin & 7, k = lin >> 3;
        const int kpre = xcd * (nwg >> 3) + (xcd < (nwg & 7) ? xcd : (nwg & 7));          // workgroups of the XCDs before this one
        const int hpre = xcd * (nH >> 3) + (xcd < (nH & 7) ? xcd : (nH & 7));             // heavy workgroups of the XCDs before this one
        const int hx = (nH >> 3) + (xcd < (nH & 7) ? 1 : 0);                              // heavy workgroups of this XCD
        int idx, pair;
        if (k < hx) {
            idx = hpre + k;                                             // index in the heavy list: [split][clip][qblk]
            pair = idx / gx;
            gz = pair / gy;
        } else {
            idx = (kpre - hpre) + (k - hx);                             // index in the light list: [clip][qblk]
            pair = idx / gx;
            gz = nfull + pair / gy;
        }
        qblk = idx - pair * gx;
        clip = pair % gy;
    }
    const int q0 = qblk * (64 * NW) + wave * 64;
    const int slot0 = gz * sps;
    const int nfr = (slot0 + sps <= ksel) ? sps : ksel - slot0;
    const int ntf = n / KT, nt = nfr * ntf;       // tiles per frame, tiles of this workgroup

    bf16x8 qf[QB][4];                             // Q^T fragments: query 16 b + c, channels 32 s + 8 g .. + 8
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qi = q0 + b * 16 + c;
        const int qc = qi < n ? qi : n - 1;
        const bf16_t* qp = qb + ((int64_t)clip * n + qc) * D + 8 * g;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[b][s] = *(const bf16x8*)(qp + 32 * s);
    }
    // ---- DMA sources of this thread (tile 0); LDS chunk q = i*256 + tid, i = 0..3 -----------------------------------
    //  K tile: row = q >> 4 (key), LDS position q & 15 holds source chunk (q & 15) ^ att_kswz(row)
    //  V^T tile: row d = q >> 3, LDS position q & 7 holds source chunk (q & 7) ^ ((d >> 1) & 7)     (chunk = 8 keys)
    const char* kbase = (const char*)(kb + (int64_t)(clip * ksel + slot0) * n * D);
    // V^T base of the picked frame the DMA stream is in.  Tiles are requested in non-decreasing order (0, 1, 2, then j + 3 clamped to the last tile), so ONE
    // live base is enough: when the stream crosses into the next picked frame -- a uniform branch taken nfr - 1 times per workgroup -- the base is
    // re-read from `sel`.  (Five live bases cost ten SGPRs the hand-scheduled loop does not have: the compiler spilled to scratch, and a scratch load is
    // one more entry on the vmcnt counter the LDS-DMA ring counts on; an indexed array of bases goes to scratch outright.)
    auto vframe = [&](int f) __attribute__((always_inline)) {
        // the LDS-DMA statements take the base as an SGPR pair; the words pass a statement that holds the 5 wait states a VALU-written SGPR needs
        // before a VMEM instruction may read it, should the compiler produce them on the VALU (tools/probe/lds_dma_hazard_probe.hip)
        const uint64_t a = (uint64_t)(uintptr_t)(vt_g + (int64_t)sel[clip * 5 + slot0 + f] * D * n);
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    const char* vcur = vframe(0);
    int vf = 0, vj0 = 0;                          // slot (relative to slot0) and first tile of the frame `vcur` points at
    auto v_tile = [&](int j) __attribute__((always_inline)) {
        if (j - vj0 >= ntf) {                     // (uniform; j < nt = nfr * ntf, so vf + 1 < nfr here)
            vj0 += ntf;
            ++vf;
            vcur = vframe(vf);
        }
        return vcur + (int64_t)(j - vj0) * KT * 2;
    };
    unsigned koff[4], voff[4];
    {
        const int d = tid >> 3;
        const unsigned v0 = (unsigned)(d * n * 2 + (((tid & 7) ^ ((d >> 1) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            koff[i] = (unsigned)(row * D * 2 + (((tid & 15) ^ att_kswz(row)) << 4));
            voff[i] = v0 + (unsigned)(i * 32 * n * 2);            // V^T rows d + 32 i  (128 rows x n keys x 2 B < 4 GiB)
        }
    }
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem + wave * 1024);
    auto issue_tile = [&](int j) __attribute__((always_inline)) {
        const unsigned st = lds_wave + (unsigned)((j & (ATT_NS - 1)) * ATT_STAGE);
        const char* kp = kbase + (int64_t)j * KT * D * 2;
        const char* vp = v_tile(j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(koff[i]), "s"(kp), "s"(st + (unsigned)(i * 4096)) : "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[i]), "s"(vp), "s"(st + (unsigned)(K_TILE + i * 4096)) : "memory");
    };

    f32x4 o[8][QB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int b = 0; b < QB; ++b) o[i][b] = (f32x4){0};
    // LDS addresses of this lane's fragments in stage 0 (attn64_asm.h moves them from stage to stage):
    //  K, block row b, k-step s: row 8 (c >> 2) + 4 b + (c & 3) (+ 32 rows = 8192 B for the tile's second sub-tile), chunk (4 s + g) ^ att_kswz(row)
    //  V^T, sub-tile p: row d = c (+ 16 rows = 2048 B per d block), chunk (4 p + g) ^ ((c >> 1) & 7)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    unsigned kaddr[8], vaddr[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int row = 8 * (c >> 2) + 4 * b + (c & 3);
#pragma unroll
        for (int s = 0; s < 4; ++s) kaddr[b * 4 + s] = lds0 + row * 256 + (((4 * s + g) ^ att_kswz(row)) << 4);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) vaddr[p] = lds0 + K_TILE + c * 128 + (((4 * p + g) ^ ((c >> 1) & 7)) << 4);

    issue_tile(0);
    if (nt > 1) issue_tile(1);
    if (nt > 2) issue_tile(2);
    if (nt > 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 sa[2][QB], sb[2][QB];
#pragma unroll
    for (int b = 0; b < 2; ++b)                   // S^T of sub-tile 0
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 kf = *(const bf16x8*)(smem + (kaddr[b * 4 + s] - lds0));
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const f32x4 c0 = (s == 0) ? (f32x4){0} : sa[b][q];
                sa[b][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[q][s], c0, 0, 0, 0);
            }
        }
    float negm[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {                // softmax reference: maximum over the first 32 keys of the frame
        float mx = fmaxf(fmaxf(sa[0][q][0], sa[0][q][1]), fmaxf(sa[0][q][2], sa[0][q][3]));
        mx = fmaxf(mx, fmaxf(fmaxf(sa[1][q][0], sa[1][q][1]), fmaxf(sa[1][q][2], sa[1][q][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        negm[q] = -(mx * scale_log2);
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(sa[b][q]));      // the loop's VALU reads the S^T tiles: VGPR half of the register file
    }
    u32x4 ring[4], vt[8] = {}, pf[QB] = {};
    f32x4 lacc[QB] = {(f32x4){0}, (f32x4){0}, (f32x4){0}, (f32x4){0}};
    constexpr unsigned ONE2 = P16 ? 0x3c003c00u : 0x3f803f80u;      // two ones in the format of P~
    u32x4 ones = {ONE2, ONE2, ONE2, ONE2};
    asm volatile("" : "+v"(ones));
    f32x2 pt[2], tt[2];
    attn64_prime(sa, ring, pt, tt, negm, scale_log2, kaddr);
    ATTN_STAMP(1)
    for (int j = 0; j < nt; ++j) {
        // tile j + 3 travels into the stage tile j - 1 left at the barrier; its 8 DMA instructions sit inside the substeps.  Past the frame's end
        // the last tile is fetched again (into a stage nobody reads): the instruction stream and the vmcnt bookkeeping stay unconditional
        const int jn = j + 3 < nt ? j + 3 : nt - 1;
        const unsigned st = lds_wave + (unsigned)(((j + 3) & (ATT_NS - 1)) * ATT_STAGE);
        const char* kp = kbase + (int64_t)jn * KT * D * 2;
        const char* vp = v_tile(jn);
        const unsigned dst[8] = {st, st + 4096u, st + 8192u, st + 12288u, st + K_TILE, st + K_TILE + 4096u, st + K_TILE + 8192u, st + K_TILE + 12288u};
        const int delta = ((j + 1) & (ATT_NS - 1)) ? ATT_STAGE : -(ATT_NS - 1) * ATT_STAGE;      // stage of tile j -> stage of tile j + 1
        attn64_substep<0, P16>(sa, sb, qf, o, ring, vt, pf, pt, tt, lacc, ones, negm, scale_log2, kaddr, vaddr, delta, koff, voff, kp, vp, dst);
        attn64_substep<1, P16>(sb, sa, qf, o, ring, vt, pf, pt, tt, lacc, ones, negm, scale_log2, kaddr, vaddr, delta, koff, voff, kp, vp, dst);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tiles <= j + 2 have landed (this thread's share; the barrier covers the others')
#if !defined(PPMS_ATTN_NOSYNC)
        __builtin_amdgcn_s_barrier();
#endif
        if (j + 1 == nt) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // drain the matrix pipe in front of whatever the compiler places at the exit
    }
    attn64_tail();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may be in flight when the workgroup's LDS is handed on
    ATTN_STAMP(2)
    if constexpr (P16) {
#pragma unroll
        for (int dblk = 0; dblk < 8; ++dblk)      // O^T[:, qb 3] += V^T P of the last sub-tile
            o[dblk][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vt[dblk]), __builtin_bit_cast(f16x8, pf[3]), o[dblk][3], 0, 0, 0);
        lacc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ones), __builtin_bit_cast(f16x8, pf[3]), lacc[3], 0, 0, 0);
    } else {
#pragma unroll
        for (int dblk = 0; dblk < 8; ++dblk)
            o[dblk][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vt[dblk]), __builtin_bit_cast(bf16x8, pf[3]), o[dblk][3], 0, 0, 0);
        lacc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pf[3]), lacc[3], 0, 0, 0);
    }
    // a score too far above the reference (or a NaN) anywhere shows in the sum: the fix-up pass redoes the tile.  bf16 P~: more than 2^60 above.
    // fp16 P~: a probability beyond fp16's largest finite value (65 504 ~ 2^16) converts to +inf and the sum with it, so the bound is "l finite"
    // (sums of finite fp16 values stay far below 2^60: at most 2^16 per key)
    bool bail = false;
#pragma unroll
    for (int q = 0; q < QB; ++q) bail = bail || !(lacc[q][0] <= 0x1p60f);
    {
        const int flag = __syncthreads_or(bail);
        if (tid == 0) {
            int32_t* f = redo + ((int64_t)(clip * nsplit + gz) * gridDim.x + qblk) * 2;
            f[0] = flag ? 1 : 0;
            f[1] = flag ? 1 : 0;
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int qi = q0 + q * 16 + c;
        if (qi >= n) continue;
        const int64_t row = ((int64_t)clip * nsplit + gz) * n + qi;
        float* po = part_o + row * D + 4 * g;
#pragma unroll
        for (int dblk = 0; dblk < 8; ++dblk) *(f32x4*)(po + dblk * 16) = o[dblk][q];
        if (g == 0) {
            part_ml[row * 2] = -negm[q];
            part_ml[row * 2 + 1] = lacc[q][0];
        }
    }
    ATTN_STAMP(3)
}

// merges the per-frame partials of split mode: O = sum_s O_s 2^(m_s - m), l = sum_s l_s 2^(m_s - m); then the same
// epilogue as the fused kernel (hid = bf16(O / l), mfg = mf + beta * hid).  One thread = one query x 8 channels.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, int nsplit,
                                                           const float* __restrict__ beta_p, ppms_sp mf, ppms_sp mfg,
                                                           bf16_t* __restrict__ out_bf16, int n, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g8 = (int)(idx & 15);
    const int64_t pix = idx >> 4;                       // clip * n + query
    const int64_t clip = pix / n;
    const int64_t q = pix - clip * n;
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part_ml[(((clip * nsplit + s) * n) + q) * 2]);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
        const int64_t row = (clip * nsplit + s) * n + q;
        const float ms = part_ml[row * 2];
        const float wgt = (ms == m) ? 1.0f : __builtin_amdgcn_exp2f(ms - m);      // NaN partials (T == 1) stay NaN
        l += part_ml[row * 2 + 1] * wgt;
        const f32x4 a = *(const f32x4*)(part_o + row * D + g8 * 8), b = *(const f32x4*)(part_o + row * D + g8 * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] += a[j] * wgt;
            acc[4 + j] += b[j] * wgt;
        }
    }
    const float inv_l = 1.0f / l, beta = beta_p[0];
    const int d = g8 * 8;
    bf16x8 hid, oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) hid[j] = (bf16_t)(acc[j] * inv_l);
    if (mf.hi != nullptr) {                             // (uniform) mfg = mf + beta * hid
        const bf16x8 mh = *(const bf16x8*)((const bf16_t*)mf.hi + pix * mf.ld + d), ml = *(const bf16x8*)((const bf16_t*)mf.lo + pix * mf.ld + d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = join_bf16(mh[j], ml[j]) + beta * (float)hid[j];
            bf16_t hh, ll;
            split_bf16(y, hh, ll);
            oh[j] = hh;
            ol[j] = ll;
        }
    } else {                                            // the view receives hid itself: bf16-exact values, an all-zero lo plane
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            oh[j] = hid[j];
            ol[j] = (bf16_t)0.0f;
        }
    }
    if (out_bf16) *(bf16x8*)(out_bf16 + pix * D + d) = hid;
    *(bf16x8*)((bf16_t*)mfg.hi + pix * mfg.ld + d) = oh;
    *(bf16x8*)((bf16_t*)mfg.lo + pix * mfg.ld + d) = ol;
}

}  // namespace

#ifdef PPMS_ATTN_TIMING
extern "C" void ppms_debug_attn_timing(long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg_dev), &p, sizeof(p)); }   // tools/attn_phase_probe.py
#endif
template <bool P16>
static void launch_mem_attn(const void* qb, const void* kb, const void* vt, const int32_t* sel, int ksel, float scale_log2, const float* beta, ppms_sp mf,
                            ppms_sp mfg, void* out_bf16, int T, int n, void* split_ws, int frames_per_workgroup, hipStream_t st) {
    static ppms_device_once once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)mem_attn_kernel<false, P16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_STAGE);
        (void)hipFuncSetAttribute((const void*)mem_attn_kernel<true, P16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_STAGE);
        (void)hipFuncSetAttribute((const void*)mem_attn64_kernel<P16>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_NS * ATT_STAGE);
    });
    // split over the picked frames when a workspace is given (ppms_mem_attn_workspace_bytes) and there is more than one
    const bool use64 = split_ws != nullptr && n % KT == 0;      // (its state lives in the workspace partials)
    const bool split = use64 || (split_ws != nullptr && ksel > 1);
    float* part_o = split ? (float*)split_ws : nullptr;
    float* part_ml = split ? part_o + (size_t)T * ksel * n * D : nullptr;
    int nsp = ksel;                                             // partial sets per clip
    if (use64) {
        const int g64 = (int)ceil_div(n, 64 * NW);
        int32_t* redo = (int32_t*)(part_ml + (size_t)T * ksel * n * 2);
        // two picked frames per workgroup where the one-frame grid is at least two rounds of the chip (the 1/4 scale: 1000 workgroups)
        const int sps = frames_per_workgroup ? frames_per_workgroup : ((g64 * T * ksel >= 2 * ppms_num_cus() && ksel > 1) ? 2 : 1);
        nsp = (int)ceil_div(ksel, sps);
        dim3 grid64(g64, T, nsp), grid32(ceil_div(n, QW * NW), T, nsp);
        hipLaunchKernelGGL(mem_attn64_kernel<P16>, grid64, dim3(256), ATT_NS * ATT_STAGE, st, (const bf16_t*)qb, (const bf16_t*)kb, (const bf16_t*)vt,
                           sel, ksel, scale_log2, n, part_o, part_ml, redo, sps);
        hipLaunchKernelGGL((mem_attn_kernel<false, P16>), grid32, dim3(256), 2 * ATT_STAGE, st, (const bf16_t*)qb, (const bf16_t*)kb, (const bf16_t*)vt,
                           sel, ksel, scale_log2, beta, mf, mfg, (bf16_t*)out_bf16, n, part_o, part_ml, redo, 2 * g64, sps);
    } else {
        dim3 grid(ceil_div(n, QW * NW), T, split ? ksel : 1);
        if (n % KT)
            hipLaunchKernelGGL((mem_attn_kernel<true, P16>), grid, dim3(256), 2 * ATT_STAGE, st, (const bf16_t*)qb, (const bf16_t*)kb, (const bf16_t*)vt,
                               sel, ksel, scale_log2, beta, mf, mfg, (bf16_t*)out_bf16, n, part_o, part_ml, (int32_t*)nullptr, 0, 1);
        else
            hipLaunchKernelGGL((mem_attn_kernel<false, P16>), grid, dim3(256), 2 * ATT_STAGE, st, (const bf16_t*)qb, (const bf16_t*)kb, (const bf16_t*)vt,
                               sel, ksel, scale_log2, beta, mf, mfg, (bf16_t*)out_bf16, n, part_o, part_ml, (int32_t*)nullptr, 0, 1);
    }
    if (split) {
        const int64_t total = (int64_t)T * n * 16;
        hipLaunchKernelGGL(attn_combine_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, part_o, part_ml, nsp, beta, mf, mfg,
                           (bf16_t*)out_bf16, n, total);
    }
}

extern "C" int ppms_mem_attn(const void* qb, const void* kb, const void* vt, const int32_t* sel, int ksel, float scale, const float* beta,
                             ppms_sp mf, ppms_sp mfg, void* out_bf16, int T, int n, void* split_ws, int frames_per_workgroup, int p_format, void* stream) {
    PPMS_REQUIRE(qb && kb && vt && sel && beta, "mem_attn: null operand");
    PPMS_REQUIRE(ksel >= 1 && ksel <= 5 && T >= 1 && n >= 1, "mem_attn: bad sizes ksel=%d T=%d n=%d", ksel, T, n);
    PPMS_REQUIRE(frames_per_workgroup >= 0 && frames_per_workgroup <= 5, "mem_attn: frames_per_workgroup must be 0 (automatic) or 1 .. 5, got %d", frames_per_workgroup);
    PPMS_REQUIRE(p_format == PPMS_ATTN_P_BF16 || p_format == PPMS_ATTN_P_FP16, "mem_attn: p_format must be PPMS_ATTN_P_BF16 (0) or PPMS_ATTN_P_FP16 (1), got %d", p_format);
    PPMS_REQUIRE(((mf.hi && mf.lo && mf.ld % 8 == 0) || (!mf.hi && !mf.lo)) && mfg.hi && mfg.lo && mfg.ld % 8 == 0,
                 "mem_attn: mf (or a NULL view: the output is hid itself) / mfg must be 16-B aligned SP views");
    const float scale_log2 = scale * 1.4426950408889634f;
    if (p_format == PPMS_ATTN_P_FP16)
        launch_mem_attn<true>(qb, kb, vt, sel, ksel, scale_log2, beta, mf, mfg, out_bf16, T, n, split_ws, frames_per_workgroup, (hipStream_t)stream);
    else
        launch_mem_attn<false>(qb, kb, vt, sel, ksel, scale_log2, beta, mf, mfg, out_bf16, T, n, split_ws, frames_per_workgroup, (hipStream_t)stream);
    return ppms_check_launch("mem_attn");
}

// partial O (T*ksel*n*128 fp32) + partial (m, l) (T*ksel*n*2 fp32) + redo flags of the 64-query kernel (2 per 256-query block)
extern "C" int64_t ppms_mem_attn_workspace_bytes(int T, int ksel, int n) {
    return (int64_t)T * ksel * n * (D + 2) * 4 + (int64_t)T * ksel * ceil_div(n, 64 * NW) * 2 * 4;
}
