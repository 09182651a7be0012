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

// TAIL = (n % 64 != 0): only that instantiation carries the per-key masking of a frame's last tile
template <bool TAIL>
__global__ __launch_bounds__(256, 2) void mem_attn_kernel(const bf16_t* __restrict__ qb, const bf16_t* __restrict__ kb,
                                                          const bf16_t* __restrict__ vt, const int32_t* __restrict__ sel, int ksel,
                                                          float scale_log2, const float* __restrict__ beta_p, ppms_sp mf, ppms_sp mfg,
                                                          bf16_t* __restrict__ out_bf16, int n, float* __restrict__ part_o,
                                                          float* __restrict__ part_ml) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int clip = blockIdx.y;
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
    // split mode (part_o != null): gridDim.z = ksel, this workgroup reads ONE picked frame and writes unnormalised
    // partials; a combine kernel merges them (more, smaller work units: 400 -> 2000 at 320x512, and the small scales
    // get ksel x the parallelism)
    const int nsplit = gridDim.z;
    const int slot0 = (nsplit > 1) ? (int)blockIdx.z : 0;
    const int it0 = slot0 * tpf;
    const int ntile = (nsplit > 1) ? it0 + tpf : ksel * tpf;

    u32x4 rk[4], rv[4];
    auto load_tile = [&](int it) {
        const int slot = it / tpf;
        const int key0 = (it - slot * tpf) * KT;
        const int frame = sel[clip * 5 + slot];
        // K': [clip][slot][key][128], 256-B rows; thread -> rows (tid>>4) + 16 i, chunk tid&15
        const bf16_t* kbase = kb + ((int64_t)(clip * ksel + slot) * n) * D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = key0 + (tid >> 4) + 16 * i;
            rk[i] = key < n ? gload16(kbase + (int64_t)key * D + (tid & 15) * 8) : (u32x4){0, 0, 0, 0};
        }
        // V^T: [frame][d][n], this tile = 64 keys (128 B) of every d row; thread -> d = (tid>>3) + 32 i, chunk tid&7
        const bf16_t* vbase = vt + (int64_t)frame * D * n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = (tid >> 3) + 32 * i;
            rv[i] = load16_guard(vbase + (int64_t)d * n, key0 + (tid & 7) * 8, n, n_vec);
        }
    };
    auto store_tile = [&](int stage) {
        char* ks = smem + stage * ATT_STAGE;
        char* vs = ks + K_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            *(u32x4*)(ks + row * 256 + (((tid & 15) ^ (row & 15)) << 4)) = rk[i];
        }
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

    // one KV tile; MASKED is only instantiated for the last tile of a frame when n % 64 != 0, so the steady-state
    // loop carries no per-key compare/select work (the softmax VALU stream, not the MFMA pipe, is the critical path:
    // every v_cndmask / range fix-up removed here is ~1 % of the kernel)
    auto process_tile = [&](const char* ks, const char* vs, int key0) {
        // ---- S^T[key][query] = K Q^T ---------------------------------------------------------------
        f32x16 st[2];
        st[0] = (f32x16){0};
        st[1] = (f32x16){0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk) {
                const int row = kblk * 32 + r;
                const bf16x8 kf = *(const bf16x8*)(ks + row * 256 + (((2 * s + h) ^ (row & 15)) << 4));
                st[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kblk], 0, 0, 0);
            }
        }
        if (TAIL && key0 + KT > n) {
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int key = key0 + kblk * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (key >= n) st[kblk][g] = -INFINITY;
                }
        }
        // ---- online softmax (fp32), one query per lane pair (r, r+32) --------------------------------
        float mx = st[0][0];
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, st[kblk][g]);
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
        const float neg_m = -m_run;
        float psum = 0.0f;
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                // raw v_exp_f32: arguments are <= 0, results in [0,1]; values below 2^-126 flush to 0 (irrelevant for P)
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kblk][g], scale_log2, neg_m));
                st[kblk][g] = p;
                psum += p;
            }
        l_run += psum;

        // ---- O^T[d][query] += V^T P^T ; P^T fragments come straight from the S^T accumulators -------------
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)st[kblk][8 * s2 + j];
                const int chunk = (kblk * 2 + s2) * 2 + h;
#pragma unroll
                for (int dblk = 0; dblk < 4; ++dblk) {
                    const int d = dblk * 32 + r;
                    const bf16x8 vf = *(const bf16x8*)(vs + d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4));
                    o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dblk], 0, 0, 0);
                }
            }
        }
    };

    load_tile(it0);
    store_tile(it0 & 1);
    __syncthreads();
    for (int it = it0; it < ntile; ++it) {
        const bool more = it + 1 < ntile;
        if (more) load_tile(it + 1);
        const char* ks = smem + (it & 1) * ATT_STAGE;
        const char* vs = ks + K_TILE;
        const int key0 = (it % tpf) * KT;
        process_tile(ks, vs, key0);
        if (more) store_tile((it + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: hid = bf16(O / l); mfg = mf + beta * hid ---------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (qi >= n) return;
    if (part_o != nullptr) {
        const int64_t row = ((int64_t)clip * nsplit + slot0) * n + qi;
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
            const bf16x4 a = *(const bf16x4*)(mh + d), b = *(const bf16x4*)(ml + d);
            bf16x4 oh, ol;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = join_bf16(a[j], b[j]) + beta * (float)hid[j];
                bf16_t hh, ll;
                split_bf16(y, hh, ll);
                oh[j] = hh;
                ol[j] = ll;
            }
            *(bf16x4*)(gh + d) = oh;
            *(bf16x4*)(gl + d) = ol;
        }
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
    const bf16x8 mh = *(const bf16x8*)((const bf16_t*)mf.hi + pix * mf.ld + d), ml = *(const bf16x8*)((const bf16_t*)mf.lo + pix * mf.ld + d);
    bf16x8 hid, oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hid[j] = (bf16_t)(acc[j] * inv_l);
        const float y = join_bf16(mh[j], ml[j]) + beta * (float)hid[j];
        bf16_t hh, ll;
        split_bf16(y, hh, ll);
        oh[j] = hh;
        ol[j] = ll;
    }
    if (out_bf16) *(bf16x8*)(out_bf16 + pix * D + d) = hid;
    *(bf16x8*)((bf16_t*)mfg.hi + pix * mfg.ld + d) = oh;
    *(bf16x8*)((bf16_t*)mfg.lo + pix * mfg.ld + d) = ol;
}

}  // namespace

extern "C" int ppms_mem_attn(const void* qb, const void* kb, const void* vt, const int32_t* sel, int ksel, float scale, const float* beta,
                             ppms_sp mf, ppms_sp mfg, void* out_bf16, int T, int n, void* split_ws, void* stream) {
    PPMS_REQUIRE(qb && kb && vt && sel && beta, "mem_attn: null operand");
    PPMS_REQUIRE(ksel >= 1 && ksel <= 5 && T >= 1 && n >= 1, "mem_attn: bad sizes ksel=%d T=%d n=%d", ksel, T, n);
    PPMS_REQUIRE(mf.hi && mf.lo && mfg.hi && mfg.lo && mf.ld % 8 == 0 && mfg.ld % 8 == 0, "mem_attn: mf / mfg must be 16-B aligned SP views");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mem_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_STAGE);
        (void)hipFuncSetAttribute((const void*)mem_attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_STAGE);
        attr_set = true;
    }
    const float scale_log2 = scale * 1.4426950408889634f;
    // split over the picked frames when a workspace is given (ppms_mem_attn_workspace_bytes) and there is more than one
    const bool split = split_ws != nullptr && ksel > 1;
    float* part_o = split ? (float*)split_ws : nullptr;
    float* part_ml = split ? part_o + (size_t)T * ksel * n * D : nullptr;
    dim3 grid(ceil_div(n, QW * NW), T, split ? ksel : 1);
    hipStream_t st = (hipStream_t)stream;
    if (n % KT)
        hipLaunchKernelGGL(mem_attn_kernel<true>, grid, dim3(256), 2 * ATT_STAGE, st, (const bf16_t*)qb, (const bf16_t*)kb, (const bf16_t*)vt, sel,
                           ksel, scale_log2, beta, mf, mfg, (bf16_t*)out_bf16, n, part_o, part_ml);
    else
        hipLaunchKernelGGL(mem_attn_kernel<false>, grid, dim3(256), 2 * ATT_STAGE, st, (const bf16_t*)qb, (const bf16_t*)kb, (const bf16_t*)vt, sel,
                           ksel, scale_log2, beta, mf, mfg, (bf16_t*)out_bf16, n, part_o, part_ml);
    if (split) {
        const int64_t total = (int64_t)T * n * 16;
        hipLaunchKernelGGL(attn_combine_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, part_o, part_ml, ksel, beta, mf, mfg,
                           (bf16_t*)out_bf16, n, total);
    }
    return ppms_check_launch("mem_attn");
}

extern "C" int64_t ppms_mem_attn_workspace_bytes(int T, int ksel, int n) { return (int64_t)T * ksel * n * (D + 2) * 4; }
