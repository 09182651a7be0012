// The packed-fp32 finding (docs/LOG_r01_r05.md section 5) on the kernel it was found in: the library's bilinear resize, compiled here WITH
// packed fp32 formation (the compiler default), in three forms:
//   variant 0: the kernel as it is in small_ops.hip;
//   variant 1: the same, plus the four taps it loaded written to a debug buffer (were the LOADED values wrong, or the arithmetic?);
//   variant 2: the same arithmetic with packed formation defeated (each product pinned through an empty asm) -- the control.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int VAR>
__global__ __launch_bounds__(256) void bilinear_probe_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ taps, int H, int W,
                                                             int OH, int OW, int align, float sh, float sw, float mul, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    float fy = align ? sh * oy : fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
    float fx = align ? sw * ox : fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = src + plane * H * W;
    const float s00 = s[y0 * W + x0], s01 = s[y0 * W + x1], s10 = s[y1 * W + x0], s11 = s[y1 * W + x1];
    float v;
    if (VAR == 2) {
        float a = hx * s00, b = lx * s01, c = hx * s10, d = lx * s11;
        asm volatile("" : "+v"(a));
        asm volatile("" : "+v"(b));
        asm volatile("" : "+v"(c));
        asm volatile("" : "+v"(d));
        v = hy * (a + b) + ly * (c + d);
    } else {
        v = hy * (hx * s00 + lx * s01) + ly * (hx * s10 + lx * s11);
    }
    dst[idx] = mul * v;
    if (VAR == 1) {
        taps[4 * idx + 0] = s00;
        taps[4 * idx + 1] = s01;
        taps[4 * idx + 2] = s10;
        taps[4 * idx + 3] = s11;
    }
}

// Hand-written core with the compiler's schedule of variant 0: four loads into the register pairs (v40, v41) / (v42, v43) in the order
// v40, v43, v42, v41; a packed (KIND 0) or two scalar (KIND 1) multiplies of the pair (v42, v43) behind vmcnt(1) and of (v40, v41) behind
// vmcnt(0), each after NOPS extra wait states (0: none).  The load destinations are zeroed first, so a read that beats the load sees 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define NOP_STR(N) "s_nop " #N "\n\t"
template <int KIND, int NOPS>
__global__ __launch_bounds__(256) void bilinear_asm_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int OH, int OW, float sh, float sw,
                                                           int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
    float fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = src + plane * H * W;
    const float *p00 = s + y0 * W + x0, *p01 = s + y0 * W + x1, *p10 = s + y1 * W + x0, *p11 = s + y1 * W + x1;
    const f32x2 wy = {hy, ly};
    f32x2 ta, tb;
#define LOADS_(A, B, C_, D)                          \
    "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t" \
    "global_load_dword v40, %" #A ", off\n\t"        \
    "global_load_dword v43, %" #D ", off\n\t"        \
    "global_load_dword v42, %" #B ", off\n\t"        \
    "global_load_dword v41, %" #C_ ", off\n\t"
#define BODY(NOPA, NOPB)                                                                                                            \
    if (KIND == 0)                                                                                                                  \
        asm volatile(LOADS_(2, 3, 4, 5) "s_waitcnt vmcnt(1)\n\t" NOPA "v_pk_mul_f32 %1, v[42:43], %6\n\t"                                        \
                           "s_waitcnt vmcnt(0)\n\t" NOPB "v_pk_mul_f32 %0, v[40:41], %6"                                             \
                     : "=&v"(ta), "=&v"(tb) : "v"(p00), "v"(p01), "v"(p10), "v"(p11), "v"(wy) : "memory", "v40", "v41", "v42", "v43"); \
    else                                                                                                                            \
        asm volatile(LOADS_(4, 5, 6, 7) "s_waitcnt vmcnt(1)\n\t" NOPA "v_mul_f32 %2, v42, %8\n\tv_mul_f32 %3, v43, %9\n\t"                         \
                           "s_waitcnt vmcnt(0)\n\t" NOPB "v_mul_f32 %0, v40, %8\n\tv_mul_f32 %1, v41, %9"                              \
                     : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(tb[0]), "=&v"(tb[1]) : "v"(p00), "v"(p01), "v"(p10), "v"(p11), "v"(hy), "v"(ly)     \
                     : "memory", "v40", "v41", "v42", "v43");
    if (NOPS == 0) { BODY("", "") }
    else if (NOPS == 1) { BODY(NOP_STR(0), NOP_STR(0)) }
    else if (NOPS == 2) { BODY(NOP_STR(1), NOP_STR(1)) }
    else if (NOPS == 4) { BODY(NOP_STR(3), NOP_STR(3)) }
    else { BODY(NOP_STR(7), NOP_STR(7)) }
    dst[idx] = hx * (ta[0] + ta[1]) + lx * (tb[0] + tb[1]);
}

// The whole tail of variant 0 in the compiler's own order: two packed multiplies of the loaded pairs, the swizzled packed add
// (op_sel:[0,1] op_sel_hi:[1,0]: lo = X.lo + Y.hi, hi = X.hi + Y.lo), a packed multiply by (lx, hx), a scalar add.
//   TAIL 0: exactly that, with the compiler's "s_nop 0" after every packed op;   TAIL 1: the swizzled packed add replaced by two v_add_f32;
//   TAIL 2: "s_nop 3" instead of "s_nop 0";   TAIL 3: no nops at all;   TAIL 4: TAIL 0 with "s_nop 7" between the last wait and its multiply
template <int TAIL>
__global__ __launch_bounds__(256) void bilinear_tail_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int OH, int OW, float sh, float sw,
                                                            int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
    float fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = src + plane * H * W;
    const float *p00 = s + y0 * W + x0, *p01 = s + y0 * W + x1, *p10 = s + y1 * W + x0, *p11 = s + y1 * W + x1;
    const f32x2 wy = {hy, ly}, wx = {lx, hx};
    float out;
#define TAIL_ASM(NOPW, NOP, ADD)                                                                                                    \
    asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"                                    \
                 "global_load_dword v40, %1, off\n\t" /* s00 */                                                                       \
                 "global_load_dword v43, %3, off\n\t" /* s10 */                                                                       \
                 "global_load_dword v42, %2, off\n\t" /* s01 */                                                                       \
                 "global_load_dword v41, %4, off\n\t" /* s11 */                                                                       \
                 "s_waitcnt vmcnt(1)\n\t"                                                                                             \
                 "v_pk_mul_f32 v[44:45], v[42:43], %5\n\t"                                                                            \
                 "s_waitcnt vmcnt(0)\n\t" NOPW                                                                                        \
                 "v_pk_mul_f32 v[46:47], v[40:41], %5\n\t" NOP ADD NOP                                                                \
                 "v_pk_mul_f32 v[50:51], %6, v[48:49]\n\t" NOP                                                                        \
                 "v_add_f32 %0, v51, v50"                                                                                             \
                 : "=&v"(out) : "v"(p00), "v"(p01), "v"(p10), "v"(p11), "v"(wy), "v"(wx)                                                \
                 : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
#define PK_ADD_SWZ "v_pk_add_f32 v[48:49], v[44:45], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define SC_ADD_SWZ "v_add_f32 v48, v44, v47\n\tv_add_f32 v49, v45, v46\n\t"
    if (TAIL == 0) { TAIL_ASM("", "s_nop 0\n\t", PK_ADD_SWZ) }
    else if (TAIL == 1) { TAIL_ASM("", "s_nop 0\n\t", SC_ADD_SWZ) }
    else if (TAIL == 2) { TAIL_ASM("", "s_nop 3\n\t", PK_ADD_SWZ) }
    else if (TAIL == 3) { TAIL_ASM("", "", PK_ADD_SWZ) }
    else { TAIL_ASM("s_nop 7\n\t", "s_nop 0\n\t", PK_ADD_SWZ) }
    dst[idx] = out;
}

// Variant 0's instruction stream from the loads to the result, register for register (REGS 0: v2-v7, v12-v15 as the compiler chose;
// REGS 1: the same stream 32 registers higher), the weights formed by VALU ops between the loads and the first wait as in the original.
template <int REGS>
__global__ __launch_bounds__(256) void bilinear_replica_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int OH, int OW, float sh,
                                                               float sw, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
    float fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float* s = src + plane * H * W;
    const float *p00 = s + y0 * W + x0, *p01 = s + y0 * W + x1, *p10 = s + y1 * W + x0, *p11 = s + y1 * W + x1;
    float out;
#define REPLICA(R2, R3, R4, R5, R6, R7, R12, R13, R14, R15, P45, P67, P23, P1213, P1415, CLOB...)                                     \
    asm volatile("global_load_dword " R12 ", %1, off\n\t"                                                                             \
                 "global_load_dword " R15 ", %3, off\n\t"                                                                             \
                 "global_load_dword " R14 ", %2, off\n\t"                                                                             \
                 "global_load_dword " R13 ", %4, off\n\t"                                                                             \
                 "v_mov_b32 " R5 ", %5\n\t"                                                                                           \
                 "v_mov_b32 " R2 ", %6\n\t"                                                                                           \
                 "v_sub_f32 " R4 ", 1.0, " R5 "\n\t"                                                                                  \
                 "v_sub_f32 " R3 ", 1.0, " R2 "\n\t"                                                                                  \
                 "s_waitcnt vmcnt(1)\n\t"                                                                                             \
                 "v_pk_mul_f32 " P67 ", " P1415 ", " P45 "\n\t"                                                                       \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                             \
                 "v_pk_mul_f32 " P45 ", " P1213 ", " P45 "\n\t"                                                                       \
                 "s_nop 0\n\t"                                                                                                        \
                 "v_pk_add_f32 " P45 ", " P67 ", " P45 " op_sel:[0,1] op_sel_hi:[1,0]\n\t"                                            \
                 "s_nop 0\n\t"                                                                                                        \
                 "v_pk_mul_f32 " P23 ", " P23 ", " P45 "\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                                        \
                 "v_add_f32 %0, " R3 ", " R2                                                                                          \
                 : "=&v"(out) : "v"(p00), "v"(p01), "v"(p10), "v"(p11), "v"(ly), "v"(lx) : "memory", CLOB);
    if (REGS == 0) {
        REPLICA("v2", "v3", "v4", "v5", "v6", "v7", "v12", "v13", "v14", "v15", "v[4:5]", "v[6:7]", "v[2:3]", "v[12:13]", "v[14:15]", "v2", "v3", "v4", "v5", "v6",
                "v7", "v12", "v13", "v14", "v15")
    } else {
        REPLICA("v34", "v35", "v36", "v37", "v38", "v39", "v44", "v45", "v46", "v47", "v[36:37]", "v[38:39]", "v[34:35]", "v[44:45]", "v[46:47]", "v34", "v35",
                "v36", "v37", "v38", "v39", "v44", "v45", "v46", "v47")
    }
    dst[idx] = out;
}

// Bisection of the failing stream (all in the low registers, <= 24 VGPRs so that the wave still fits beside the conv's waves):
//   BIS 0: sentinel 2.0 in the load destinations before the loads + the four products dumped (which one is wrong, and is it a stale 2.0 * w?)
//   BIS 1: s_nop 7 behind both waits;   BIS 2: the two packed multiplies of the loaded pairs as scalar multiplies;
//   BIS 3: the swizzled packed add as scalar adds;   BIS 4: the last packed multiply as scalar multiplies;
//   BIS 5: every packed op scalar (the control in the same registers)
template <int BIS>
__global__ __launch_bounds__(256) void bilinear_bisect_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ comps, int H, int W,
                                                              int OH, int OW, float sh, float sw, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
    float fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float* s = src + plane * H * W;
    const float *p00 = s + y0 * W + x0, *p01 = s + y0 * W + x1, *p10 = s + y1 * W + x0, *p11 = s + y1 * W + x1;
    float out, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
#define PK_W1 "v_pk_mul_f32 v[6:7], v[14:15], v[4:5]\n\t"
#define SC_W1 "v_mul_f32 v6, v14, v4\n\tv_mul_f32 v7, v15, v5\n\t"
#define PK_W0 "v_pk_mul_f32 v[4:5], v[12:13], v[4:5]\n\t"
#define SC_W0 "v_mul_f32 v4, v12, v4\n\tv_mul_f32 v5, v13, v5\n\t"
#define PK_ADD "v_pk_add_f32 v[4:5], v[6:7], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define SC_ADD "v_add_f32 v8, v6, v5\n\tv_add_f32 v5, v7, v4\n\tv_mov_b32 v4, v8\n\t"
#define PK_MUL "v_pk_mul_f32 v[2:3], v[2:3], v[4:5]\n\t"
#define SC_MUL "v_mul_f32 v2, v2, v4\n\tv_mul_f32 v3, v3, v5\n\t"
#define NOP0 "s_nop 0\n\t"
#define BISECT(PREFILL, NOPW, W1, W0, AFTER_W0, ADD, MUL, DUMP)                                                                       \
    asm volatile(PREFILL                                                                                                              \
                 "global_load_dword v12, %5, off\n\t"                                                                                 \
                 "global_load_dword v15, %7, off\n\t"                                                                                 \
                 "global_load_dword v14, %6, off\n\t"                                                                                 \
                 "global_load_dword v13, %8, off\n\t"                                                                                 \
                 "v_mov_b32 v5, %9\n\t"                                                                                               \
                 "v_mov_b32 v2, %10\n\t"                                                                                              \
                 "v_sub_f32 v4, 1.0, v5\n\t"                                                                                          \
                 "v_sub_f32 v3, 1.0, v2\n\t"                                                                                          \
                 "s_waitcnt vmcnt(1)\n\t" NOPW W1 "s_waitcnt vmcnt(0)\n\t" NOPW W0 AFTER_W0 ADD NOP0 MUL NOP0 "v_add_f32 %0, v3, v2\n\t" DUMP         \
                 : "=&v"(out), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)                                                                    \
                 : "v"(p00), "v"(p01), "v"(p10), "v"(p11), "v"(ly), "v"(lx)                                                             \
                 : "memory", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v12", "v13", "v14", "v15");
    if (BIS == 0) {
        BISECT("v_mov_b32 v12, 2.0\n\tv_mov_b32 v13, 2.0\n\tv_mov_b32 v14, 2.0\n\tv_mov_b32 v15, 2.0\n\t", "", PK_W1, PK_W0, "v_mov_b32 v9, v4\n\tv_mov_b32 v10, v5\n\t",
               PK_ADD, PK_MUL, "v_mov_b32 %1, v6\n\tv_mov_b32 %2, v7\n\tv_mov_b32 %3, v9\n\tv_mov_b32 %4, v10")
    } else if (BIS == 1) { BISECT("", "s_nop 7\n\t", PK_W1, PK_W0, NOP0, PK_ADD, PK_MUL, "")
    } else if (BIS == 2) { BISECT("", "", SC_W1, SC_W0, NOP0, PK_ADD, PK_MUL, "")
    } else if (BIS == 3) { BISECT("", "", PK_W1, PK_W0, NOP0, SC_ADD, PK_MUL, "")
    } else if (BIS == 4) { BISECT("", "", PK_W1, PK_W0, NOP0, PK_ADD, SC_MUL, "")
    } else { BISECT("", "", SC_W1, SC_W0, NOP0, SC_ADD, SC_MUL, "") }
    dst[idx] = out;
    if (BIS == 0) {
        comps[4 * idx + 0] = c0;      // X.lo = s01 * hy
        comps[4 * idx + 1] = c1;      // X.hi = s10 * ly
        comps[4 * idx + 2] = c2;      // Y.lo = s00 * hy
        comps[4 * idx + 3] = c3;      // Y.hi = s11 * ly
    }
}

template <int BIS>
static void launch_bisect(const float* src, float* dst, float* comps, int NC, int H, int W, int OH, int OW, hipStream_t st) {
    const int64_t n = (int64_t)NC * OH * OW;
    hipLaunchKernelGGL((bilinear_bisect_kernel<BIS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, comps, H, W, OH, OW, (float)H / (float)OH,
                       (float)W / (float)OW, n);
}

// variants 50..55: bilinear_bisect_kernel<0..5>; comps: [n][4] floats (written by variant 50 only)
extern "C" int bilinear_bisect_launch(int variant, const float* src, float* dst, float* comps, int NC, int H, int W, int OH, int OW, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 50: launch_bisect<0>(src, dst, comps, NC, H, W, OH, OW, st); break;
        case 51: launch_bisect<1>(src, dst, comps, NC, H, W, OH, OW, st); break;
        case 52: launch_bisect<2>(src, dst, comps, NC, H, W, OH, OW, st); break;
        case 53: launch_bisect<3>(src, dst, comps, NC, H, W, OH, OW, st); break;
        case 54: launch_bisect<4>(src, dst, comps, NC, H, W, OH, OW, st); break;
        case 55: launch_bisect<5>(src, dst, comps, NC, H, W, OH, OW, st); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

template <int REGS>
static void launch_replica(const float* src, float* dst, int NC, int H, int W, int OH, int OW, hipStream_t st) {
    const int64_t n = (int64_t)NC * OH * OW;
    hipLaunchKernelGGL((bilinear_replica_kernel<REGS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, H, W, OH, OW, (float)H / (float)OH,
                       (float)W / (float)OW, n);
}

template <int TAIL>
static void launch_tail(const float* src, float* dst, int NC, int H, int W, int OH, int OW, hipStream_t st) {
    const int64_t n = (int64_t)NC * OH * OW;
    hipLaunchKernelGGL((bilinear_tail_kernel<TAIL>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, H, W, OH, OW, (float)H / (float)OH,
                       (float)W / (float)OW, n);
}

template <int KIND, int NOPS>
static void launch_asm(const float* src, float* dst, int NC, int H, int W, int OH, int OW, hipStream_t st) {
    const int64_t n = (int64_t)NC * OH * OW;
    hipLaunchKernelGGL((bilinear_asm_kernel<KIND, NOPS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, H, W, OH, OW, (float)H / (float)OH,
                       (float)W / (float)OW, n);
}

// variants 10 + NOPS: packed multiplies, 20 + NOPS: scalar multiplies (NOPS in 0, 1, 2, 4, 8)
extern "C" int bilinear_asm_launch(int variant, const float* src, float* dst, int NC, int H, int W, int OH, int OW, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 10: launch_asm<0, 0>(src, dst, NC, H, W, OH, OW, st); break;
        case 11: launch_asm<0, 1>(src, dst, NC, H, W, OH, OW, st); break;
        case 12: launch_asm<0, 2>(src, dst, NC, H, W, OH, OW, st); break;
        case 14: launch_asm<0, 4>(src, dst, NC, H, W, OH, OW, st); break;
        case 18: launch_asm<0, 8>(src, dst, NC, H, W, OH, OW, st); break;
        case 20: launch_asm<1, 0>(src, dst, NC, H, W, OH, OW, st); break;
        case 40: launch_replica<0>(src, dst, NC, H, W, OH, OW, st); break;
        case 41: launch_replica<1>(src, dst, NC, H, W, OH, OW, st); break;
        case 30: launch_tail<0>(src, dst, NC, H, W, OH, OW, st); break;
        case 31: launch_tail<1>(src, dst, NC, H, W, OH, OW, st); break;
        case 32: launch_tail<2>(src, dst, NC, H, W, OH, OW, st); break;
        case 33: launch_tail<3>(src, dst, NC, H, W, OH, OW, st); break;
        case 34: launch_tail<4>(src, dst, NC, H, W, OH, OW, st); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int bilinear_probe_launch(int variant, const float* src, float* dst, float* taps, int NC, int H, int W, int OH, int OW, void* stream) {
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    const int64_t n = (int64_t)NC * OH * OW;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) hipLaunchKernelGGL(bilinear_probe_kernel<0>, grid, block, 0, st, src, dst, taps, H, W, OH, OW, 0, sh, sw, 1.0f, n);
    else if (variant == 1) hipLaunchKernelGGL(bilinear_probe_kernel<1>, grid, block, 0, st, src, dst, taps, H, W, OH, OW, 0, sh, sw, 1.0f, n);
    else hipLaunchKernelGGL(bilinear_probe_kernel<2>, grid, block, 0, st, src, dst, taps, H, W, OH, OW, 0, sh, sw, 1.0f, n);
    return (int)hipGetLastError();
}
