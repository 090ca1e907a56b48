// The K-major contraction of gemm_tn.hip on the bf16 matrix instruction with bf16x3-split operands:
//
//   C[M][N] += sum_k A[k][m] * B[k][n]        A: [K][lda], m contiguous;  B: [K][ldb], n contiguous
//
// -- the train step's weight gradients dW = dZ^T . X (keras_train.py:195, backward of seq2seq.py:237-390), the largest fp32-input
// share left in the step after round 6's first half (16.7 of 57.5 ms at configs[3]).  Arithmetic, tile shape and LDS layout are
// gemm_split.hip's: every fp32 value x = x0 + x1 + x2 in bf16 (exact), six v_mfma_f32_32x32x16_bf16 products per term with fp32
// accumulation, 256x256 block tile, 8 waves x (64 rows x 128 columns), three bf16 planes per operand in 32-byte rows whose 16-byte
// halves are swapped where bit 4 of the row is set (conflict-free ds_read_b128).
//
// What differs is the staging: the MFMA wants 8 consecutive k of one row per lane, the operands lie with m (n) contiguous.  A thread
// therefore stages rows r0 and r0 + 128 of both operands at k = 4 kc .. 4 kc + 3 as FOUR 4-byte loads each, r0 = tid & 127 being the
// lane-contiguous index: a wave's load instruction reads 256 contiguous bytes of one k-row (as coalesced as the fp32 kernel's 16-byte
// loads), its four values are the thread's own float4 along k -- the transposition costs no shuffle and no extra LDS traffic -- and
// the LDS stores of a wave go to 64 consecutive rows (conflict-free).  Sixteen load instructions per thread and tile instead of four.
// The loop is compiler-scheduled (no hidden loads): tile t + 2 is requested while tile t is contracted, one barrier per tile.
//
// Long K is split over workgroups (one per CU) with float atomics into C, K share fastest over the linear workgroup index so that a
// share's k-rows stay in one XCD's L2 (gemm_tn.hip).  Column sums of A (the bias gradient) ride along as there.
#include "common.h"

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TS_BM = 256, TS_BN = 256, TS_BK = 16;
constexpr int TS_PLANE = 256 * 32;                 // bytes: one bf16 plane of an operand tile
constexpr int TS_BUF = 6 * TS_PLANE;               // A planes 0..2, B planes 0..2
constexpr int TS_LDS = 2 * TS_BUF;                 // 96 KB

__global__ __launch_bounds__(512, 1) void gemm_tn_split256_kernel(const TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nbn = g.N / TS_BN;
    const int nsplit = g.nsplit;
    const int tile = (int)blockIdx.x / nsplit, zidx = (int)blockIdx.x % nsplit;        // K share fastest (gemm_tn.hip)
    const int bn = tile % nbn, bm = tile / nbn;
    const int m0 = bm * TS_BM, n0 = bn * TS_BN;
    const int ktiles_all = g.K / TS_BK;
    const int per = (ktiles_all + nsplit - 1) / nsplit;
    const int kt_begin = zidx * per;
    const int ntiles = ktiles_all - kt_begin < per ? (ktiles_all - kt_begin > 0 ? ktiles_all - kt_begin : 0) : per;
    if (ntiles <= 0) return;

    // ---- staging: thread (r0, kc) holds k = 4 kc .. 4 kc + 3 of rows r0 and r0 + 128 of both operands ----
    const int r0 = tid & 127, kc = tid >> 7;
    const float* ap = g.A + (long long)(kt_begin * TS_BK + 4 * kc) * g.lda + m0 + r0;
    const float* bp = g.B + (long long)(kt_begin * TS_BK + 4 * kc) * g.ldb + n0 + r0;
    const long long astep = (long long)TS_BK * g.lda, bstep = (long long)TS_BK * g.ldb;
    struct GTile { f32x4 a[2], b[2]; };
    auto load_tile = [&](GTile& gt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gt.a[0][i] = ap[(long long)i * g.lda]; gt.a[1][i] = ap[(long long)i * g.lda + 128];
            gt.b[0][i] = bp[(long long)i * g.ldb]; gt.b[1][i] = bp[(long long)i * g.ldb + 128];
        }
        ap += astep; bp += bstep;
    };
    auto split4 = [&](const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2) {       // gemm_split.hip's split, bit for bit
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 v = {x[2 * h], x[2 * h + 1]};
            const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            const f32x2 r1 = v - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
            const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
            const f32x2 r2 = r1 - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
            p0[h] = q0; p1[h] = q1; p2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        }
    };
    // (rows r0 and r0 + 128 share bit 4: one offset serves both)
    const int st_off = r0 * 32 + ((((kc >> 1) ^ (r0 >> 4)) & 1) * 16) + (kc & 1) * 8;
    auto store_op = [&](const f32x4 v0, const f32x4 v1, int buf, int plane0) {
        char* base = ts_smem + buf * TS_BUF + plane0 * TS_PLANE + st_off;
        u32x2 p0, p1, p2;
        split4(v0, p0, p1, p2);
        *reinterpret_cast<u32x2*>(base) = p0; *reinterpret_cast<u32x2*>(base + TS_PLANE) = p1; *reinterpret_cast<u32x2*>(base + 2 * TS_PLANE) = p2;
        split4(v1, p0, p1, p2);
        *reinterpret_cast<u32x2*>(base + 128 * 32) = p0; *reinterpret_cast<u32x2*>(base + TS_PLANE + 128 * 32) = p1;
        *reinterpret_cast<u32x2*>(base + 2 * TS_PLANE + 128 * 32) = p2;
    };
    // column sums of A over k (the bias gradient that goes with a weight gradient): the first column tile of every row tile adds up
    // the values it stages anyway
    const bool do_colsum = g.colsum != nullptr && bn == 0;
    float cs0 = 0.f, cs1 = 0.f;
    auto store_tile = [&](const GTile& gt, int buf) {
        if (do_colsum) { cs0 += (gt.a[0][0] + gt.a[0][1]) + (gt.a[0][2] + gt.a[0][3]); cs1 += (gt.a[1][0] + gt.a[1][1]) + (gt.a[1][2] + gt.a[1][3]); }
        store_op(gt.a[0], gt.a[1], buf, 0);
        store_op(gt.b[0], gt.b[1], buf, 3);
    };
    // a lane's 8 k of its row: the 16-byte half lh (k = 8 lh .. 8 lh + 7, the same for both operands)
    const int fr_off = l31 * 32 + (((lh ^ (l31 >> 4)) & 1) * 16);
    auto frag_a = [&](int buf, int plane, int rb) {
        return *reinterpret_cast<const bf16x8*>(ts_smem + buf * TS_BUF + plane * TS_PLANE + (wm * 64 + rb * 32) * 32 + fr_off);
    };
    auto frag_b = [&](int buf, int plane, int c) {
        return *reinterpret_cast<const bf16x8*>(ts_smem + buf * TS_BUF + (3 + plane) * TS_PLANE + (wn * 128 + c * 32) * 32 + fr_off);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][c][r] = 0.0f;
    bf16x8 fb[4][3], fa[2][3];
#define CASV_TS_MMA(PA, PB)                                                                               \
    _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                  \
            acc[rb_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb_][PA], fb[c_][PB], acc[rb_][c_], 0, 0, 0);

    GTile cur;
    load_tile(cur);
    store_tile(cur, 0);
    if (ntiles > 1) load_tile(cur);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int c = 0; c < 4; ++c) fb[c][p] = frag_b(buf, p, c);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) fa[rb][p] = frag_a(buf, p, rb);
        }
        // tile t + 1 (requested one tile ago) is split and stored into the other buffer, whose fragments were read before the last
        // barrier; tile t + 2 is requested
        if (t + 1 < ntiles) store_tile(cur, buf ^ 1);
        if (t + 2 < ntiles) load_tile(cur);
        // the product order of gemm_split.hip (smallest terms first)
        CASV_TS_MMA(1, 1) CASV_TS_MMA(0, 2) CASV_TS_MMA(0, 1) CASV_TS_MMA(2, 0) CASV_TS_MMA(1, 0) CASV_TS_MMA(0, 0)
        __syncthreads();
    }
#undef CASV_TS_MMA

    if (do_colsum) {            // (every wave is past the K loop's last barrier: the tile buffers are free)
        float* red = reinterpret_cast<float*>(ts_smem);
        red[kc * 256 + r0] = cs0; red[kc * 256 + 128 + r0] = cs1;
        __syncthreads();
        if (tid < 256 && m0 + tid < g.Mstore) atomicAdd(g.colsum + m0 + tid, (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]));
    }
    // ---- epilogue: C += acc (atomics where K is shared out) ----
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + wn * 128 + c * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < g.Mstore) {
                    float* dst = g.C + (long long)m * g.ldc + n;
                    if (nsplit > 1) atomicAdd(dst, acc[rb][c][r]);
                    else *dst = g.accumulate ? (*dst + acc[rb][c][r]) : acc[rb][c][r];
                }
            }
        }
}

// true: launched.  false: the shape has no 256x256 split form (the caller takes the fp32-input kernel).
bool launch_gemm_tn_split(const TnArgs& g, hipStream_t stream) {
    static const bool off = [] { const char* e = getenv("CASV_TN_SPLIT"); return e && e[0] == '0'; }();
    if (off) return false;
    if (g.M <= 0 || g.N <= 0 || g.M % TS_BM || g.N % TS_BN || g.K < 64 * TS_BK || g.K % TS_BK) return false;
    if (g.Mstore <= 0 || g.Mstore > g.M) return false;
    static const int ncu = [] { hipDeviceProp_t pr{}; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
    const int tiles = (g.M / TS_BM) * (g.N / TS_BN), ktiles = g.K / TS_BK;
    // one workgroup per CU: fill the chip, keep >= 64 k-tiles per workgroup so that the atomic epilogue stays a small part
    int ks = ncu / tiles;
    if (ks > ktiles / 64) ks = ktiles / 64;
    if (ks < 1) ks = 1;
    if (tiles * ks < ncu / 2) return false;                 // (too few workgroups for this tile shape: the fp32-input kernel's 128x128 tiles fill the chip better)
    if (ks > 1 && !g.accumulate && !g.out_zeroed) {
        if (g.ldc == g.N) (void)hipMemsetAsync(g.C, 0, (size_t)g.Mstore * g.N * sizeof(float), stream);
        else (void)hipMemset2DAsync(g.C, (size_t)g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.Mstore, stream);
    }
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_split256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    TnArgs gg = g;
    gg.nsplit = ks;
    hipLaunchKernelGGL(gemm_tn_split256_kernel, dim3(tiles * ks), dim3(512), TS_LDS, stream, gg);
    return true;
}

}  // namespace casv
